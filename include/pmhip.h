/* pmhip.h -- C ABI of the MI355X-native PatchMatch depth-map engine (libpmhip.so).
 *
 * Drop-in boundary: the reference dispatches every depth-map estimation through
 *     if (pmCUDA) { pmCUDA->EstimateDepthMap(arrDepthData[idxImage]); return true; }
 * (libs/MVS/SceneDensify.cpp:618-623); the plug-in object has four methods
 * (libs/MVS/PatchMatchCUDA.inl:102-108): ctor(device), Init(bGeomConsistency), Release(),
 * EstimateDepthMap(DepthData&), and is created / re-initialised at
 * libs/MVS/SceneDensify.cpp:1872-1881 and :1910-1916.  Section 1 mirrors exactly that surface
 * with plain pointers and sizes.  Section 2 is the HBM-resident scene interface the reference
 * reaches through files (depthNNNN.dmap written per view and re-read by the geometric rounds,
 * libs/MVS/SceneDensify.cpp:378-414,2095-2117): all images, cameras and depth maps stay on the
 * device and many reference views are estimated concurrently -- this is what fills the GPU.
 *
 * Algorithm = the reference's *CPU* estimator (libs/MVS/DepthMap.cpp:415-971 driven by
 * libs/MVS/SceneDensify.cpp:490-805), not its CUDA variant; see DESIGN.md.
 * All functions return 0 on success, a negative PMHIP_E_* code otherwise; none ever exits.
 * A handle is single-caller (the reference serialises calls with a semaphore,
 * SceneDensify.cpp:2041-2059) but may be used from different host threads sequentially.
 */
#ifndef PMHIP_H_
#define PMHIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMHIP_MAX_SOURCES 16 /* >= OPTDENSE::nMaxViews (12), libs/MVS/DepthMap.cpp:76 */

enum {
	PMHIP_OK = 0,
	PMHIP_E_ARG = -1,      /* bad argument */
	PMHIP_E_SIZE = -2,     /* views differ in size, or image too small for the requested sub-resolution levels */
	PMHIP_E_HIP = -3,      /* HIP runtime error (see pmhip_last_error) */
	PMHIP_E_STATE = -4,    /* call out of order (e.g. estimate before init) */
	PMHIP_E_NODEVICE = -5  /* no usable GPU: the adapter falls back like SceneDensify.cpp:1876-1877 */
};

typedef struct pmhip_engine pmhip_engine;

/* OPTDENSE subset consumed by the estimator; defaults = libs/MVS/DepthMap.cpp:69-114. */
typedef struct PMHipParams {
	uint32_t nSubResolutionLevels;      /* 2 */
	uint32_t nEstimationIters;          /* 3 */
	uint32_t nEstimationGeometricIters; /* 2 (only decides the finalize threshold x1.333, SceneDensify.cpp:774-776) */
	uint32_t nRandomIters;              /* 6 */
	float fEstimationGeometricWeight;   /* 0.1 */
	float fRandomDepthRatio;            /* 0.003 */
	float fRandomAngle1Range;           /* 16 deg */
	float fRandomAngle2Range;           /* 10 deg */
	float fRandomSmoothDepth;           /* 0.02 */
	float fRandomSmoothNormal;          /* 13 deg */
	float fRandomSmoothBonus;           /* 0.93 */
	float fNCCThresholdKeep;            /* 0.9 */
	float fDescriptorMinMagnitudeThreshold; /* 0.02 */
	uint32_t seed;                      /* counter-based RNG seed (the reference seeds from random_device) */
} PMHipParams;

/* DepthData::ViewData (libs/MVS/DepthMap.h:158-205) as plain data; images are contiguous
 * row-major float gray in [0,1] (cv::Mat1f), cameras are x_cam = R (X - C). */
typedef struct PMHipView {
	const float* image;       /* host pointer, w*h floats */
	int32_t w, h;
	double K[9], R[9], C[3];
	const float* depth;       /* nullable; known depth-map of this source view => geometric pass */
	double Kd[9], Rd[9], Cd[3]; /* camera stored with that depth-map (cameraDepthMap); all zero = the view's own camera */
	uint32_t id;              /* global view ID (mixed into the RNG key for views[0]) */
	int32_t dw, dh;           /* size of that depth-map; 0, 0 = w, h.  The map is addressed through Kd/Rd/Cd, so it need not have the image's size */
} PMHipView;

/* DepthData (libs/MVS/DepthMap.h:157-271): views[0] is the reference view. depthMap/normalMap
 * are in/out (zero depth / zero normal == "unset", randomised in the init pass exactly like
 * ScoreDepthMapTmp, SceneDensify.cpp:505-512); confMap is out ([0,1], 1 best); all caller-owned. */
typedef struct PMHipDepthData {
	const PMHipView* views;
	int32_t nViews;           /* 1 + number of source views, 2..1+PMHIP_MAX_SOURCES */
	float* depthMap;          /* w*h */
	float* normalMap;         /* w*h*3, camera space */
	float* confMap;           /* w*h */
	float dMin, dMax;
} PMHipDepthData;

/* ---- 1. PatchMatchCUDA-shaped interface -------------------------------------------------- */
int pmhip_default_params(PMHipParams* p);
/* PatchMatchCUDA::PatchMatchCUDA(int device), PatchMatchCUDA.cpp:46-51 */
int pmhip_create(int device, pmhip_engine** out);
/* PatchMatchCUDA::~PatchMatchCUDA, PatchMatchCUDA.cpp:53-56 */
void pmhip_destroy(pmhip_engine* e);
/* PatchMatchCUDA::Init(bool bGeomConsistency), PatchMatchCUDA.cpp:97-106 */
int pmhip_init(pmhip_engine* e, int bGeomConsistency);
/* PatchMatchCUDA::Release(), PatchMatchCUDA.cpp:58-81 */
int pmhip_release(pmhip_engine* e);
/* PatchMatchCUDA::EstimateDepthMap(DepthData&), PatchMatchCUDA.cpp:174-416; semantics of
 * DepthMapsData::EstimateDepthMap(idx, nGeometricIter), SceneDensify.cpp:616-805:
 * nGeometricIter < 0 -> photometric pass over the pyramid; >= 0 -> that geometric round
 * (requires pmhip_init(e,1) and views[1..].depth). Blocking.
 * views[1..] may have any size (the reference rescales a neighbour whose scale differs by >= 15 %, ViewData::ScaleImage, DepthMap.h:194-204,
 * SceneDensify.cpp:325-349; the resampling itself -- INTER_AREA / INTER_CUBIC -- stays with the caller, as there): every source view is
 * projected into with its own K and sampled within its own bounds, at every pyramid level (ScaleDepthData scales each image by 1/2^l). */
int pmhip_estimate_depth_map(pmhip_engine* e, PMHipDepthData* dd, const PMHipParams* p, int nGeometricIter);
/* The same with DepthData::mask (libs/MVS/DepthMap.h:211): mask = w*h bytes, 0 = the pixel is ignored (see pmhip_scene_set_mask), or NULL.
 * maskOption != 0 says that OPTDENSE::nIgnoreMaskLabel is set even though this view has no mask: the level hand-off then resizes the depth map
 * with INTER_NEAREST as the reference does whenever the option is on (SceneDensify.cpp:661). */
int pmhip_estimate_depth_map_masked(pmhip_engine* e, PMHipDepthData* dd, const unsigned char* mask, int maskOption, const PMHipParams* p, int nGeometricIter);
const char* pmhip_last_error(pmhip_engine* e);

/* ---- 2. HBM-resident scene interface ------------------------------------------------------ */
/* Allocate device storage for nImages views of w x h (image pyramids, depth/normal/conf maps,
 * the previous-round depth snapshot).  Replaces Scene images + depthNNNN.dmap files. */
int pmhip_scene_create(pmhip_engine* e, int nImages, int w, int h, int nLevels);
/* Upload (onDevice == 0) or adopt-by-copy (onDevice != 0, device pointer) one view.
 * neighbors: global IDs of its source views, best first (ViewScore order, DepthMap.h:209). */
int pmhip_scene_set_view(pmhip_engine* e, int idx, const float* gray, int onDevice,
                         const double K[9], const double R[9], const double C[3],
                         float dMin, float dMax, const int32_t* neighbors, int nNeighbors);
/* The identity under which slot idx draws its random numbers (the Philox key of a view is derived from it; default: idx).  A process that holds only PART of a scene
 * -- a rank of the multi-GPU driver keeps its block of reference views and their neighbours in a compact scene of local slots -- gives every slot the view's index in
 * the whole scene, so that the depth maps do not depend on how the scene was split (openmvs_amd/distributed.py; the reference seeds per estimator, DepthMap.cpp:370-372). */
int pmhip_scene_set_view_id(pmhip_engine* e, int idx, uint32_t viewID);
/* The same for a view whose image has its own size w x h (the reference sizes every DepthData on its own image, DepthMapsData::InitViews,
 * libs/MVS/SceneDensify.cpp:306-459; also a neighbour rescaled by ViewData::ScaleImage): it keeps its own pyramid and its own depth / normal /
 * confidence maps of that size, and is a view like any other -- source view, reference view (a batch is swept one size class after the other),
 * per-map filters, cross-view filter and fusion against neighbours of other sizes.  pmhip_scene_get_maps / set_maps / set_conf / set_color of such
 * a view (and its ignore mask, pmhip_scene_set_mask) have w*h entries; pmhip_scene_device_ptr returns its own buffers.  gray is required. */
int pmhip_scene_set_view_sized(pmhip_engine* e, int idx, const float* gray, int w, int h, int onDevice,
                               const double K[9], const double R[9], const double C[3],
                               float dMin, float dMax, const int32_t* neighbors, int nNeighbors);
/* Known depth-map of view idx (host pointer, dw*dh floats) for the geometric rounds in which it is a source view, with the camera it was stored
 * with (DepthData::ViewData::depthMap / cameraDepthMap, loaded from the neighbour's .dmap at SceneDensify.cpp:378-393).  While installed it is
 * read instead of the scene's snapshot of that view; depth == NULL removes it. */
int pmhip_scene_set_source_depth(pmhip_engine* e, int idx, const float* depth, int dw, int dh,
                                 const double Kd[9], const double Rd[9], const double Cd[3]);
/* Estimate the depth maps of viewIds[0..nViews) concurrently (one EstimateDepthMap per view,
 * SceneDensify.cpp:616-805).  nGeometricIter < 0: photometric; >= 0: geometric round reading the
 * snapshot taken by pmhip_scene_commit_round.  Asynchronous on the engine stream unless sync != 0. */
int pmhip_scene_estimate(pmhip_engine* e, const int32_t* viewIds, int nViews, const PMHipParams* p,
                         int nGeometricIter, int sync);
/* End of a round: snapshot depth maps of all views as the "saved .dmap" the next geometric round
 * reads (the reference renames depthNNNN.geo.dmap -> .dmap, SceneDensify.cpp:1943-1950). */
int pmhip_scene_commit_round(pmhip_engine* e);
/* Clear a view's maps to "unset" (InitViews with an empty point cloud, SceneDensify.cpp:417-425). */
int pmhip_scene_reset_view(pmhip_engine* e, int idx);
/* Provide an initial depth/normal estimate (host pointers, nullable each). */
int pmhip_scene_set_maps(pmhip_engine* e, int idx, const float* depth, const float* normal);
/* Ignore mask of a view (OPTDENSE::nIgnoreMaskLabel, DepthEstimator::ImportIgnoreMask, DepthMap.cpp:296-323): w*h bytes at image
 * resolution, 0 = the pixel is ignored -- not estimated at any pyramid level, depth / normal / confidence 0 -- exactly like the masked
 * MapMatrix2ZigzagIdx + DepthData::ApplyIgnoreMask (SceneDensify.cpp:679-683); level masks are INTER_NEAREST resamples.  NULL removes it.
 * While masks are in use the level hand-off resizes depth maps with INTER_NEAREST as the reference does (:661);
 * pmhip_scene_set_mask_mode forces that on (1) / off (0) for scenes where the option is set but no view has a mask file; -1 = automatic. */
int pmhip_scene_set_mask(pmhip_engine* e, int idx, const unsigned char* mask);
int pmhip_scene_set_mask_mode(pmhip_engine* e, int mode);
/* Install a confidence map (host pointer), e.g. one read back from a .dmap before filtering / fusing without re-estimating. */
int pmhip_scene_set_conf(pmhip_engine* e, int idx, const float* conf);
/* Download maps (any pointer may be NULL); a view with its own size returns maps of that size. */
int pmhip_scene_get_maps(pmhip_engine* e, int idx, float* depth, float* normal, float* conf);
/* Device pointers for collectives (RCCL all-gather of the snapshot / broadcast of images).
 * what: 0 image level 0, 1 depth, 2 normal, 3 conf, 4 snapshot depth.  The per-kind arrays are
 * single contiguous allocations ordered by view index, so idx 0 addresses the whole set (views with their own size live outside them: their
 * pointer addresses that view alone, and pmhip_scene_copy takes them one by one).  After writing images call pmhip_scene_images_updated,
 * after writing depth maps pmhip_scene_maps_updated: the engine cannot see writes through these pointers. */
void* pmhip_scene_device_ptr(pmhip_engine* e, int what, int idx);
/* DepthMapsData::FilterDepthMap (SceneDensify.cpp:1050-1299) for each view of viewIds against its first <= 8 neighbours
 * (Scene::DenseReconstructionFilter, :2136-2170): cross-view splat + z-test, then the confidence-weighted fusion
 * (bAdjust != 0, OPTDENSE::bFilterAdjust) or the strict agreement test.  Defaults: nMinViewsFilter 2, nMinViewsFilterAdjust 1,
 * fDepthDiffThreshold 0.01 (DepthMap.cpp:77-78,91).  Results are staged until pmhip_scene_filter_commit, because every view
 * must be filtered against the unfiltered maps of its neighbours (:2183-2210). */
int pmhip_scene_filter(pmhip_engine* e, const int32_t* viewIds, int nViews, int bAdjust, uint32_t nMinViewsFilter,
                       uint32_t nMinViewsFilterAdjust, float fDepthDiffThreshold, int sync);
int pmhip_scene_filter_commit(pmhip_engine* e);
/* DepthMapsData::RemoveSmallSegments (SceneDensify.cpp:809-900; OPTDENSE::nSpeckleSize 100, fDepthDiffThreshold 0.01): invalidates
 * depth segments smaller than nSpeckleSize pixels, in place on the device; same result as the sequential region growing. */
int pmhip_scene_remove_small_segments(pmhip_engine* e, const int32_t* viewIds, int nViews, uint32_t nSpeckleSize, float fDepthDiffThreshold);
/* DepthMapsData::GapInterpolation (SceneDensify.cpp:904-1045; OPTDENSE::nIpolGapSize 7, fDepthDiffThreshold 0.01): fills row then
 * column gaps of the depth / normal / confidence maps of these views, in place on the device. */
int pmhip_scene_gap_interpolation(pmhip_engine* e, const int32_t* viewIds, int nViews, uint32_t nIpolGapSize, float fDepthDiffThreshold);
/* DepthMapsData::FuseDepthMaps (SceneDensify.cpp:1372-1650) over the resident depth / normal / confidence maps: every unclaimed depth
 * seeds a 3D point, claims the agreeing pixels of the view's neighbours (depth within fDepthDiffThreshold, normals within
 * fNormalDiffThreshold degrees), is kept if it gathered nMinViewsFuse views and then zeroes the neighbour depths it occludes.
 * `order` lists the views to fuse, best connected first (the reference sorts by Image::neighbors.size(), :1423-1450); each view uses
 * the neighbour list given to pmhip_scene_set_view (the IDs stored in its .dmap).  Runs on working copies: the scene's maps are not
 * modified.  Point order, views, weights, positions, colours and normals equal the sequential reference (see csrc/pm_fuse.h).
 * nMinViewsFuse < 2 selects DepthMapsData::MergeDepthMaps instead (:1305-1368, as Scene::DenseReconstruction does, :1695-1698): every valid
 * depth becomes a single-view point (pass the views in index order; weights come back 0, the reference produces none).
 * Defaults (DepthMap.cpp:75,92,93,101,102): nMinViewsFuse 2, fDepthDiffThreshold 0.01, fNormalDiffThreshold 25, colours and normals on. */
typedef struct PMHipFuseParams {
	uint32_t nMinViewsFuse;
	float fDepthDiffThreshold;
	float fNormalDiffThreshold;   /* degrees */
	int32_t bEstimateColor;       /* needs pmhip_scene_set_color for every view */
	int32_t bEstimateNormal;
} PMHipFuseParams;
/* 8-bit BGR image of view idx at depth-map resolution (Image::image, read at SceneDensify.cpp:1529,1567), host pointer, w*h*3 bytes. */
int pmhip_scene_set_color(pmhip_engine* e, int idx, const unsigned char* bgr);
int pmhip_scene_fuse(pmhip_engine* e, const int32_t* order, int nOrder, const PMHipFuseParams* params,
                     uint64_t* nPoints, uint64_t* nViews, uint64_t* nDepths);
/* Download the fused cloud (PointCloud::points / pointViews / pointWeights / colors / normals, libs/MVS/PointCloud.h): points 3*nPoints
 * floats, viewStart nPoints+1 offsets into views / weights / projs (nViews entries; projs = x,y of the pixel each view contributed),
 * colors 3*nPoints BGR bytes, normals 3*nPoints floats.  Any pointer may be NULL. */
int pmhip_scene_fuse_get(pmhip_engine* e, float* points, uint32_t* viewStart, uint32_t* views, float* weights, uint16_t* projs,
                         unsigned char* colors, float* normals);
/* Reservation rounds the last pmhip_scene_fuse needed, summed over its views (diagnostic). */
uint64_t pmhip_scene_fuse_rounds(pmhip_engine* e);
/* Device-to-device copy between a caller buffer (e.g. a torch tensor used for an RCCL collective)
 * and `count` consecutive views of one per-kind array, starting at view firstIdx; `what` as above.
 * toEngine != 0 copies caller -> engine.  Asynchronous on the engine stream. */
int pmhip_scene_copy(pmhip_engine* e, int what, int firstIdx, int count, void* devPtr, int toEngine);
/* Rebuild image pyramids after image level 0 was written through pmhip_scene_device_ptr. */
int pmhip_scene_images_updated(pmhip_engine* e);
/* Depth maps of views [firstIdx, firstIdx + count) were written through pmhip_scene_device_ptr(e, 1, ...) (e.g. by an RCCL all-gather straight
 * into the depth array): marks them as holding a depth map, which pmhip_scene_filter / pmhip_scene_fuse require of a neighbour before they use it
 * (DepthData::IsValid(), SceneDensify.cpp:2150-2163).  pmhip_scene_estimate, _set_maps and _copy(what == 1, toEngine) do this themselves; a raw
 * pointer write cannot, and such neighbours would be skipped without this call. */
int pmhip_scene_maps_updated(pmhip_engine* e, int firstIdx, int count);
/* Device memory the resident scene holds right now, in bytes (images with their layouts, maps, masks, filter staging, batch scratch; not the fusion's working buffers). */
uint64_t pmhip_scene_bytes(pmhip_engine* e);
int pmhip_sync(pmhip_engine* e);
/* Engine stream (hipStream_t) so callers can bracket work with their own events. */
void* pmhip_stream(pmhip_engine* e);

/* Timing of the dominant kernel (the diagonal sweep) measured with HIP events on the streams the
 * launches go to, since the last reset: number of launches, summed milliseconds (sweepMs / sweepLaunches
 * == the kernel's average duration as rocprofv3 reports it, also when view groups run concurrently),
 * summed algorithmic bytes (SURVEY.md 8d model) and pixel-updates. */
typedef struct PMHipKernelStats {
	uint64_t sweepLaunches; double sweepMs; double sweepBytes; uint64_t sweepPixels;
	uint64_t initLaunches; double initMs;
	double sweepWallMs; /* wall time of the passes (hand-off, init, sweeps, finalize of all view groups, which overlap); sweepMs sums the per-stream times of the sweeps alone */
	double sweepHostMs; /* host time spent enqueueing the passes (a few per cent of sweepWallMs: one thread enqueues a launch in ~3 us, profiles/r04_call9_launch_rate.log) */
} PMHipKernelStats;
int pmhip_stats_reset(pmhip_engine* e, int enableEvents);
int pmhip_stats_get(pmhip_engine* e, PMHipKernelStats* out);

/* How the engine maps a batch onto the GPU -- never WHAT it computes: every setting gives the same bits.  pmhip_create fills the defaults (the measured choices of
 * csrc/pm_engine.hip); 0 in a field of pmhip_set_tuning keeps the current value.  The library reads no environment variable. */
typedef struct PMHipTuning {
	int32_t viewGroups;      /* view groups of a batch; each runs its whole pass (every level's hand-off, init, sweeps, finalize) on its own stream, the groups meet at the end of the call (2) */
	int32_t wideMaxViews;    /* batches of at most this many reference views use the speculative sweep kernels for every launch (32); -1 = no speculative kernels at all (also clears widePixels / wide8Pixels unless set in the same call) */
	int32_t wideHyps;        /* hypotheses per round of the speculative kernel: 8, 4 or 2; -1 = by batch size (8 for one or two views, else 2) */
	int32_t sweepLanes;      /* lanes per pixel of pm_sweep2_kernel: 4, 8 or 16; -1 = by batch size */
	int32_t quadBuffer;      /* 1: tap rows address the level's quad images as one buffer, 2: through each view's pointer */
	int32_t widePixels;      /* larger batches: a diagonal launch of at most this many pixels (diagonal length x views of the group) uses the two-wide speculative kernel (20000); -1 = none */
	int32_t wide8Pixels;     /* ... and one of at most this many pixels the eight-wide speculative kernel; -1 = none */
	int32_t reserved0;       /* ignored by pmhip_set_tuning, 0 from pmhip_get_tuning (`launchThreads` until round 4, absent in round 5: the struct is 32 bytes again -- PMHIP_ABI_VERSION) */
} PMHipTuning;
/* OPT-IN, NOT the reference's estimator: tiled sweeps.  The pixels that take part in the estimation are cut into tileW x tileH tiles; a sweep (DepthMap.cpp:329-356 order)
 * runs inside every tile, and a neighbour in another tile is read as the previous sweep left it.  The tiles of a sweep are independent, so a sweep is tileW + tileH - 1
 * launches that each cover every tile of every view instead of w + h launches of one anti-diagonal: what a small batch (one depth map per call, a rank of an 8-GPU split)
 * needs to fill the GPU.  The result is deterministic and equals oracle/pm_oracle.cpp with Opt::tileW / tileH bit for bit, but it is NOT the sequential sweep's: against
 * that it differs like two runs of the reference differ from each other.  0, 0 = off (the default: the reference's sweep, bit for bit). */
int pmhip_set_sweep_tiles(pmhip_engine* e, int tileW, int tileH);
/* The layout of the structs of this header as a number: a host program compares pmhip_abi_version() with the PMHIP_ABI_VERSION it was compiled against before it hands the
 * library a struct (6: PMHipTuning is 32 bytes again -- reserved0; pmhip_set_sweep_tiles). */
#define PMHIP_ABI_VERSION 6u
uint32_t pmhip_abi_version(void);
int pmhip_get_tuning(pmhip_engine* e, PMHipTuning* out);
int pmhip_set_tuning(pmhip_engine* e, const PMHipTuning* t);

/* ---- 3. self-test hooks (used by tests/, no oracle involved) ------------------------------ */
/* Evaluate csrc/pm_math.h on the device: kind 0 exp, 1 acos, 2 atan2(a,b), 3 sin, 4 cos, 5 sqrt, 6 a/b, 7 (float)sqrt((double)a*a+(double)b*b). */
int pmhip_math_eval(pmhip_engine* e, int kind, const float* a, const float* b, float* out, size_t n);
/* Per-phase cycle counters of the sweep kernel (all zero unless the library was built with -DPM_PROFILE). */
int pmhip_prof_get(pmhip_engine* e, unsigned long long out16[16], int reset);
/* Device resampling kernels on host buffers: kind 0 area (factor f = arg), 1 linear x2, 2 nearest x2. */
int pmhip_resize(pmhip_engine* e, int kind, const float* src, int w, int h, int arg, float* dst);

#ifdef __cplusplus
}
#endif
#endif /* PMHIP_H_ */
