// pm_engine.hip -- host side of libpmhip.so: the extern "C" boundary declared in include/pmhip.h.
//
// Mirrors the reference's GPU plug-in boundary (PatchMatchCUDA, libs/MVS/PatchMatchCUDA.inl:102-108)
// and the pass structure of DepthMapsData::EstimateDepthMap (libs/MVS/SceneDensify.cpp:616-805):
// per pyramid level {level hand-off, init-score pass, nEstimationIters sweeps}, then finalize.
// Everything stays in HBM between passes; one stream; no host sync inside a call.
#include "../../include/pmhip.h"
#include "pm_kernels.hip"
#include "pm_band.hip"
#include "pm_wide_n.hip"
#include "pm_filter.hip"
#include "pm_fuse.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <set>
#include <string>
#include <vector>

#ifndef PMHIP_DEFAULT_WIDE_PIXELS
#define PMHIP_DEFAULT_WIDE_PIXELS 20000   // larger batches: diagonal launches of at most this many pixels (diagonal length x views of the group) use the two-wide speculative
                                         // kernel -- the ramps of the fine level and all of the coarse ones.  profiles/r04_call9_lanes_100.log (100 views, Mpix/s): none 46.4,
                                         // <= 8000: 48.1, <= 16000: 48.6, <= 24000: 48.6; 50 views: 43.4 against 42.7 with the two-wide kernel everywhere
#define PMHIP_DEFAULT_WIDE8_PIXELS 0     // ... and of at most this many pixels the eight-wide one (no effect measured at 13 / 100 views: off)
#endif
#ifndef PMHIP_WIDE_DIAGONAL
#define PMHIP_WIDE_DIAGONAL 400   // ... and of at most this many pixels per view of the group
#endif
#ifndef PMHIP_DEFAULT_WIDE
#define PMHIP_DEFAULT_WIDE 32   // batches of at most this many reference views use the speculative kernels: eight hypotheses per round (one wave per pixel) for 1-2 views, two per round
                                // (four pixels per wave, pm_wide_n.hip) from 3 views on.  Measured in round 4 (profiles/r04_call7_lanes_*.log, full schedule at 1920x1080, Mpix/s;
                                // two-wide / pm_sweep2_kernel): 13 views 26.8 / 19.7, 25: 37.5 / 31.2, 50: 41.0 / 39.3, 100: 42.6 / 44.9 (<4,2>)
#endif
#ifndef PMHIP_DEFAULT_LANES
#define PMHIP_DEFAULT_LANES 0    // sweep kernel: lanes per pixel; 0 = by batch size (one view per lane for small batches, four lanes and two views per lane
                                 // from PMHIP_LANES4_FROM reference views on: measured 41.2 vs 39.7 Mpix/s at 100 views, 11.5 vs 16.6 at 13, profiles/r03_variants_call6_sweep2.log)
#ifndef PMHIP_LANES4_FROM
#define PMHIP_LANES4_FROM 80     // measured (profiles/r03_variants_call21_mid_batches.log): one view per lane wins up to 70 reference views (25: 28.2 vs 21.2, 50: 36.8 vs 33.3, 70: 38.9 vs 38.2 Mpix/s), two per lane at 100 (42.9 vs 41.5)
#endif
#endif
#ifndef PMHIP_DEFAULT_GROUPS
#define PMHIP_DEFAULT_GROUPS 2
#endif
namespace {

#define HIPCHK(e, call) do { hipError_t _r = (call); if (_r != hipSuccess) { (e)->err = std::string(#call) + ": " + hipGetErrorString(_r); return PMHIP_E_HIP; } } while (0)

struct SceneView {
	double K[9], R[9], C[3];
	float dMin = 0, dMax = 0;
	int nNb = 0; int nb[PM_MAX_SRC];
	uint32_t id = 0;
	bool set = false;
	bool hasMaps = false;   // a depth map exists for this view (estimated, uploaded or copied in): DepthData::IsValid() of the reference's filter / fuse loops
	// A view whose image has another size than the scene's (the reference sizes every DepthData on its own, DepthMapsData::InitViews, SceneDensify.cpp:306-459; a
	// neighbour rescaled by ViewData::ScaleImage, DepthMap.h:194-204): it keeps its own pyramid and its own maps here.  sw == 0: image and maps live in the scene arrays.
	int sw = 0, sh = 0;
	float *oDepth = nullptr, *oNormal = nullptr, *oConf = nullptr, *oSnap = nullptr;   // sw x sh (x 3): depth, normal, confidence (cost), previous round's depth
	float *oFDepth = nullptr, *oFConf = nullptr;                                       // staged results of the cross-view filter (pmhip_scene_filter / _commit)
	uint8_t* oBgr = nullptr;                                                           // its 8-bit BGR image (pmhip_scene_set_color), sw x sh x 3
	unsigned char* oMask[4] = {nullptr, nullptr, nullptr, nullptr};                    // its ignore mask per pyramid level (pmhip_scene_set_mask)
	float* sImg[4] = {nullptr, nullptr, nullptr, nullptr};
	float* sImgS[4] = {nullptr, nullptr, nullptr, nullptr};
	float4* sImgQ[4] = {nullptr, nullptr, nullptr, nullptr};
	bool sideDirty = false;
	// A known depth-map of this view to be read by geometric rounds instead of the scene's snapshot, of its own size and with the camera it
	// was stored with (DepthData::ViewData::depthMap / cameraDepthMap, filled from the neighbour's .dmap at SceneDensify.cpp:378-393)
	float* sDepth = nullptr; int dw = 0, dh = 0; double Kd[9], Rd[9], Cd[3];
};
static int lvlSize(int n, int l) { return (int)nearbyint((double)n / (double)(1 << l)); }   // cvRound(size / 2^l), ties to even
static void freeSide(SceneView& v) {
	for (int l = 0; l < 4; ++l) { if (v.sImg[l]) hipFree(v.sImg[l]); if (v.sImgS[l]) hipFree(v.sImgS[l]); if (v.sImgQ[l]) hipFree(v.sImgQ[l]); v.sImg[l] = v.sImgS[l] = nullptr; v.sImgQ[l] = nullptr; }
	if (v.sDepth) hipFree(v.sDepth);
	if (v.oDepth) hipFree(v.oDepth); if (v.oNormal) hipFree(v.oNormal); if (v.oConf) hipFree(v.oConf); if (v.oSnap) hipFree(v.oSnap);
	if (v.oFDepth) hipFree(v.oFDepth); if (v.oFConf) hipFree(v.oFConf); if (v.oBgr) hipFree(v.oBgr);
	for (int l = 0; l < 4; ++l) { if (v.oMask[l]) hipFree(v.oMask[l]); v.oMask[l] = nullptr; }
	v.oDepth = v.oNormal = v.oConf = v.oSnap = v.oFDepth = v.oFConf = nullptr; v.oBgr = nullptr;
	v.sDepth = nullptr; v.sw = v.sh = v.dw = v.dh = 0; v.sideDirty = false;
}

// cv::Matx product convention (accumulate from 0, left to right)
void mul33(const double* a, const double* b, double* c) {
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[i*3+k] * b[k*3+j]; c[i*3+j] = s; }
}
void mul31(const double* a, const double* v, double* c) {
	for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += a[i*3+k] * v[k]; c[i] = s; }
}
void transp33(const double* a, double* t) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[j*3+i] = a[i*3+j]; }
// cv::Matx<double,3,3>::inv(): adjugate / determinant (OpenCV matx.hpp, Matx_FastInvOp<_Tp,3,3>)
void inv33(const double* a, double* b) {
	double d = a[0]*(a[4]*a[8] - a[7]*a[5]) - a[1]*(a[3]*a[8] - a[6]*a[5]) + a[2]*(a[3]*a[7] - a[6]*a[4]);
	d = 1 / d;
	b[0] = (a[4]*a[8] - a[5]*a[7]) * d; b[1] = (a[2]*a[7] - a[1]*a[8]) * d; b[2] = (a[1]*a[5] - a[2]*a[4]) * d;
	b[3] = (a[5]*a[6] - a[3]*a[8]) * d; b[4] = (a[0]*a[8] - a[2]*a[6]) * d; b[5] = (a[2]*a[3] - a[0]*a[5]) * d;
	b[6] = (a[3]*a[7] - a[4]*a[6]) * d; b[7] = (a[1]*a[6] - a[0]*a[7]) * d; b[8] = (a[0]*a[4] - a[1]*a[3]) * d;
}
// Camera::InvK, libs/MVS/Camera.h:176-185
void invK(const double* K, double* o) {
	for (int i = 0; i < 9; ++i) o[i] = (i % 4 == 0) ? 1.0 : 0.0;
	o[0] = 1.0 / K[0]; o[4] = 1.0 / K[4]; o[2] = -K[2] * o[0]; o[5] = -K[5] * o[4];
}
// Camera::ScaleK(K, size, newSize), libs/MVS/Camera.h:160-170
void scaleK(const double* K, int w, int h, int nw, int nh, double* o) {
	const double sx = (double)nw / (double)w, sy = (double)nh / (double)h;
	o[0] = K[0]*sx; o[1] = K[1]*sx; o[2] = (K[2]+0.5)*sx-0.5;
	o[3] = 0;       o[4] = K[4]*sy; o[5] = (K[5]+0.5)*sy-0.5;
	o[6] = 0; o[7] = 0; o[8] = 1;
}

} // namespace

struct pmhip_engine {
	int device = 0;
	hipStream_t stream = nullptr;
	// view groups of a batch sweep on their own streams so that the tail of one group's diagonal launch
	// overlaps the next launch of another group (views are independent; diagonals of one view are not)
	int nGroups = 1;
	int wideHyps = 0;                        // hypotheses per round of the speculative kernel: 0 = by batch size (8 for one or two views, else 2); PMHipTuning::wideHyps = 8 / 4 / 2 fixes it
	int wideMaxViews = PMHIP_DEFAULT_WIDE;   // batches of at most this many views use the one-wave-per-pixel sweep kernel (PMHIP_WIDE)
	int widePixels = PMHIP_DEFAULT_WIDE_PIXELS;     // in larger batches: a diagonal launch of at most this many pixels (diagonal length x views of the group) uses the two-wide speculative kernel (PMHIP_WIDE_PIXELS)
	int wide8Pixels = PMHIP_DEFAULT_WIDE8_PIXELS;   // ... and one of at most this many pixels the eight-wide one (PMHIP_WIDE8_PIXELS)
	int sweepLanes = PMHIP_DEFAULT_LANES;   // lanes per pixel of the sweep kernel (PMHIP_LANES); the rest of a view's sources go to views-per-lane
	int quadBuffer = 1;                     // tap rows address the level's quad images as one buffer (entry index); 0 = through each view's own pointer (PMHipTuning::quadBuffer; forced for
	                                        // batches that read a source view with its own image size, which lives outside the level's buffer)
	hipStream_t gstream[16] = {};
	hipEvent_t forkEv = nullptr, joinEv[16] = {};
	bool inited = false, geom = false;
	std::string err;
	// scene (HBM resident)
	int nImages = 0, w = 0, h = 0, nLevels = 0; // nLevels = sub-resolution levels available (pyramid has nLevels+1 entries)
	float* d_img[4] = {nullptr, nullptr, nullptr, nullptr};
	float* d_imgS[4] = {nullptr, nullptr, nullptr, nullptr}; // folded anti-diagonal-major copies (PMTask::refS: the reference patch of a sweep visit), w_l*h_l floats per image
	float4* d_imgQ[4] = {nullptr, nullptr, nullptr, nullptr}; // anti-diagonal-major quad images (PMSrcView::imgQ): texel (u,v)'s entry at (u+v)*h_l + v, (w_l+h_l-1)*h_l entries of 16 bytes per image
	size_t skewPitch(int l) const { return (size_t)(lw(l) + lh(l) - 1) * lh(l); }
	float *d_depth = nullptr, *d_normal = nullptr, *d_conf = nullptr, *d_snap = nullptr;
	// ignore masks (nIgnoreMaskLabel): per level [nImages][P_l] bytes, allocated with the first mask; maskMode -1 = on iff a mask is set
	unsigned char* d_mask[4] = {nullptr, nullptr, nullptr, nullptr}; std::vector<unsigned char> hasMask; bool maskDirty = false; int maskMode = -1;
	// FilterDepthMap staging: filtered depth/conf of every view (committed after all views are filtered) and splat buffers
	float *d_fdepth = nullptr, *d_fconf = nullptr; unsigned char* d_fvalid = nullptr;
	unsigned long long* d_splat = nullptr; int splatCap = 0; size_t splatPix = 0; PMFTask* d_ftasks = nullptr; PMFTask* h_ftasks = nullptr; int ftaskCap = 0;
	std::vector<SceneView> views;
	bool pyramidDirty = true;
	// FuseDepthMaps state (pm_fuse.hip); buffers live until the scene is released
	struct Fuse {
		float* depth = nullptr; uint32_t *claimed = nullptr, *resv = nullptr; uint8_t* bgr = nullptr; PMFuseCam* cams = nullptr;
		uint8_t *recN = nullptr, *recColor = nullptr; float *recX = nullptr, *recWeight = nullptr, *recNormal = nullptr; uint32_t *recView = nullptr, *recProj = nullptr;
		uint32_t* pend[2] = {nullptr, nullptr}; uint32_t* counters = nullptr; unsigned long long* nDepthsDev = nullptr;
		uint2 *tileSums = nullptr, *tileOff = nullptr;
		PMFuseOut out{}; size_t cap = 0;
		uint32_t* pin = nullptr;
		std::vector<unsigned char> hasBgr;
		uint64_t nPoints = 0, nViews = 0, nDepths = 0, rounds = 0; bool haveColor = false, haveNormal = false;
		size_t slab = 0;                       // pixels per image the buffers above were allocated for
		// scenes whose views differ in size: every image's normal / confidence / colour gathered into [nImages][slab] arrays, and the sizes
		float *normalS = nullptr, *confS = nullptr; uint8_t* bgrS = nullptr; int* dims = nullptr;
	} fu;
	// batch scratch (grow only)
	int batchCap = 0;
	float* d_lvl[4] = {nullptr, nullptr, nullptr, nullptr}; // level l>=1: [batch][6][h_l*w_l]; level 0: prior [batch][h*w]
	float* d_old[4] = {nullptr, nullptr, nullptr, nullptr}; // tiled sweeps only: [batch][5][h_l*w_l] -- depth, normal, conf as the running sweep found them (PMTask::depthOld ...)
	int oldCap = 0, oldW = 0, oldH = 0;
	int tileW = 0, tileH = 0;                                // pmhip_set_sweep_tiles: 0 = the reference's sweep
	PMTask* d_tasks = nullptr; PMTask* h_tasks = nullptr;     // [4 levels][batchCap]
	PMUpTask* d_ups = nullptr; PMUpTask* h_ups = nullptr;     // [4][batchCap]
	// single-view interface staging
	bool ownsSingle = false;
	// stats
	bool statsOn = false;
	struct Ev { hipEvent_t a, b; int kind; };
	std::vector<Ev> events;
	PMHipKernelStats stats{};
	// cv::resize(img, img, Size(), 1/2^l, 1/2^l): output size = cvRound(size / 2^l), ties to even (ScaleDepthData, SceneDensify.cpp:586)
	int lw(int l) const { return (int)nearbyint((double)w / (double)(1 << l)); }
	int lh(int l) const { return (int)nearbyint((double)h / (double)(1 << l)); }
	// a view's own size and maps (a view with its own size keeps them itself, SceneView::sw)
	int vw(int i) const { return views[i].sw ? views[i].sw : w; }
	int vh(int i) const { return views[i].sw ? views[i].sh : h; }
	size_t vpix(int i) const { return (size_t)vw(i) * vh(i); }
	float* depthOf(int i) const { return views[i].sw ? views[i].oDepth : d_depth + (size_t)w * h * i; }
	float* normalOf(int i) const { return views[i].sw ? views[i].oNormal : d_normal + (size_t)w * h * 3 * i; }
	float* confOf(int i) const { return views[i].sw ? views[i].oConf : d_conf + (size_t)w * h * i; }
	float* snapOf(int i) const { return views[i].sw ? views[i].oSnap : d_snap + (size_t)w * h * i; }
	int batchW = 0, batchH = 0;   // size the batch scratch was allocated for
};

static void freeFuseOut(pmhip_engine* e) {
	auto& f = e->fu;
	if (f.out.points) hipFree(f.out.points); if (f.out.viewStart) hipFree(f.out.viewStart); if (f.out.views) hipFree(f.out.views);
	if (f.out.weights) hipFree(f.out.weights); if (f.out.projs) hipFree(f.out.projs); if (f.out.colors) hipFree(f.out.colors); if (f.out.normals) hipFree(f.out.normals);
	f.out = PMFuseOut{}; f.cap = 0;
}
static void freeFuse(pmhip_engine* e) {
	auto& f = e->fu;
	void* ptrs[] = {f.depth, f.claimed, f.resv, f.bgr, f.cams, f.recN, f.recColor, f.recX, f.recWeight, f.recNormal, f.recView, f.recProj,
	                f.pend[0], f.pend[1], f.counters, f.nDepthsDev, f.tileSums, f.tileOff, f.normalS, f.confS, f.bgrS, f.dims};
	for (void* q : ptrs) if (q) hipFree(q);
	if (f.pin) hipHostFree(f.pin);
	freeFuseOut(e);
	f = pmhip_engine::Fuse{};
}

static void freeScene(pmhip_engine* e) {
	hipSetDevice(e->device);
	freeFuse(e);
	for (int l = 0; l < 4; ++l) { if (e->d_img[l]) hipFree(e->d_img[l]); e->d_img[l] = nullptr; if (e->d_imgS[l]) hipFree(e->d_imgS[l]); e->d_imgS[l] = nullptr; if (e->d_imgQ[l]) hipFree(e->d_imgQ[l]); e->d_imgQ[l] = nullptr; if (e->d_lvl[l]) hipFree(e->d_lvl[l]); e->d_lvl[l] = nullptr; if (e->d_old[l]) hipFree(e->d_old[l]); e->d_old[l] = nullptr; }
	e->oldCap = 0;
	if (e->d_depth) hipFree(e->d_depth); if (e->d_normal) hipFree(e->d_normal); if (e->d_conf) hipFree(e->d_conf); if (e->d_snap) hipFree(e->d_snap);
	e->d_depth = e->d_normal = e->d_conf = e->d_snap = nullptr;
	for (int l = 0; l < 4; ++l) { if (e->d_mask[l]) hipFree(e->d_mask[l]); e->d_mask[l] = nullptr; }
	e->hasMask.clear(); e->maskDirty = false; e->maskMode = -1;
	if (e->d_fdepth) hipFree(e->d_fdepth); if (e->d_fconf) hipFree(e->d_fconf); if (e->d_fvalid) hipFree(e->d_fvalid);
	if (e->d_splat) hipFree(e->d_splat); if (e->d_ftasks) hipFree(e->d_ftasks); if (e->h_ftasks) hipHostFree(e->h_ftasks);
	e->d_fdepth = e->d_fconf = nullptr; e->d_fvalid = nullptr; e->d_splat = nullptr; e->d_ftasks = nullptr; e->h_ftasks = nullptr; e->splatCap = e->ftaskCap = 0; e->splatPix = 0;
	if (e->d_tasks) hipFree(e->d_tasks); if (e->h_tasks) hipHostFree(e->h_tasks);
	if (e->d_ups) hipFree(e->d_ups); if (e->h_ups) hipHostFree(e->h_ups);
	e->d_tasks = nullptr; e->h_tasks = nullptr; e->d_ups = nullptr; e->h_ups = nullptr;
	for (SceneView& v : e->views) freeSide(v);
	e->batchCap = 0; e->batchW = e->batchH = 0; e->nImages = 0; e->views.clear();
}

static int ensureBatch(pmhip_engine* e, int n, int bw, int bh) {
	if (n <= e->batchCap && bw <= e->batchW && bh <= e->batchH) return 0;
	n = std::max(n, e->batchCap); bw = std::max(bw, e->batchW); bh = std::max(bh, e->batchH);
	HIPCHK(e, hipStreamSynchronize(e->stream));
	for (int l = 0; l < 4; ++l) { if (e->d_lvl[l]) hipFree(e->d_lvl[l]); e->d_lvl[l] = nullptr; }
	if (e->d_tasks) hipFree(e->d_tasks); if (e->h_tasks) hipHostFree(e->h_tasks);
	if (e->d_ups) hipFree(e->d_ups); if (e->h_ups) hipHostFree(e->h_ups);
	e->d_tasks = nullptr; e->h_tasks = nullptr; e->d_ups = nullptr; e->h_ups = nullptr;
	const int cap = std::max(n, 1);
	HIPCHK(e, hipMalloc(&e->d_lvl[0], sizeof(float) * (size_t)cap * bw * bh));
	for (int l = 1; l <= e->nLevels; ++l)
		HIPCHK(e, hipMalloc(&e->d_lvl[l], sizeof(float) * (size_t)cap * 6 * lvlSize(bw, l) * lvlSize(bh, l)));
	HIPCHK(e, hipMalloc(&e->d_tasks, sizeof(PMTask) * 4 * cap));
	HIPCHK(e, hipHostMalloc(&e->h_tasks, sizeof(PMTask) * 4 * cap));
	HIPCHK(e, hipMalloc(&e->d_ups, sizeof(PMUpTask) * 4 * cap));
	HIPCHK(e, hipHostMalloc(&e->h_ups, sizeof(PMUpTask) * 4 * cap));
	e->batchCap = cap; e->batchW = bw; e->batchH = bh;
	return 0;
}

// tiled sweeps: snapshot storage for the batch (grow only)
static int ensureOld(pmhip_engine* e) {
	if (e->tileW <= 0 || e->tileH <= 0) return 0;
	if (e->oldCap >= e->batchCap && e->oldW >= e->batchW && e->oldH >= e->batchH) return 0;
	HIPCHK(e, hipStreamSynchronize(e->stream));
	for (int l = 0; l < 4; ++l) { if (e->d_old[l]) hipFree(e->d_old[l]); e->d_old[l] = nullptr; }
	for (int l = 0; l <= e->nLevels; ++l)
		HIPCHK(e, hipMalloc(&e->d_old[l], sizeof(float) * (size_t)e->batchCap * 5 * lvlSize(e->batchW, l) * lvlSize(e->batchH, l)));
	e->oldCap = e->batchCap; e->oldW = e->batchW; e->oldH = e->batchH;
	return 0;
}

static int buildPyramid(pmhip_engine* e) {
	if (!e->pyramidDirty) return 0;
	for (int l = 1; l <= e->nLevels; ++l) {
		const size_t n = (size_t)e->lw(l) * e->lh(l) * e->nImages;
		const int blocks = (int)std::min<size_t>((n + 255) / 256, 65535);
		// every level is resampled from the full-resolution image (ScaleDepthData(fullRes, 1/2^l), SceneDensify.cpp:654)
		hipLaunchKernelGGL(pm_area_kernel, dim3(blocks), dim3(256), 0, e->stream, e->d_img[0], e->d_img[l], e->w, e->h, e->lw(l), e->lh(l), 1 << l, e->nImages);
	}
	for (int l = 0; l <= e->nLevels; ++l) {
		const size_t n = (size_t)e->lw(l) * e->lh(l) * e->nImages;
		const int blocks = (int)std::min<size_t>((n + 255) / 256, 65535);
		hipLaunchKernelGGL(pm_skew_kernel, dim3(blocks), dim3(256), 0, e->stream, e->d_img[l], e->d_imgS[l], e->lw(l), e->lh(l), e->nImages);
		hipLaunchKernelGGL(pm_quad_kernel, dim3(blocks), dim3(256), 0, e->stream, e->d_img[l], e->d_imgQ[l], e->lw(l), e->lh(l), e->nImages);
	}
	HIPCHK(e, hipGetLastError());
	e->pyramidDirty = false;
	return 0;
}
// pyramids of the views that carry their own image size (source views only)
static int buildSidePyramids(pmhip_engine* e) {
	for (SceneView& v : e->views) {
		if (!v.sw || !v.sideDirty) continue;
		for (int l = 1; l <= e->nLevels; ++l) {
			const int lw = lvlSize(v.sw, l), lh = lvlSize(v.sh, l);
			if (lw < 1 || lh < 1 || !v.sImg[l]) break;      // a view too small for this level has no pyramid entry there (estimateBatch reports PMHIP_E_SIZE if the level is used)
			const size_t n = (size_t)lw * lh;
			hipLaunchKernelGGL(pm_area_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, e->stream, v.sImg[0], v.sImg[l], v.sw, v.sh, lw, lh, 1 << l, 1);
		}
		for (int l = 0; l <= e->nLevels; ++l) {
			const int lw = lvlSize(v.sw, l), lh = lvlSize(v.sh, l);
			if (lw < 1 || lh < 1 || !v.sImg[l]) break;
			const size_t n = (size_t)lw * lh;
			hipLaunchKernelGGL(pm_skew_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, e->stream, v.sImg[l], v.sImgS[l], lw, lh, 1);
			hipLaunchKernelGGL(pm_quad_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, e->stream, v.sImg[l], v.sImgQ[l], lw, lh, 1);
		}
		HIPCHK(e, hipGetLastError());
		v.sideDirty = false;
	}
	return 0;
}

static PMKParams makeKParams(const PMHipParams& p) {
	// DepthEstimator ctor, libs/MVS/DepthMap.cpp:397-406 (same float expressions)
	PMKParams k;
	k.smoothBonusDepth = 1.f - p.fRandomSmoothBonus;
	k.smoothBonusNormal = (1.f - p.fRandomSmoothBonus) * 0.96f;
	k.smoothSigmaDepth = -1.f / (2.f * (p.fRandomSmoothDepth * p.fRandomSmoothDepth));
	const float sn = PM_FD2R(p.fRandomSmoothNormal);
	k.smoothSigmaNormal = -1.f / (2.f * (sn * sn));
	k.thMagnitudeSq = p.fDescriptorMinMagnitudeThreshold > 0 ? p.fDescriptorMinMagnitudeThreshold * p.fDescriptorMinMagnitudeThreshold : -1.f;
	k.angle1Range = PM_FD2R(p.fRandomAngle1Range);
	k.angle2Range = PM_FD2R(p.fRandomAngle2Range);
	k.thConfSmall = p.fNCCThresholdKeep * 0.66f;
	k.thConfBig = p.fNCCThresholdKeep * 0.9f;
	k.thConfRand = p.fNCCThresholdKeep * 1.1f;
	k.thRobust = p.fNCCThresholdKeep * 4.f / 3.f;
	k.thKeep = p.fNCCThresholdKeep;
	k.geoWeight = p.fEstimationGeometricWeight;
	k.depthRatio = p.fRandomDepthRatio;
	k.nRandomIters = p.nRandomIters;
	return k;
}

// nv = next_pow2(source views of the batch); a pixel gets PM_INIT_LANES lanes (fewer if it has fewer views) and a lane scores nv / lanes views.  P: pixels of the level.
#ifndef PM_INIT_LANES
#define PM_INIT_LANES 2   // 100 views: 4 lanes 51.8, 2 lanes 52.1, 1 lane 51.8 Mpix/s (profiles/r06_call8); one view per lane (round 5): 51.1 (r06_call7)
#endif
template <bool GEO, int MODE, int G, int VPL>
static void launchInitAs(size_t P, int nT, hipStream_t s, const PMTask* t, const PMKParams& kp, uint32_t pass) {
	constexpr int PPB = PM_BLOCK / G;
	hipLaunchKernelGGL((pm_init_kernel<G, GEO, MODE, VPL>), dim3((unsigned)((P + PPB - 1) / PPB), nT), dim3(PM_BLOCK), 0, s, t, kp, pass);
}
template <bool GEO, int MODE>
static void launchInit(int nv, size_t P, int nT, hipStream_t s, const PMTask* t, const PMKParams& kp, uint32_t pass) {
	constexpr int L = PM_INIT_LANES;
	switch (nv) {
	case 1: launchInitAs<GEO, MODE, 1, 1>(P, nT, s, t, kp, pass); break;
	case 2: launchInitAs<GEO, MODE, 2, 1>(P, nT, s, t, kp, pass); break;
	case 4: launchInitAs<GEO, MODE, (L < 4 ? L : 4), 4 / (L < 4 ? L : 4)>(P, nT, s, t, kp, pass); break;
	case 8: launchInitAs<GEO, MODE, L, 8 / L>(P, nT, s, t, kp, pass); break;
	default: launchInitAs<GEO, MODE, 2 * L, 16 / (2 * L)>(P, nT, s, t, kp, pass); break;
	}
}
// Lanes per pixel for a batch whose views have at most maxSrc sources: G * VPL = next_pow2(maxSrc).  `lanes` (PMHipTuning::sweepLanes or the built-in
// default) caps G; VPL is what is left, limited to the instantiated mappings.
static void sweepMapping(int maxSrc, int lanes, int& G, int& VPL) {
	int NV = 1; while (NV < maxSrc) NV <<= 1;
	G = NV; VPL = 1;
	while (G > 4 && G > lanes && VPL < 4) { G >>= 1; VPL <<= 1; }   // a pixel gets at least a quad of lanes (one smoothness slot per lane)
	if (G < 4) G = 4;
	if (G == 8 && VPL > 2) { G <<= 1; VPL >>= 1; }   // (8,4) is not instantiated
}

// workgroups of a sweep launch whose workgroups hold ppw pixels each: the launch's pixels are numbered tile by tile, PMStep::len per tile
static unsigned stepBlocks(const PMStep& st, int ppw) { return (unsigned)(((long)st.len * st.ntx * st.nty + ppw - 1) / ppw); }
// (tiled sweeps are instantiated for the quad-buffer addressing only: a batch with source views of their own image size runs the reference's sweep)
template <bool GEO, bool BUF>
static bool launchSweep2(int G, int VPL, int nTasks, hipStream_t s, const PMTask* t, const PMKParams& kp, const PMStep& st, uint32_t pass) {
	const dim3 grid(stepBlocks(st, 64 / G), (unsigned)nTasks);
	const bool tiled = st.ntx * st.nty > 1;
#define PM_SWEEP2_CASE(g, vpl) case (g) * 16 + (vpl): \
		if constexpr (BUF) { if (tiled) { hipLaunchKernelGGL((pm_sweep2_kernel<g, vpl, GEO, BUF, true>), grid, dim3(64), 0, s, t, kp, st, pass); return true; } } \
		hipLaunchKernelGGL((pm_sweep2_kernel<g, vpl, GEO, BUF, false>), grid, dim3(64), 0, s, t, kp, st, pass); return true
	switch (G * 16 + VPL) {
	PM_SWEEP2_CASE(4, 1); PM_SWEEP2_CASE(8, 1); PM_SWEEP2_CASE(16, 1);
	PM_SWEEP2_CASE(4, 2); PM_SWEEP2_CASE(8, 2);
	PM_SWEEP2_CASE(4, 4);
	default: return false;
	}
#undef PM_SWEEP2_CASE
}

// the eight-wide speculative kernel (one wave per pixel; batches of one or two views, never with tiles): launch k of the reference's one-tile sweep as an anti-diagonal
template <bool GEO, bool BUF>
static void launchSweepWide(int nTasks, hipStream_t s, const PMTask* t, const PMKParams& kp, const PMStep& st, int lw, int lh, uint32_t pass) {
	const int dLo = 2 * PM_HW, dHi = (lw - 1 - PM_HW) + (lh - 1 - PM_HW);
	const int d = st.dir == 0 ? dLo + st.k : dHi - st.k;
	const int xlo = std::max(PM_HW, d - (lh - 1 - PM_HW)), xhi = std::min(lw - 1 - PM_HW, d - PM_HW);
	const int count = xhi - xlo + 1;
	if (count <= 0) return;
	hipLaunchKernelGGL((pm_sweep_wide_kernel<GEO, BUF>), dim3((unsigned)count, (unsigned)nTasks), dim3(64), 0, s, t, kp, st.dir, d, xlo, count, pass);
}
// the speculative kernel at 4 or 2 hypotheses per round (pm_wide_n.hip; PMHipTuning::wideHyps): 2 or 4 pixels per wave
template <bool GEO, bool BUF>
static void launchSweepWideN(int hyps, int nTasks, hipStream_t s, const PMTask* t, const PMKParams& kp, PMStep st, uint32_t pass) {
	const dim3 grid(stepBlocks(st, 8 / hyps), (unsigned)nTasks);
	if constexpr (BUF) if (st.ntx * st.nty > 1) {
		if (hyps == 4) hipLaunchKernelGGL((pm_sweep_widen_kernel<GEO, 4, BUF, true>), grid, dim3(64), 0, s, t, kp, st, pass);
		else hipLaunchKernelGGL((pm_sweep_widen_kernel<GEO, 2, BUF, true>), grid, dim3(64), 0, s, t, kp, st, pass);
		return;
	}
	if (hyps == 4) hipLaunchKernelGGL((pm_sweep_widen_kernel<GEO, 4, BUF, false>), grid, dim3(64), 0, s, t, kp, st, pass);
	else hipLaunchKernelGGL((pm_sweep_widen_kernel<GEO, 2, BUF, false>), grid, dim3(64), 0, s, t, kp, st, pass);
}
// one launch of a sweep for one view group with the kernel the batch calls for; false: the (lanes, views per lane) mapping is not instantiated
template <bool GEO, bool BUF>
static bool launchDiagonal(bool wide, int hyps, int G2, int V2, int nTasks, hipStream_t st, const PMTask* t, const PMKParams& kp, const PMStep& sp, int lw, int lh, uint32_t pass) {
	if (wide && hyps < 8) { launchSweepWideN<GEO, BUF>(hyps, nTasks, st, t, kp, sp, pass); return true; }
	if (wide) { launchSweepWide<GEO, BUF>(nTasks, st, t, kp, sp, lw, lh, pass); return true; }
	return launchSweep2<GEO, BUF>(G2, V2, nTasks, st, t, kp, sp, pass);
}

#ifdef PM_PROBES
// measurement builds (-DPM_PROBES, tools/): never part of the product library
static int g_probeRepeat = 1;
extern "C" int pmhip_probe_set(int key, int val) { if (key == 0) g_probeRepeat = val < 1 ? 1 : val; return 0; }
#endif
static size_t evBeginOn(pmhip_engine* e, int kind, hipStream_t st) {
	if (!e->statsOn) return 0;
	pmhip_engine::Ev ev; ev.kind = kind;
	hipEventCreate(&ev.a); hipEventCreate(&ev.b);
	hipEventRecord(ev.a, st);
	e->events.push_back(ev);
	return e->events.size() - 1;
}
static void evEndOn(pmhip_engine* e, size_t idx, hipStream_t st) {
	if (!e->statsOn) return;
	hipEventRecord(e->events[idx].b, st);
}
static void evBegin(pmhip_engine* e, int kind) { evBeginOn(e, kind, e->stream); }
static void evEnd(pmhip_engine* e) { if (e->statsOn) hipEventRecord(e->events.back().b, e->stream); }

// One DepthMapsData::EstimateDepthMap (SceneDensify.cpp:616-805) for each view of the batch, concurrently.
// ids: views of ONE size class (cw x ch: the scene's size, or the own size these views carry); estimateBatch below splits a batch into its classes.
static int estimateClass(pmhip_engine* e, const int32_t* ids, int nB, int cw, int ch, const PMHipParams& p, int nGeometricIter) {
	const unsigned iterBegin = nGeometricIter < 0 ? 0u : p.nEstimationIters + (unsigned)nGeometricIter;
	const unsigned iterEnd = nGeometricIter < 0 ? p.nEstimationIters : iterBegin + 1;
	const int S = nGeometricIter < 0 ? (int)p.nSubResolutionLevels : 0;
	if (lvlSize(cw, S) < 2 * PM_HW + 1 || lvlSize(ch, S) < 2 * PM_HW + 1) { e->err = "image too small for this many sub-resolution levels"; return PMHIP_E_SIZE; }
	const PMKParams kp = makeKParams(p);
	const bool geo = nGeometricIter >= 0;
	bool anyMask = false;
	for (unsigned char m : e->hasMask) anyMask = anyMask || m;
	const int nearestDepth = (e->maskMode < 0 ? anyMask : e->maskMode != 0) ? 1 : 0;
	if (anyMask && e->maskDirty) {
		for (int l = 1; l <= e->nLevels; ++l) {
			const size_t Pm = (size_t)e->lw(l) * e->lh(l);
			for (int i = 0; i < e->nImages; ++i) if (e->hasMask[i]) {
				const SceneView& mv = e->views[i];
				if (mv.sw) {   // a view with its own size: its own level masks
					const int mlw = lvlSize(mv.sw, l), mlh = lvlSize(mv.sh, l);
					if (mlw < 1 || mlh < 1 || !mv.oMask[l]) continue;
					hipLaunchKernelGGL(pm_mask_level_kernel, dim3((unsigned)std::min<size_t>(((size_t)mlw * mlh + 255) / 256, 4096)), dim3(256), 0, e->stream, mv.oMask[0], mv.oMask[l], mv.sw, mv.sh, mlw, mlh);
				} else
					hipLaunchKernelGGL(pm_mask_level_kernel, dim3((unsigned)std::min<size_t>((Pm + 255) / 256, 4096)), dim3(256), 0, e->stream,
						e->d_mask[0] + (size_t)e->w * e->h * i, e->d_mask[l] + Pm * i, e->w, e->h, e->lw(l), e->lh(l));
			}
		}
		HIPCHK(e, hipGetLastError());
		e->maskDirty = false;
	}
	int maxSrc = 0;
	bool buf = e->quadBuffer != 0;
	for (int b = 0; b < nB; ++b) {
		const int id = ids[b];
		if (id < 0 || id >= e->nImages || !e->views[id].set) { e->err = "view not set"; return PMHIP_E_ARG; }
		const SceneView& v = e->views[id];
		if (v.nNb < 1) { e->err = "view has no source views"; return PMHIP_E_ARG; }
		for (int k = 0; k < v.nNb; ++k) if (v.nb[k] < 0 || v.nb[k] >= e->nImages || !e->views[v.nb[k]].set) { e->err = "neighbour view not set"; return PMHIP_E_ARG; }
		maxSrc = std::max(maxSrc, v.nNb);
		for (int k = 0; k < v.nNb; ++k) if (e->views[v.nb[k]].sw) buf = false;   // a source image of its own size is not in the level's quad buffer
	}
	// the buffer path addresses a sample by a 32-bit entry index into the level's quad buffer (PMTask::qCount, PMSrcView::qBase): a level-0 buffer of 2^32 entries or more
	// (about 330 views of 3840x2160) goes through the views' own pointers instead
	if (e->skewPitch(0) * (size_t)e->nImages > 0xFFFFFFFFull) buf = false;
	int G = 1; while (G < maxSrc) G <<= 1;          // init kernel: one view per lane
	int SG = G, VPL = 1;                             // sweep kernel: (lanes per pixel, views per lane)
	sweepMapping(maxSrc, e->sweepLanes > 0 ? e->sweepLanes : ((nB >= PMHIP_LANES4_FROM || (e->tileW > 0 && e->tileH > 0)) && maxSrc > 4 ? 4 : 16), SG, VPL);
	// latency mode (one wave per pixel, pm_sweep_wide_kernel) for batches too small to fill the GPU with one wave per 64 / G pixels
	const bool wideBatch = nB <= e->wideMaxViews && maxSrc <= 8;
	const size_t P0 = (size_t)cw * ch;                      // this class's pixels; the scene arrays are indexed with the scene's own
	const size_t P0s = (size_t)e->w * e->h;
	// the staging buffers are reused by the next call (and by the next size class): make sure the previous copies are done
	HIPCHK(e, hipStreamSynchronize(e->stream));
	for (int l = S; l >= 0; --l) {
		const int lw = lvlSize(cw, l), lh = lvlSize(ch, l);
		const size_t Pl = (size_t)lw * lh;
		const int slw = e->lw(l), slh = e->lh(l);            // the scene's size at this level: source views that live in the scene arrays
		const size_t Pls = (size_t)slw * slh;
		PMTask* ht = e->h_tasks + (size_t)l * e->batchCap;
		PMUpTask* hu = e->h_ups + (size_t)l * e->batchCap;
		for (int b = 0; b < nB; ++b) {
			const int id = ids[b];
			const SceneView& v = e->views[id];
			PMTask& t = ht[b];
			memset(&t, 0, sizeof(t));
			if (l == 0) {
				t.depth = e->depthOf(id); t.normal = e->normalOf(id); t.conf = e->confOf(id);
				t.prior = (S > 0) ? e->d_lvl[0] + P0 * b : nullptr;
			} else {
				float* base = e->d_lvl[l] + Pl * 6 * b;
				t.depth = base; t.normal = base + Pl; t.conf = base + Pl * 4;
				t.prior = (l < S) ? base + Pl * 5 : nullptr;
			}
			if (e->tileW > 0 && e->tileH > 0) { float* ob = e->d_old[l] + Pl * 5 * b; t.depthOld = ob; t.normalOld = ob + Pl; t.confOld = ob + Pl * 4; }
			if (v.sw) { t.ref = v.sImg[l]; t.refS = v.sImgS[l]; }
			else { t.ref = e->d_img[l] + Pls * id; t.refS = e->d_imgS[l] + Pls * id; }
			t.qArr = e->d_imgQ[l]; t.qCount = (unsigned)(e->skewPitch(l) * (size_t)e->nImages);
			t.mask = (anyMask && e->hasMask[id]) ? (v.sw ? v.oMask[l] : e->d_mask[l] + Pls * id) : nullptr;
			t.w = lw; t.h = lh; t.nSrc = v.nNb;
			double K0[9];
			if (l == 0) memcpy(K0, v.K, sizeof(K0)); else scaleK(v.K, cw, ch, lw, lh, K0);
			inv33(K0, t.Hr);
			t.hrUpper = (t.Hr[1] == 0.0 && t.Hr[3] == 0.0 && t.Hr[6] == 0.0 && t.Hr[7] == 0.0) ? 1 : 0;
			t.fx = K0[0]; t.fy = K0[4]; t.cx = K0[2]; t.cy = K0[5];
			t.dMin = v.dMin; t.dMax = v.dMax; t.dMinSqr = sqrtf(v.dMin); t.dMaxSqr = sqrtf(v.dMax);
			t.k0 = p.seed; t.k1base = v.id * 0x9E3779B1u;
			double R0T[9]; transp33(v.R, R0T);
			double KR0[9]; mul33(K0, v.R, KR0);
			for (int k = 0; k < v.nNb; ++k) {
				const SceneView& sv = e->views[v.nb[k]];
				PMSrcView& s = t.src[k];
				double Kj[9];
				if (sv.sw) {
					// a source image of its own size: its own pyramid, its camera scaled from its own size (ScaleDepthData, SceneDensify.cpp:586-588)
					const int jw = lvlSize(sv.sw, l), jh = lvlSize(sv.sh, l);
					if (jw < 3 || jh < 3) { e->err = "source image too small for this many sub-resolution levels"; return PMHIP_E_SIZE; }
					s.img = sv.sImg[l]; s.imgQ = sv.sImgQ[l]; s.w = jw; s.h = jh;
					if (l == 0) memcpy(Kj, sv.K, sizeof(Kj)); else scaleK(sv.K, sv.sw, sv.sh, jw, jh, Kj);
				} else {
					s.img = e->d_img[l] + Pls * v.nb[k];
					s.imgQ = e->d_imgQ[l] + e->skewPitch(l) * v.nb[k];
					s.qBase = (unsigned)(e->skewPitch(l) * (size_t)v.nb[k]);
					s.w = slw; s.h = slh;
					if (l == 0) memcpy(Kj, sv.K, sizeof(Kj)); else scaleK(sv.K, e->w, e->h, slw, slh, Kj);
				}
				double KR[9], dC[3];
				mul33(Kj, sv.R, KR);
				mul33(KR, R0T, s.Hl);
				for (int i = 0; i < 3; ++i) dC[i] = v.C[i] - sv.C[i];
				mul31(KR, dC, s.Hm);
				s.depth = nullptr;
				if (geo) {
					// ViewData::Init geometric part, DepthMap.h:179-184.  cameraDepthMap is the neighbour's own camera when the map is the scene's
					// snapshot, or the camera stored with the map the caller installed (pmhip_scene_set_source_depth), whose size may differ too
					double tm[9], vv[3], RdT[9], iKd[9], t2[9], KdRd[9];
					const double* Kd = Kj; const double* Rd = sv.R; const double* Cd = sv.C;
					if (sv.sDepth) { s.depth = sv.sDepth; s.dw = sv.dw; s.dh = sv.dh; Kd = sv.Kd; Rd = sv.Rd; Cd = sv.Cd; }
					else if (sv.sw) { s.depth = sv.oSnap; s.dw = sv.sw; s.dh = sv.sh; }   // its own previous-round map (own size, own camera = Kj at level 0)
					else { s.depth = e->d_snap + P0s * v.nb[k]; s.dw = e->w; s.dh = e->h; }
					mul33(Kd, Rd, KdRd);
					mul33(KdRd, R0T, tm); for (int i = 0; i < 9; ++i) s.Tl[i] = (float)tm[i];
					for (int i = 0; i < 3; ++i) dC[i] = v.C[i] - Cd[i];
					mul31(KdRd, dC, vv); for (int i = 0; i < 3; ++i) s.Tm[i] = (float)vv[i];
					transp33(Rd, RdT); mul33(KR0, RdT, tm); invK(Kd, iKd); mul33(tm, iKd, t2);
					for (int i = 0; i < 9; ++i) s.Tr[i] = (float)t2[i];
					for (int i = 0; i < 3; ++i) dC[i] = Cd[i] - v.C[i];
					mul31(KR0, dC, vv); for (int i = 0; i < 3; ++i) s.Tn[i] = (float)vv[i];
				}
			}
			// level hand-off descriptors
			PMUpTask& u = hu[b];
			memset(&u, 0, sizeof(u));
			if (l == S && S > 0) { // coarsest: INTER_NEAREST of the caller's initial estimate
				u.sdepth = e->depthOf(id); u.snormal = e->normalOf(id); u.ddepth = t.depth; u.dnormal = t.normal; u.dprior = nullptr;
			} else if (l < S) {
				const size_t Pc = (size_t)lvlSize(cw, l + 1) * lvlSize(ch, l + 1);
				float* cb = e->d_lvl[l + 1] + Pc * 6 * b;
				u.sdepth = cb; u.snormal = cb + Pc; u.ddepth = t.depth; u.dnormal = t.normal; u.dprior = const_cast<float*>(t.prior);
			}
		}
		HIPCHK(e, hipMemcpyAsync(e->d_tasks + (size_t)l * e->batchCap, ht, sizeof(PMTask) * nB, hipMemcpyHostToDevice, e->stream));
		HIPCHK(e, hipMemcpyAsync(e->d_ups + (size_t)l * e->batchCap, hu, sizeof(PMUpTask) * nB, hipMemcpyHostToDevice, e->stream));
	}
	// ---- the pass as a sequence of steps that is the same for every view group: per level {hand-off, ScoreDepthMapTmp, sweeps of one launch per anti-diagonal}, EndDepthMapTmp.
	// A view group runs ALL of them on its own stream (views are independent; the steps of one view are not); the groups start together and meet again at the end of the
	// call.  Scheduling only: the maps cannot depend on it.
	// Measured and not kept (round 5, MI355X, full schedule at 1920x1080; all bit-identical):
	//  * group g + 1 starting behind group g, so that one group's short diagonals, coarse levels and init pass run under another group's long diagonals (commit c352a63,
	//    PMHipTuning::groupOffset): slower for every offset and batch size -- 100 views 47.8 -> 46.5 (5 % of the pass) -> 41.9 Mpix/s (40 %), 13 views 26.6 -> 25.4 -> 20.0
	//    (profiles/r05_call1_groups_*.log).  The late group also finishes late, and a group's launch chain runs slower beside the other's long diagonals than beside its ramps.
	//  * the groups waiting for each other before every sweep (round 4's fork / join per sweep): no difference (48.8 vs 48.8, profiles/r05_call4_ab_100.log).
	//  * the views of a group staggered along the pass, so that every launch mixes anti-diagonals, sweeps and levels and carries about the mean number of pixels
	//    (profiles/r05_view_stagger_experiment.diff): 100 views 46.2 -> 44.0 (5 steps per view) -> 37.7 (30) -> 35.0 Mpix/s (100), profiles/r05_call3_stagger_*.log.  A launch
	//    lasts one wave-visit at whatever fill, so evening out the fill buys nothing, while every step of the longer chain then costs the heavy kernel's visit.
	struct Step { int kind, l; unsigned iter; int k; };   // kind 0: level hand-off, 1: init pass, 2: launch k of sweep `iter`, 3: finalize, 4: snapshot of the maps before a tiled sweep
	std::vector<Step> steps;
	// a level's sweep geometry (PMStep): the reference's sweep is one tile = all pixels that take part; pmhip_set_sweep_tiles cuts them into tiles
	const bool tilesOn = e->tileW > 0 && e->tileH > 0;
	if (tilesOn && !buf) { e->err = "tiled sweeps (pmhip_set_sweep_tiles) address the level's quad buffer: not with source views of their own image size, PMHipTuning::quadBuffer = 2 or a level-0 buffer of 2^32 entries"; return PMHIP_E_ARG; }
	auto stepOf = [&](int l, int dir, int k) {
		const int vw = lvlSize(cw, l) - 2 * PM_HW, vh = lvlSize(ch, l) - 2 * PM_HW;
		PMStep sp; sp.dir = dir; sp.k = k;
		sp.tw = tilesOn ? std::min(e->tileW, vw) : vw; sp.th = tilesOn ? std::min(e->tileH, vh) : vh;
		sp.ntx = (vw + sp.tw - 1) / sp.tw; sp.nty = (vh + sp.th - 1) / sp.th;
		sp.len = std::max(0, std::min(std::min(k, sp.tw + sp.th - 2 - k), std::min(sp.tw, sp.th) - 1) + 1);
		return sp;
	};
	for (int l = S; l >= 0; --l) {
		if (S > 0) steps.push_back({0, l, 0u, 0});
		steps.push_back({1, l, 0u, 0});
		const PMStep g0s = stepOf(l, 0, 0);
		const int nDiag = g0s.tw + g0s.th - 1;
		for (unsigned iter = iterBegin; iter < iterEnd; ++iter) {
			if (g0s.ntx * g0s.nty > 1) steps.push_back({4, l, iter, 0});
			for (int k = 0; k < nDiag; ++k) steps.push_back({2, l, iter, k});
		}
	}
	steps.push_back({3, 0, 0u, 0});
	const long nSteps = (long)steps.size();
	const int NG = std::max(1, std::min(e->nGroups, nB));
	// group g's stream: the engine's own for a single group.  (Letting group 0 of several sweep on the engine's stream, or more than three groups, falls off a cliff:
	// 13 views 27 -> 15.6 Mpix/s, whatever GPU_MAX_HW_QUEUES says -- profiles/r04_call10_lanes_13.log.)
	auto gs = [&](int g) { return NG > 1 ? e->gstream[g] : e->stream; };
	auto g0 = [&](int g) { return (int)((long)nB * g / NG); };
	const size_t evWall = evBeginOn(e, 2, e->stream);
	if (NG > 1) {
		HIPCHK(e, hipEventRecord(e->forkEv, e->stream));
		for (int g = 0; g < NG; ++g) HIPCHK(e, hipStreamWaitEvent(gs(g), e->forkEv, 0));
	}
	float thFinal = p.fNCCThresholdKeep;   // EndDepthMapTmp: threshold x1.333 when geometric rounds will follow, SceneDensify.cpp:774-776
	if (nGeometricIter < 0 && p.nEstimationGeometricIters) thFinal *= 1.333f;
	size_t evSweep[16] = {}; bool evOpen[16] = {};
	size_t nLaunched = 0;
	const auto hostT0 = std::chrono::steady_clock::now();
	auto issue = [&](int g, const Step& sp) -> bool {
		const int l = sp.l, s0 = g0(g), nT = g0(g + 1) - s0;
		const int lw = lvlSize(cw, l), lh = lvlSize(ch, l);
		const size_t Pl = (size_t)lw * lh;
		const PMTask* dt = e->d_tasks + (size_t)l * e->batchCap + s0;
		const PMUpTask* du = e->d_ups + (size_t)l * e->batchCap + s0;
		hipStream_t st = gs(g);
		if (sp.kind != 2 && evOpen[g]) { evEndOn(e, evSweep[g], st); evOpen[g] = false; }
		switch (sp.kind) {
		case 0: {
			const int eb = (int)std::min<size_t>((Pl + 255) / 256, 4096);
			if (l == S) hipLaunchKernelGGL(pm_nearest_down_kernel, dim3(eb, nT), dim3(256), 0, st, du, cw, ch, lw, lh, 1 << S);
			else hipLaunchKernelGGL(pm_upsample_kernel, dim3(eb, nT), dim3(256), 0, st, du, lvlSize(cw, l + 1), lvlSize(ch, l + 1), lw, lh, nearestDepth);
			return true;
		}
		case 1: {
			// pass A: ScoreDepthMapTmp, row-major pixels.  (Round 4, 24 views resident: the pass was latency-bound, and the same evaluation on anti-diagonals with the sweep's optimistic
			// quad rows was 9 % SLOWER -- the maps are row-major, and a wave that walks a diagonal reads and writes them one cache line per lane:
			// profiles/r04_call14_diagonal_init_kernel_stats.csv.  Round 6, 100 views resident: bound by VALU issue (0.89 busy); row-major pixels WITH the optimistic quad rows: +0.8 %.)
			const uint32_t passInit = (uint32_t)l * 64u + 32u + (geo ? 16u + (uint32_t)nGeometricIter : 0u);
			const size_t ev = evBeginOn(e, 1, st);
			// (optimistic rows from the level's quad buffer; the guarded rows from the row-major images for batches that read source views outside that buffer)
			if (buf && PM_INIT_MODE == 2) { if (geo) launchInit<true, 2>(G, Pl, nT, st, dt, kp, passInit); else launchInit<false, 2>(G, Pl, nT, st, dt, kp, passInit); }
			else { if (geo) launchInit<true, 0>(G, Pl, nT, st, dt, kp, passInit); else launchInit<false, 0>(G, Pl, nT, st, dt, kp, passInit); }
			evEndOn(e, ev, st);
			if (e->statsOn && g == 0) e->stats.initLaunches += 1;
			return true;
		}
		case 2: {
			// pass B: one launch per anti-diagonal.  The kernels compute the same bits, so the choice is per launch: the speculative kernels (more lanes per pixel, shorter
			// dependent chain) for batches and for diagonals too small to fill the GPU with 64 / G pixels per wave
			const int dir = (int)(sp.iter % 2u);
			const uint32_t pass = (uint32_t)l * 64u + sp.iter;
			const PMStep ps = stepOf(l, dir, sp.k);
			// pixels of this launch: a full tile's k-th anti-diagonal holds min(k, tw - 1, th - 1, tw + th - 2 - k) + 1 of them (the tiles at the right and bottom border fewer)
			const int perTile = ps.len;
			if (perTile <= 0) return true;
			const bool tiled = ps.ntx * ps.nty > 1;
			if (!evOpen[g]) { evSweep[g] = evBeginOn(e, 0, st); evOpen[g] = e->statsOn; }
			const long npx = (long)perTile * ps.ntx * ps.nty * nT;
			// larger batches: launches of at most widePixels pixels AND at most PMHIP_WIDE_DIAGONAL pixels per view go to the two-wide kernel (round 6: 100 views in two groups are
			// best at 17-20 000 pixels, 50 views at <= 10 000 -- the same ~400 pixels of diagonal per view: profiles/r06_call18, r06_call19)
			const long wpx = tiled ? (long)e->widePixels : std::min<long>(e->widePixels, (long)PMHIP_WIDE_DIAGONAL * nT);
			const bool wide = (wideBatch && !(tiled && npx > e->widePixels)) || (maxSrc <= 8 && npx <= wpx);
			int hyps = (wideBatch && e->wideHyps > 0) ? e->wideHyps : ((nB <= 2 || npx <= e->wide8Pixels) ? 8 : 2);
			if (tiled && hyps == 8) hyps = 2;   // (the eight-wide kernel walks whole anti-diagonals of the map)
			++nLaunched;
#ifdef PM_PROBES
			for (int r = 1; r < g_probeRepeat; ++r)   // (measurement builds only) the same diagonal again, back to back: what does a launch find in the caches its predecessor filled?
				geo ? (buf ? launchDiagonal<true, true>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass) : launchDiagonal<true, false>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass))
				    : (buf ? launchDiagonal<false, true>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass) : launchDiagonal<false, false>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass));
#endif
			return geo ? (buf ? launchDiagonal<true, true>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass) : launchDiagonal<true, false>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass))
			           : (buf ? launchDiagonal<false, true>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass) : launchDiagonal<false, false>(wide, hyps, SG, VPL, nT, st, dt, kp, ps, lw, lh, pass));
		}
		case 4: {
			// tiled sweeps: the maps as this sweep finds them, for the reads across tile borders
			hipLaunchKernelGGL(pm_snapshot_kernel, dim3((unsigned)std::min<size_t>((Pl + 255) / 256, 2048), nT), dim3(256), 0, st, dt, Pl);
			return true;
		}
		default:
			hipLaunchKernelGGL(pm_finalize_kernel, dim3((unsigned)std::min<size_t>((P0 + 255) / 256, 4096), nT), dim3(256), 0, st, e->d_tasks + s0, thFinal);
			return true;
		}
	};
	// the host feeds the groups' streams in turn, step by step
	for (long i = 0; i < nSteps; ++i)
		for (int g = 0; g < NG; ++g)
			if (!issue(g, steps[i])) { e->err = "sweep kernel: (lanes per pixel, views per lane) mapping not instantiated"; return PMHIP_E_ARG; }
	if (NG > 1) for (int g = 0; g < NG; ++g) { HIPCHK(e, hipEventRecord(e->joinEv[g], gs(g))); HIPCHK(e, hipStreamWaitEvent(e->stream, e->joinEv[g], 0)); }
	evEndOn(e, evWall, e->stream);
	HIPCHK(e, hipGetLastError());
	if (e->statsOn) {
		e->stats.sweepLaunches += nLaunched;
		e->stats.sweepHostMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hostT0).count();
		// algorithmic bytes of the sweeps, SURVEY.md 8(d): P_l * [4(1+N) + 20 + 20 + 4[prior] + 4N[geo]] per view and sweep
		for (int l = S; l >= 0; --l) {
			const size_t Pl = (size_t)lvlSize(cw, l) * lvlSize(ch, l);
			double bytes = 0;
			for (int b = 0; b < nB; ++b) { const int N = e->views[ids[b]].nNb; bytes += (double)Pl * (4.0 * (1 + N) + 40.0 + (l < S ? 4.0 : 0.0) + (geo ? 4.0 * N : 0.0)); }
			e->stats.sweepBytes += bytes * (iterEnd - iterBegin);
			e->stats.sweepPixels += (uint64_t)Pl * nB * (iterEnd - iterBegin);
		}
	}
	for (int b = 0; b < nB; ++b) e->views[ids[b]].hasMaps = true;
	return 0;
}


// One DepthMapsData::EstimateDepthMap for each view of the batch.  The reference sizes every depth map on its own image (DepthMapsData::InitViews, SceneDensify.cpp:306-459):
// the views of a batch are grouped by size -- the scene's, or the one a view carries (pmhip_scene_set_view_sized) -- and each size class sweeps on its own.
static int estimateBatch(pmhip_engine* e, const int32_t* ids, int nB, const PMHipParams& p, int nGeometricIter) {
	if (nB <= 0) return 0;
	if (nGeometricIter >= 0 && !e->geom) { e->err = "geometric round requested but engine initialised with bGeomConsistency == 0"; return PMHIP_E_STATE; }
	const int S = nGeometricIter < 0 ? (int)p.nSubResolutionLevels : 0;
	if (S > e->nLevels || S > 3) { e->err = "nSubResolutionLevels exceeds the pyramid allocated by pmhip_scene_create"; return PMHIP_E_ARG; }
	int rc = buildPyramid(e); if (rc) return rc;
	rc = buildSidePyramids(e); if (rc) return rc;
	std::vector<std::pair<int, int>> sizes; std::vector<std::vector<int32_t>> members;   // size classes in order of first appearance
	int maxN = 0, maxW = 0, maxH = 0;
	for (int b = 0; b < nB; ++b) {
		const int id = ids[b];
		if (id < 0 || id >= e->nImages || !e->views[id].set) { e->err = "view not set"; return PMHIP_E_ARG; }
		const std::pair<int, int> sz(e->vw(id), e->vh(id));
		size_t c = 0; while (c < sizes.size() && sizes[c] != sz) ++c;
		if (c == sizes.size()) { sizes.push_back(sz); members.emplace_back(); }
		members[c].push_back(id);
		maxN = std::max(maxN, (int)members[c].size()); maxW = std::max(maxW, sz.first); maxH = std::max(maxH, sz.second);
	}
	rc = ensureBatch(e, maxN, maxW, maxH); if (rc) return rc;
	rc = ensureOld(e); if (rc) return rc;
	for (size_t c = 0; c < sizes.size(); ++c) {
		rc = estimateClass(e, members[c].data(), (int)members[c].size(), sizes[c].first, sizes[c].second, p, nGeometricIter);
		if (rc) return rc;
	}
	return 0;
}

static int collectStats(pmhip_engine* e) {
	if (e->events.empty()) return 0;
	HIPCHK(e, hipStreamSynchronize(e->stream));
	for (auto& ev : e->events) {
		float ms = 0; hipEventElapsedTime(&ms, ev.a, ev.b);
		if (ev.kind == 0) e->stats.sweepMs += ms; else if (ev.kind == 2) e->stats.sweepWallMs += ms; else e->stats.initMs += ms;
		hipEventDestroy(ev.a); hipEventDestroy(ev.b);
	}
	e->events.clear();
	return 0;
}

extern "C" {

int pmhip_default_params(PMHipParams* p) {
	if (!p) return PMHIP_E_ARG;
	// libs/MVS/DepthMap.cpp:69-114
	p->nSubResolutionLevels = 2; p->nEstimationIters = 3; p->nEstimationGeometricIters = 2; p->nRandomIters = 6;
	p->fEstimationGeometricWeight = 0.1f; p->fRandomDepthRatio = 0.003f; p->fRandomAngle1Range = 16.f; p->fRandomAngle2Range = 10.f;
	p->fRandomSmoothDepth = 0.02f; p->fRandomSmoothNormal = 13.f; p->fRandomSmoothBonus = 0.93f; p->fNCCThresholdKeep = 0.9f;
	p->fDescriptorMinMagnitudeThreshold = 0.02f; p->seed = 0;
	return 0;
}

int pmhip_create(int device, pmhip_engine** out) {
	if (!out) return PMHIP_E_ARG;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PMHIP_E_NODEVICE;
	if (device < 0) device = 0; // "-1 = best device" (DensifyPointCloud.cpp:112): one GPU per process here
	if (device >= n) return PMHIP_E_NODEVICE;
	pmhip_engine* e = new pmhip_engine();
	e->device = device;
	if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return PMHIP_E_HIP; }
	e->nGroups = PMHIP_DEFAULT_GROUPS;   // (every mapping choice is PMHipTuning's: the library reads no environment)
	for (int g = 0; g < e->nGroups; ++g)
		if (hipStreamCreateWithFlags(&e->gstream[g], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->joinEv[g], hipEventDisableTiming) != hipSuccess) { delete e; return PMHIP_E_HIP; }
	if (hipEventCreateWithFlags(&e->forkEv, hipEventDisableTiming) != hipSuccess) { delete e; return PMHIP_E_HIP; }
	*out = e;
	return 0;
}

void pmhip_destroy(pmhip_engine* e) {
	if (!e) return;
	hipSetDevice(e->device);
	if (e->stream) hipStreamSynchronize(e->stream);
	for (auto& ev : e->events) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
	freeScene(e);
	for (int g = 0; g < 16; ++g) { if (e->gstream[g]) hipStreamDestroy(e->gstream[g]); if (e->joinEv[g]) hipEventDestroy(e->joinEv[g]); }
	if (e->forkEv) hipEventDestroy(e->forkEv);
	if (e->stream) hipStreamDestroy(e->stream);
	delete e;
}

int pmhip_init(pmhip_engine* e, int bGeomConsistency) {
	if (!e) return PMHIP_E_ARG;
	e->inited = true; e->geom = bGeomConsistency != 0;
	return 0;
}

int pmhip_release(pmhip_engine* e) {
	if (!e) return PMHIP_E_ARG;
	hipSetDevice(e->device);
	hipStreamSynchronize(e->stream);
	freeScene(e);
	e->inited = false;
	return 0;
}

const char* pmhip_last_error(pmhip_engine* e) { return e ? e->err.c_str() : "null engine"; }

int pmhip_scene_create(pmhip_engine* e, int nImages, int w, int h, int nLevels) {
	if (!e || nImages < 2 || w < 2 * PM_HW + 1 || h < 2 * PM_HW + 1 || nLevels < 0 || nLevels > 3) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	freeScene(e);
	e->nImages = nImages; e->w = w; e->h = h; e->nLevels = nLevels;
	const size_t P0 = (size_t)w * h;
	for (int l = 0; l <= nLevels; ++l) {
		HIPCHK(e, hipMalloc(&e->d_img[l], sizeof(float) * (size_t)e->lw(l) * e->lh(l) * nImages));
		HIPCHK(e, hipMalloc(&e->d_imgS[l], sizeof(float) * (size_t)e->lw(l) * e->lh(l) * nImages));
		HIPCHK(e, hipMalloc(&e->d_imgQ[l], sizeof(float4) * e->skewPitch(l) * nImages));
	}
	HIPCHK(e, hipMalloc(&e->d_depth, sizeof(float) * P0 * nImages));
	HIPCHK(e, hipMalloc(&e->d_normal, sizeof(float) * P0 * 3 * nImages));
	HIPCHK(e, hipMalloc(&e->d_conf, sizeof(float) * P0 * nImages));
	HIPCHK(e, hipMalloc(&e->d_snap, sizeof(float) * P0 * nImages));
	HIPCHK(e, hipMemsetAsync(e->d_depth, 0, sizeof(float) * P0 * nImages, e->stream));
	HIPCHK(e, hipMemsetAsync(e->d_normal, 0, sizeof(float) * P0 * 3 * nImages, e->stream));
	HIPCHK(e, hipMemsetAsync(e->d_conf, 0, sizeof(float) * P0 * nImages, e->stream));
	HIPCHK(e, hipMemsetAsync(e->d_snap, 0, sizeof(float) * P0 * nImages, e->stream));
	e->views.assign(nImages, SceneView());
	for (int i = 0; i < nImages; ++i) e->views[i].id = (uint32_t)i;
	e->pyramidDirty = true;
	return 0;
}

int pmhip_scene_set_view(pmhip_engine* e, int idx, const float* gray, int onDevice, const double K[9], const double R[9], const double C[3],
		float dMin, float dMax, const int32_t* neighbors, int nNeighbors) {
	if (!e || idx < 0 || idx >= e->nImages || !K || !R || !C || nNeighbors < 0 || nNeighbors > PM_MAX_SRC) return PMHIP_E_ARG;
	if (!(dMin > 0) || !(dMin < dMax)) { e->err = "need 0 < dMin < dMax"; return PMHIP_E_ARG; }
	HIPCHK(e, hipSetDevice(e->device));
	SceneView& v = e->views[idx];
	memcpy(v.K, K, 72); memcpy(v.R, R, 72); memcpy(v.C, C, 24);
	v.dMin = dMin; v.dMax = dMax; v.nNb = nNeighbors;
	for (int k = 0; k < nNeighbors; ++k) v.nb[k] = neighbors[k];
	v.set = true;
	if (gray) {
		if (v.sw) {   // the view goes back to the scene's size: its own pyramid is not needed any more
			HIPCHK(e, hipStreamSynchronize(e->stream));
			for (int l = 0; l < 4; ++l) { if (v.sImg[l]) hipFree(v.sImg[l]); if (v.sImgS[l]) hipFree(v.sImgS[l]); if (v.sImgQ[l]) hipFree(v.sImgQ[l]); v.sImg[l] = v.sImgS[l] = nullptr; v.sImgQ[l] = nullptr; }
			if (v.oDepth) hipFree(v.oDepth); if (v.oNormal) hipFree(v.oNormal); if (v.oConf) hipFree(v.oConf); if (v.oSnap) hipFree(v.oSnap);
			if (v.oFDepth) hipFree(v.oFDepth); if (v.oFConf) hipFree(v.oFConf); if (v.oBgr) hipFree(v.oBgr);
			for (int l = 0; l < 4; ++l) { if (v.oMask[l]) hipFree(v.oMask[l]); v.oMask[l] = nullptr; }
			if (!e->hasMask.empty()) e->hasMask[idx] = 0;
			if (!e->fu.hasBgr.empty()) e->fu.hasBgr[idx] = 0;      // (its colour image went with the side storage)
			v.oDepth = v.oNormal = v.oConf = v.oSnap = v.oFDepth = v.oFConf = nullptr; v.oBgr = nullptr;
			v.sw = v.sh = 0; v.sideDirty = false; v.hasMaps = false;
			// its maps in the scene arrays: "unset", like after pmhip_scene_create (whatever the slot held before the view carried its own size is not an estimate of this image)
			const size_t Ps = (size_t)e->w * e->h;
			HIPCHK(e, hipMemsetAsync(e->d_depth + Ps * idx, 0, sizeof(float) * Ps, e->stream)); HIPCHK(e, hipMemsetAsync(e->d_normal + Ps * 3 * idx, 0, sizeof(float) * Ps * 3, e->stream));
			HIPCHK(e, hipMemsetAsync(e->d_conf + Ps * idx, 0, sizeof(float) * Ps, e->stream)); HIPCHK(e, hipMemsetAsync(e->d_snap + Ps * idx, 0, sizeof(float) * Ps, e->stream));
		}
		const size_t P0 = (size_t)e->w * e->h;
		HIPCHK(e, hipMemcpyAsync(e->d_img[0] + P0 * idx, gray, sizeof(float) * P0, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
		if (!onDevice) HIPCHK(e, hipStreamSynchronize(e->stream)); // caller may free the host buffer
		e->pyramidDirty = true;
	}
	return 0;
}

int pmhip_scene_images_updated(pmhip_engine* e) { if (!e) return PMHIP_E_ARG; e->pyramidDirty = true; return 0; }
uint32_t pmhip_abi_version(void) { return PMHIP_ABI_VERSION; }
int pmhip_get_tuning(pmhip_engine* e, PMHipTuning* out) {
	if (!e || !out) return PMHIP_E_ARG;
	out->viewGroups = e->nGroups; out->wideMaxViews = e->wideMaxViews > 0 ? e->wideMaxViews : -1; out->wideHyps = e->wideHyps > 0 ? e->wideHyps : -1;
	out->sweepLanes = e->sweepLanes > 0 ? e->sweepLanes : -1; out->quadBuffer = e->quadBuffer ? 1 : 2;
	out->widePixels = e->widePixels > 0 ? e->widePixels : -1; out->wide8Pixels = e->wide8Pixels > 0 ? e->wide8Pixels : -1;
	out->reserved0 = 0;
	return 0;
}
int pmhip_set_tuning(pmhip_engine* e, const PMHipTuning* t) {
	if (!e || !t) return PMHIP_E_ARG;
	if (t->viewGroups < 0 || t->viewGroups > 16 || (t->wideHyps > 0 && t->wideHyps != 8 && t->wideHyps != 4 && t->wideHyps != 2) ||
	    (t->sweepLanes > 0 && t->sweepLanes != 4 && t->sweepLanes != 8 && t->sweepLanes != 16) || t->quadBuffer < 0 || t->quadBuffer > 2) { e->err = "pmhip_set_tuning: value out of range"; return PMHIP_E_ARG; }
	HIPCHK(e, hipSetDevice(e->device));
	if (t->viewGroups > 0) {
		for (int g = e->nGroups; g < t->viewGroups; ++g) if (!e->gstream[g]) {   // streams of the additional view groups
			if (hipStreamCreateWithFlags(&e->gstream[g], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->joinEv[g], hipEventDisableTiming) != hipSuccess) { e->err = "pmhip_set_tuning: stream"; return PMHIP_E_HIP; }
		}
		e->nGroups = t->viewGroups;
	}
	if (t->wideMaxViews != 0) { e->wideMaxViews = t->wideMaxViews < 0 ? 0 : t->wideMaxViews; if (t->wideMaxViews < 0) e->widePixels = e->wide8Pixels = 0; }   // "never" = no speculative kernels at all, unless this call sets the per-launch rule
	if (t->wideHyps != 0) e->wideHyps = t->wideHyps < 0 ? 0 : t->wideHyps;
	if (t->sweepLanes != 0) e->sweepLanes = t->sweepLanes < 0 ? 0 : t->sweepLanes;
	if (t->quadBuffer != 0) e->quadBuffer = t->quadBuffer == 1;
	if (t->widePixels != 0) e->widePixels = t->widePixels < 0 ? 0 : t->widePixels;
	if (t->wide8Pixels != 0) e->wide8Pixels = t->wide8Pixels < 0 ? 0 : t->wide8Pixels;
	return 0;
}
int pmhip_set_sweep_tiles(pmhip_engine* e, int tileW, int tileH) {
	if (!e || tileW < 0 || tileH < 0 || (tileW > 0) != (tileH > 0) || (tileW > 0 && (tileW < 8 || tileH < 8))) { if (e) e->err = "pmhip_set_sweep_tiles: 0 x 0 (off) or at least 8 x 8"; return PMHIP_E_ARG; }
	HIPCHK(e, hipSetDevice(e->device)); HIPCHK(e, hipStreamSynchronize(e->stream));
	e->tileW = tileW; e->tileH = tileH;
	return 0;
}
int pmhip_scene_set_view_id(pmhip_engine* e, int idx, uint32_t viewID) {
	if (!e || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	e->views[idx].id = viewID;
	return 0;
}
int pmhip_scene_maps_updated(pmhip_engine* e, int firstIdx, int count) {
	if (!e || firstIdx < 0 || count < 0 || firstIdx + count > e->nImages) return PMHIP_E_ARG;
	for (int i = firstIdx; i < firstIdx + count; ++i) e->views[i].hasMaps = true;
	return 0;
}


// A source view whose image has its own size (see SceneView::sw): same as pmhip_scene_set_view otherwise.
int pmhip_scene_set_view_sized(pmhip_engine* e, int idx, const float* gray, int w, int h, int onDevice, const double K[9], const double R[9], const double C[3],
		float dMin, float dMax, const int32_t* neighbors, int nNeighbors) {
	if (!e || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	if (w == e->w && h == e->h) return pmhip_scene_set_view(e, idx, gray, onDevice, K, R, C, dMin, dMax, neighbors, nNeighbors);
	if (!gray || w < 3 || h < 3) { e->err = "a view with its own size needs its image and at least 3 x 3 pixels"; return PMHIP_E_ARG; }
	int rc = pmhip_scene_set_view(e, idx, nullptr, 0, K, R, C, dMin, dMax, neighbors, nNeighbors);
	if (rc) return rc;
	SceneView& v = e->views[idx];
	HIPCHK(e, hipStreamSynchronize(e->stream));
	if (v.sw != w || v.sh != h) {
		float* keep = v.sDepth; v.sDepth = nullptr; const int kdw = v.dw, kdh = v.dh;
		freeSide(v);
		v.sDepth = keep; v.dw = kdw; v.dh = kdh;
		for (int l = 0; l <= e->nLevels; ++l) {
			const int lw = lvlSize(w, l), lh = lvlSize(h, l);
			if (lw < 1 || lh < 1) break;
			HIPCHK(e, hipMalloc(&v.sImg[l], sizeof(float) * (size_t)lw * lh));
			HIPCHK(e, hipMalloc(&v.sImgS[l], sizeof(float) * (size_t)lw * lh));
			HIPCHK(e, hipMalloc(&v.sImgQ[l], sizeof(float4) * (size_t)(lw + lh - 1) * lh));
		}
		// its own maps (DepthData::depthMap / normalMap / confMap of its own size), "unset" like the scene's after pmhip_scene_create
		const size_t P = (size_t)w * h;
		HIPCHK(e, hipMalloc(&v.oDepth, sizeof(float) * P)); HIPCHK(e, hipMalloc(&v.oNormal, sizeof(float) * P * 3));
		HIPCHK(e, hipMalloc(&v.oConf, sizeof(float) * P)); HIPCHK(e, hipMalloc(&v.oSnap, sizeof(float) * P));
		HIPCHK(e, hipMemsetAsync(v.oDepth, 0, sizeof(float) * P, e->stream)); HIPCHK(e, hipMemsetAsync(v.oNormal, 0, sizeof(float) * P * 3, e->stream));
		HIPCHK(e, hipMemsetAsync(v.oConf, 0, sizeof(float) * P, e->stream)); HIPCHK(e, hipMemsetAsync(v.oSnap, 0, sizeof(float) * P, e->stream));
		v.sw = w; v.sh = h; v.hasMaps = false;
		if (!e->hasMask.empty()) e->hasMask[idx] = 0;     // (a mask of the previous size went with the side storage)
	}
	HIPCHK(e, hipMemcpyAsync(v.sImg[0], gray, sizeof(float) * (size_t)w * h, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
	if (!onDevice) HIPCHK(e, hipStreamSynchronize(e->stream));
	v.sideDirty = true;
	return 0;
}

// Known depth-map of view idx for the geometric rounds in which it is a SOURCE view, with the camera it was stored with and its own size
// (DepthData::ViewData::depthMap + cameraDepthMap, SceneDensify.cpp:378-393).  While installed it is read instead of the scene's snapshot of that
// view; depth == NULL removes it.
int pmhip_scene_set_source_depth(pmhip_engine* e, int idx, const float* depth, int dw, int dh, const double Kd[9], const double Rd[9], const double Cd[3]) {
	if (!e || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	SceneView& v = e->views[idx];
	HIPCHK(e, hipStreamSynchronize(e->stream));
	if (!depth) { if (v.sDepth) hipFree(v.sDepth); v.sDepth = nullptr; v.dw = v.dh = 0; return 0; }
	if (dw < 3 || dh < 3 || !Kd || !Rd || !Cd) return PMHIP_E_ARG;
	if (v.dw != dw || v.dh != dh || !v.sDepth) {
		if (v.sDepth) hipFree(v.sDepth);
		v.sDepth = nullptr;
		HIPCHK(e, hipMalloc(&v.sDepth, sizeof(float) * (size_t)dw * dh));
		v.dw = dw; v.dh = dh;
	}
	HIPCHK(e, hipMemcpyAsync(v.sDepth, depth, sizeof(float) * (size_t)dw * dh, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	memcpy(v.Kd, Kd, 72); memcpy(v.Rd, Rd, 72); memcpy(v.Cd, Cd, 24);
	return 0;
}

int pmhip_scene_estimate(pmhip_engine* e, const int32_t* viewIds, int nViews, const PMHipParams* p, int nGeometricIter, int sync) {
	if (!e || !viewIds || !p || nViews < 0) return PMHIP_E_ARG;
	if (!e->inited) { e->err = "pmhip_init not called"; return PMHIP_E_STATE; }
	HIPCHK(e, hipSetDevice(e->device));
	int rc = estimateBatch(e, viewIds, nViews, *p, nGeometricIter);
	if (rc) return rc;
	if (sync) HIPCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int pmhip_scene_commit_round(pmhip_engine* e) {
	if (!e || !e->d_snap) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	HIPCHK(e, hipMemcpyAsync(e->d_snap, e->d_depth, sizeof(float) * (size_t)e->w * e->h * e->nImages, hipMemcpyDeviceToDevice, e->stream));
	for (const SceneView& v : e->views) if (v.sw) HIPCHK(e, hipMemcpyAsync(v.oSnap, v.oDepth, sizeof(float) * (size_t)v.sw * v.sh, hipMemcpyDeviceToDevice, e->stream));
	return 0;
}

int pmhip_scene_reset_view(pmhip_engine* e, int idx) {
	if (!e || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t P0 = e->vpix(idx);
	HIPCHK(e, hipMemsetAsync(e->depthOf(idx), 0, sizeof(float) * P0, e->stream));
	HIPCHK(e, hipMemsetAsync(e->normalOf(idx), 0, sizeof(float) * P0 * 3, e->stream));
	HIPCHK(e, hipMemsetAsync(e->confOf(idx), 0, sizeof(float) * P0, e->stream));
	e->views[idx].hasMaps = false;
	return 0;
}

int pmhip_scene_set_maps(pmhip_engine* e, int idx, const float* depth, const float* normal) {
	if (!e || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t P0 = e->vpix(idx);
	if (depth) HIPCHK(e, hipMemcpyAsync(e->depthOf(idx), depth, sizeof(float) * P0, hipMemcpyHostToDevice, e->stream));
	if (normal) HIPCHK(e, hipMemcpyAsync(e->normalOf(idx), normal, sizeof(float) * P0 * 3, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	if (depth) e->views[idx].hasMaps = true;
	return 0;
}

int pmhip_scene_set_mask(pmhip_engine* e, int idx, const unsigned char* mask) {
	if (!e || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t P0 = (size_t)e->w * e->h;
	if (e->hasMask.empty()) e->hasMask.assign(e->nImages, 0);
	if (!mask) { e->hasMask[idx] = 0; return 0; }
	if (e->views[idx].sw) {                       // a view with its own size keeps its own masks
		SceneView& v = e->views[idx];
		for (int l = 0; l <= e->nLevels; ++l) {
			const int mlw = lvlSize(v.sw, l), mlh = lvlSize(v.sh, l);
			if (mlw < 1 || mlh < 1) break;
			if (!v.oMask[l]) HIPCHK(e, hipMalloc(&v.oMask[l], (size_t)mlw * mlh));
		}
		HIPCHK(e, hipMemcpyAsync(v.oMask[0], mask, (size_t)v.sw * v.sh, hipMemcpyHostToDevice, e->stream));
		HIPCHK(e, hipStreamSynchronize(e->stream));
		e->hasMask[idx] = 1; e->maskDirty = true;
		return 0;
	}
	if (!e->d_mask[0]) for (int l = 0; l <= e->nLevels; ++l) HIPCHK(e, hipMalloc(&e->d_mask[l], (size_t)e->lw(l) * e->lh(l) * e->nImages));
	HIPCHK(e, hipMemcpyAsync(e->d_mask[0] + P0 * idx, mask, P0, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	e->hasMask[idx] = 1; e->maskDirty = true;
	return 0;
}
int pmhip_scene_set_mask_mode(pmhip_engine* e, int mode) {
	if (!e || mode < -1 || mode > 1) return PMHIP_E_ARG;
	e->maskMode = mode;
	return 0;
}

int pmhip_scene_set_conf(pmhip_engine* e, int idx, const float* conf) {
	if (!e || !conf || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t P0 = e->vpix(idx);
	HIPCHK(e, hipMemcpyAsync(e->confOf(idx), conf, sizeof(float) * P0, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int pmhip_scene_get_maps(pmhip_engine* e, int idx, float* depth, float* normal, float* conf) {
	if (!e || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t P0 = e->vpix(idx);   // (a view with its own size returns maps of that size)
	if (depth) HIPCHK(e, hipMemcpyAsync(depth, e->depthOf(idx), sizeof(float) * P0, hipMemcpyDeviceToHost, e->stream));
	if (normal) HIPCHK(e, hipMemcpyAsync(normal, e->normalOf(idx), sizeof(float) * P0 * 3, hipMemcpyDeviceToHost, e->stream));
	if (conf) HIPCHK(e, hipMemcpyAsync(conf, e->confOf(idx), sizeof(float) * P0, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

void* pmhip_scene_device_ptr(pmhip_engine* e, int what, int idx) {
	if (!e || idx < 0 || idx >= e->nImages) return nullptr;
	const size_t P0 = (size_t)e->w * e->h;
	switch (what) {
	case 0: return e->views[idx].sw ? e->views[idx].sImg[0] : e->d_img[0] + P0 * idx;
	case 1: return e->depthOf(idx);
	case 2: return e->normalOf(idx);
	case 3: return e->confOf(idx);
	case 4: return e->snapOf(idx);
	default: return nullptr;
	}
}

// DepthMapsData::FilterDepthMap for each view of viewIds against its first <= 8 neighbours (Scene::DenseReconstructionFilter,
// SceneDensify.cpp:2136-2170).  Results are staged; pmhip_scene_filter_commit installs them once every view has been
// filtered against the *unfiltered* maps of its neighbours (EVT_ADJUSTDEPTHMAP is processed after all filter events, :2183-2210).
int pmhip_scene_filter(pmhip_engine* e, const int32_t* viewIds, int nViews, int bAdjust, uint32_t nMinViewsFilter,
		uint32_t nMinViewsFilterAdjust, float fDepthDiffThreshold, int sync) {
	if (!e || !viewIds || nViews <= 0 || e->nImages < 2) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t P0 = (size_t)e->w * e->h;
	if (!e->d_fdepth) {
		HIPCHK(e, hipMalloc(&e->d_fdepth, sizeof(float) * P0 * e->nImages));
		HIPCHK(e, hipMalloc(&e->d_fconf, sizeof(float) * P0 * e->nImages));
		HIPCHK(e, hipMalloc(&e->d_fvalid, e->nImages));
		HIPCHK(e, hipMemsetAsync(e->d_fvalid, 0, e->nImages, e->stream));
	}
	// every view is filtered at its own size (the reference sizes each depth map on its own image): the splat buffer holds the largest reference view of the call,
	// a view with its own size stages its result in its own buffers
	size_t Pref = 0, Pany = 0;
	for (int b = 0; b < nViews; ++b) {
		const int id = viewIds[b];
		if (id < 0 || id >= e->nImages || !e->views[id].set) { e->err = "view not set"; return PMHIP_E_ARG; }
		Pref = std::max(Pref, e->vpix(id)); Pany = std::max(Pany, e->vpix(id));
		SceneView& v = e->views[id];
		for (int k = 0; k < v.nNb; ++k) if (v.nb[k] >= 0 && v.nb[k] < e->nImages) Pany = std::max(Pany, e->vpix(v.nb[k]));
		if (v.sw && !v.oFDepth) { HIPCHK(e, hipMalloc(&v.oFDepth, sizeof(float) * e->vpix(id))); HIPCHK(e, hipMalloc(&v.oFConf, sizeof(float) * e->vpix(id))); }
	}
	const int CH = std::min(nViews, 4); // reference views per launch: bounds the splat buffer (8 x 8 B per pixel per view)
	if (e->splatCap < CH || e->splatPix < Pref) {
		HIPCHK(e, hipStreamSynchronize(e->stream));
		if (e->d_splat) hipFree(e->d_splat); if (e->d_ftasks) hipFree(e->d_ftasks); if (e->h_ftasks) hipHostFree(e->h_ftasks);
		e->d_splat = nullptr; e->d_ftasks = nullptr; e->h_ftasks = nullptr;
		const int cap = std::max(CH, e->splatCap); const size_t pix = std::max(Pref, e->splatPix);
		HIPCHK(e, hipMalloc(&e->d_splat, sizeof(unsigned long long) * pix * PMF_MAXN * cap));
		HIPCHK(e, hipMalloc(&e->d_ftasks, sizeof(PMFTask) * cap));
		HIPCHK(e, hipHostMalloc(&e->h_ftasks, sizeof(PMFTask) * cap));
		e->splatCap = cap; e->splatPix = pix;
	}
	const unsigned nCal = (unsigned)e->nImages;
	const unsigned nMinViews = std::min(nMinViewsFilter, nCal - 1), nMinViewsAdjust = std::min(nMinViewsFilterAdjust, nCal - 1);
	std::vector<unsigned char> hv(e->nImages, 2); // 2 = untouched
	for (int b0 = 0; b0 < nViews; b0 += CH) {
		const int nb = std::min(CH, nViews - b0);
		HIPCHK(e, hipStreamSynchronize(e->stream)); // staging reuse
		for (int b = 0; b < nb; ++b) {
			const int id = viewIds[b0 + b];
			const SceneView& v = e->views[id];
			PMFTask& t = e->h_ftasks[b];
			memset(&t, 0, sizeof(t));
			memcpy(t.ref.K, v.K, 72); memcpy(t.ref.R, v.R, 72); memcpy(t.ref.C, v.C, 24);
			t.refDepth = e->depthOf(id); t.refConf = e->confOf(id);
			t.N = 0;
			for (int k = 0; k < v.nNb && t.N < PMF_MAXN; ++k) {
				const int j = v.nb[k];
				// neighbours without a depth map are skipped before the eight slots are filled (SceneDensify.cpp:2150-2163: !depthData.IsValid())
				if (j < 0 || j >= e->nImages || !e->views[j].set || !e->views[j].hasMaps) continue;
				const SceneView& sv = e->views[j];
				memcpy(t.nb[t.N].K, sv.K, 72); memcpy(t.nb[t.N].R, sv.R, 72); memcpy(t.nb[t.N].C, sv.C, 24);
				t.nbDepth[t.N] = e->depthOf(j); t.nbConf[t.N] = e->confOf(j); t.nbw[t.N] = e->vw(j); t.nbh[t.N] = e->vh(j);
				++t.N;
			}
			t.splat = e->d_splat + e->splatPix * PMF_MAXN * b;
			t.outDepth = v.sw ? v.oFDepth : e->d_fdepth + P0 * id; t.outConf = v.sw ? v.oFConf : e->d_fconf + P0 * id;
			t.w = e->vw(id); t.h = e->vh(id); t.dMin = v.dMin; t.dMax = v.dMax;
			t.filterable = !((unsigned)t.N < nMinViews || (unsigned)t.N < nMinViewsAdjust); // :1060-1063
			hv[id] = t.filterable ? 1 : 0;
		}
		HIPCHK(e, hipMemcpyAsync(e->d_ftasks, e->h_ftasks, sizeof(PMFTask) * nb, hipMemcpyHostToDevice, e->stream));
		const size_t nS = e->splatPix * PMF_MAXN * nb;
		hipLaunchKernelGGL(pmf_clear_kernel, dim3((unsigned)std::min<size_t>((nS + 255) / 256, 65535)), dim3(256), 0, e->stream, e->d_splat, nS);
		const unsigned gx = (unsigned)std::min<size_t>((Pany + 255) / 256, 2048);
		hipLaunchKernelGGL(pmf_splat_kernel, dim3(gx, nb, PMF_MAXN), dim3(256), 0, e->stream, e->d_ftasks);
		hipLaunchKernelGGL(pmf_vote_kernel, dim3(gx, nb), dim3(256), 0, e->stream, e->d_ftasks, bAdjust, nMinViews, nMinViewsAdjust, fDepthDiffThreshold);
		HIPCHK(e, hipGetLastError());
	}
	HIPCHK(e, hipStreamSynchronize(e->stream));
	for (int i = 0; i < e->nImages; ++i) if (hv[i] != 2) HIPCHK(e, hipMemcpyAsync(e->d_fvalid + i, &hv[i], 1, hipMemcpyHostToDevice, e->stream));
	// hv lives on this stack frame: the small copies above must have left it whatever the caller asked for; `sync` only says whether the caller
	// wants the filter kernels themselves finished on return (they are, as a consequence) -- kept in the ABI for symmetry with pmhip_scene_estimate
	HIPCHK(e, hipStreamSynchronize(e->stream));
	(void)sync;
	return 0;
}

// DepthMapsData::GapInterpolation (SceneDensify.cpp:904-1045) on the maps of these views, in place (row pass, then column pass).
int pmhip_scene_gap_interpolation(pmhip_engine* e, const int32_t* viewIds, int nViews, uint32_t nIpolGapSize, float fDepthDiffThreshold) {
	if (!e || !viewIds || nViews <= 0) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	size_t Pmax = 0;
	for (int b = 0; b < nViews; ++b) { if (viewIds[b] < 0 || viewIds[b] >= e->nImages) return PMHIP_E_ARG; Pmax = std::max(Pmax, e->vpix(viewIds[b])); }
	float* tmp = nullptr; PMGTask* dt = nullptr;
	HIPCHK(e, hipMalloc(&tmp, sizeof(float) * Pmax * 5));
	HIPCHK(e, hipMalloc(&dt, sizeof(PMGTask) * 2));
	const float th = fDepthDiffThreshold * 2.5f;
	int rc = 0;
	for (int b = 0; b < nViews && rc == 0; ++b) {
		const int id = viewIds[b];
		const size_t P0 = e->vpix(id); const int vw = e->vw(id), vh = e->vh(id);
		const unsigned gx = (unsigned)std::min<size_t>((P0 + 255) / 256, 4096);
		float* D = e->depthOf(id); float* N = e->normalOf(id); float* Cf = e->confOf(id);
		PMGTask ht[2] = {{D, N, Cf, tmp, tmp + P0, tmp + P0 * 4, vw, vh}, {tmp, tmp + P0, tmp + P0 * 4, D, N, Cf, vw, vh}};
		if (hipMemcpyAsync(dt, ht, sizeof(ht), hipMemcpyHostToDevice, e->stream) != hipSuccess) { rc = PMHIP_E_HIP; break; }
		hipLaunchKernelGGL(pmf_gap_kernel, dim3(gx, 1), dim3(256), 0, e->stream, dt, 1, nIpolGapSize, th);       // 1. row-wise
		hipLaunchKernelGGL(pmf_gap_kernel, dim3(gx, 1), dim3(256), 0, e->stream, dt + 1, 0, nIpolGapSize, th);   // 2. column-wise
		if (hipStreamSynchronize(e->stream) != hipSuccess) { rc = PMHIP_E_HIP; break; }
	}
	hipFree(tmp); hipFree(dt);
	if (rc == PMHIP_E_HIP) e->err = "gap interpolation: HIP error";
	return rc;
}

// DepthMapsData::RemoveSmallSegments (SceneDensify.cpp:809-900) on the maps of these views, in place; see pm_filter.hip.
int pmhip_scene_remove_small_segments(pmhip_engine* e, const int32_t* viewIds, int nViews, uint32_t nSpeckleSize, float fDepthDiffThreshold) {
	if (!e || !viewIds || nViews <= 0) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	int nmax = 0;
	for (int b = 0; b < nViews; ++b) { if (viewIds[b] < 0 || viewIds[b] >= e->nImages) return PMHIP_E_ARG; nmax = std::max(nmax, (int)e->vpix(viewIds[b])); }
	const int cap = nmax; // asymmetric edges are rare; n pairs is far more than ever needed
	int *parent = nullptr, *size = nullptr, *edges = nullptr, *nEdges = nullptr, *ovr = nullptr;
	HIPCHK(e, hipMalloc(&parent, sizeof(int) * nmax)); HIPCHK(e, hipMalloc(&size, sizeof(int) * nmax));
	HIPCHK(e, hipMalloc(&edges, sizeof(int) * 2 * cap)); HIPCHK(e, hipMalloc(&nEdges, sizeof(int))); HIPCHK(e, hipMalloc(&ovr, sizeof(int) * 2 * cap));
	const float th = fDepthDiffThreshold * 0.7f;
	int rc = 0;
	std::vector<int> hedges, hsize;
	for (int b = 0; b < nViews && rc == 0; ++b) {
		const int id = viewIds[b];
		const int n = (int)e->vpix(id), vw = e->vw(id), vh = e->vh(id);
		const unsigned gx = (unsigned)std::min<size_t>(((size_t)n + 255) / 256, 4096);
		float* D = e->depthOf(id); float* N = e->normalOf(id); float* Cf = e->confOf(id);
		hipMemsetAsync(nEdges, 0, sizeof(int), e->stream);
		hipLaunchKernelGGL(pmf_cc_init_kernel, dim3(gx), dim3(256), 0, e->stream, parent, size, n);
		hipLaunchKernelGGL(pmf_cc_hook_kernel, dim3(gx), dim3(256), 0, e->stream, D, parent, vw, vh, th);
		hipLaunchKernelGGL(pmf_cc_flatten_kernel, dim3(gx), dim3(256), 0, e->stream, parent, size, n);
		hipLaunchKernelGGL(pmf_cc_asym_kernel, dim3(gx), dim3(256), 0, e->stream, D, parent, vw, vh, th, edges, nEdges, cap);
		int ne = 0;
		if (hipMemcpyAsync(&ne, nEdges, sizeof(int), hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) { rc = PMHIP_E_HIP; break; }
		if (ne > cap) { e->err = "remove_small_segments: asymmetric edge list overflow"; rc = PMHIP_E_HIP; break; }
		if (ne > 0) {
			// replay the reference's seed order on the quotient graph of components linked by one-directional edges
			hedges.resize(2 * (size_t)ne); hsize.resize(2 * (size_t)ne);
			hipLaunchKernelGGL(pmf_cc_gather_kernel, dim3((2 * ne + 255) / 256), dim3(256), 0, e->stream, size, edges, ovr, 2 * ne);   // (a 4K view: 2*ne ints instead of 33 MB)
			if (hipMemcpyAsync(hedges.data(), edges, sizeof(int) * 2 * ne, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
				hipMemcpyAsync(hsize.data(), ovr, sizeof(int) * 2 * ne, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) { rc = PMHIP_E_HIP; break; }
			std::map<int, int> csize;                   // component root -> size
			for (int k = 0; k < 2 * ne; ++k) csize[hedges[k]] = hsize[k];
			std::map<int, std::set<int>> adj;
			for (int k = 0; k < ne; ++k) { adj[hedges[2 * k]].insert(hedges[2 * k + 1]); adj[hedges[2 * k + 1]]; }
			std::set<int> done;
			std::vector<int> pairs;
			for (auto& kv : adj) {                      // std::map iterates roots in increasing (= seed) order
				const int r = kv.first;
				if (done.count(r)) continue;
				std::vector<int> seg{r}; done.insert(r);
				for (size_t q = 0; q < seg.size(); ++q) for (int nb : adj[seg[q]]) if (!done.count(nb)) { done.insert(nb); seg.push_back(nb); }
				long total = 0; for (int x : seg) total += csize[x];
				const int val = total < (long)nSpeckleSize ? 0 : (int)nSpeckleSize;   // forces remove / keep for every member
				for (int x : seg) { pairs.push_back(x); pairs.push_back(val); }
			}
			const int np = (int)(pairs.size() / 2);
			if (hipMemcpy(ovr, pairs.data(), sizeof(int) * pairs.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = PMHIP_E_HIP; break; }
			hipLaunchKernelGGL(pmf_cc_override_kernel, dim3((np + 255) / 256), dim3(256), 0, e->stream, size, ovr, np);
		}
		hipLaunchKernelGGL(pmf_cc_apply_kernel, dim3(gx), dim3(256), 0, e->stream, D, N, Cf, parent, size, vw, vh, (int)nSpeckleSize);
		if (hipStreamSynchronize(e->stream) != hipSuccess) { rc = PMHIP_E_HIP; break; }
	}
	hipFree(parent); hipFree(size); hipFree(edges); hipFree(nEdges); hipFree(ovr);
	if (rc == PMHIP_E_HIP && e->err.empty()) e->err = "remove_small_segments: HIP error";
	return rc;
}

// install the staged filtered depth / confidence maps of the views filtered since the last commit (normal maps are
// left untouched, exactly like the reference: LoadDepthMap + LoadConfidenceMap only, SceneDensify.cpp:2190-2192)
int pmhip_scene_filter_commit(pmhip_engine* e) {
	if (!e || !e->d_fdepth) return PMHIP_E_STATE;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t P0 = (size_t)e->w * e->h;
	std::vector<unsigned char> hv(e->nImages);
	HIPCHK(e, hipMemcpy(hv.data(), e->d_fvalid, e->nImages, hipMemcpyDeviceToHost));
	for (int i = 0; i < e->nImages; ++i) if (hv[i] == 1) {
		const SceneView& v = e->views[i];
		HIPCHK(e, hipMemcpyAsync(e->depthOf(i), v.sw ? v.oFDepth : e->d_fdepth + P0 * i, sizeof(float) * e->vpix(i), hipMemcpyDeviceToDevice, e->stream));
		HIPCHK(e, hipMemcpyAsync(e->confOf(i), v.sw ? v.oFConf : e->d_fconf + P0 * i, sizeof(float) * e->vpix(i), hipMemcpyDeviceToDevice, e->stream));
	}
	HIPCHK(e, hipMemsetAsync(e->d_fvalid, 0, e->nImages, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int pmhip_scene_copy(pmhip_engine* e, int what, int firstIdx, int count, void* devPtr, int toEngine) {
	if (!e || !devPtr || count <= 0 || firstIdx < 0 || firstIdx + count > e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	char* base = (char*)pmhip_scene_device_ptr(e, what, firstIdx);
	if (!base) return PMHIP_E_ARG;
	bool sized = false;
	for (int i = firstIdx; i < firstIdx + count; ++i) sized = sized || e->views[i].sw;
	if (sized && count > 1) { e->err = "pmhip_scene_copy: a range that contains a view with its own size must be copied view by view"; return PMHIP_E_SIZE; }
	const size_t bytes = sizeof(float) * (sized ? e->vpix(firstIdx) : (size_t)e->w * e->h * count) * (what == 2 ? 3 : 1);
	if (sized && toEngine && what == 0) e->views[firstIdx].sideDirty = true;
	HIPCHK(e, hipMemcpyAsync(toEngine ? (void*)base : devPtr, toEngine ? devPtr : (void*)base, bytes, hipMemcpyDeviceToDevice, e->stream));
	if (toEngine && what == 0) e->pyramidDirty = true;
	if (toEngine && what == 1) for (int i = firstIdx; i < firstIdx + count; ++i) e->views[i].hasMaps = true;   // depth maps gathered from other ranks
	return 0;
}

// HBM the scene holds right now: image pyramids with their quad layouts, the four map arrays, masks, the filter's staging, the batch scratch of the largest batch estimated
// so far, and the own storage of views that carry their own size.  Not the working buffers of pmhip_scene_fuse (they live only during fusion and its download).
uint64_t pmhip_scene_bytes(pmhip_engine* e) {
	if (!e || e->nImages <= 0) return 0;
	const size_t N = (size_t)e->nImages, P0 = (size_t)e->w * e->h;
	size_t b = 0;
	for (int l = 0; l <= e->nLevels; ++l) {
		if (e->d_img[l]) b += sizeof(float) * (size_t)e->lw(l) * e->lh(l) * N * 2;   // row-major + folded anti-diagonal-major
		if (e->d_imgQ[l]) b += sizeof(float4) * e->skewPitch(l) * N;
		if (e->d_mask[l]) b += (size_t)e->lw(l) * e->lh(l) * N;
	}
	if (e->d_depth) b += sizeof(float) * P0 * N * 6;                       // depth, normal (3), confidence, previous round's depth
	if (e->d_fdepth) b += sizeof(float) * P0 * N * 2 + P0 * N;             // FilterDepthMap staging
	if (e->d_splat) b += sizeof(unsigned long long) * e->splatPix * PMF_MAXN * (size_t)e->splatCap;
	if (e->batchCap) {
		b += sizeof(float) * (size_t)e->batchCap * e->batchW * e->batchH;
		for (int l = 1; l <= e->nLevels; ++l) b += sizeof(float) * (size_t)e->batchCap * 6 * lvlSize(e->batchW, l) * lvlSize(e->batchH, l);
		b += (sizeof(PMTask) + sizeof(PMUpTask)) * 4 * (size_t)e->batchCap;
	}
	for (const SceneView& v : e->views) {
		if (v.sDepth) b += sizeof(float) * (size_t)v.dw * v.dh;
		if (!v.sw) continue;
		const size_t P = (size_t)v.sw * v.sh;
		b += sizeof(float) * P * 6 + (v.oFDepth ? sizeof(float) * P * 2 : 0) + (v.oBgr ? 3 * P : 0);
		for (int l = 0; l <= e->nLevels; ++l) {
			const size_t lw = (size_t)lvlSize(v.sw, l), lh = (size_t)lvlSize(v.sh, l);
			if (v.sImg[l]) b += sizeof(float) * lw * lh * 2;
			if (v.sImgQ[l]) b += sizeof(float4) * (lw + lh - 1) * lh;
			if (v.oMask[l]) b += lw * lh;
		}
	}
	return (uint64_t)b;
}

int pmhip_sync(pmhip_engine* e) { if (!e) return PMHIP_E_ARG; HIPCHK(e, hipSetDevice(e->device)); HIPCHK(e, hipStreamSynchronize(e->stream)); return 0; }
void* pmhip_stream(pmhip_engine* e) { return e ? (void*)e->stream : nullptr; }

int pmhip_stats_reset(pmhip_engine* e, int enableEvents) {
	if (!e) return PMHIP_E_ARG;
	hipSetDevice(e->device);
	collectStats(e);
	memset(&e->stats, 0, sizeof(e->stats));
	e->statsOn = enableEvents != 0;
	return 0;
}
int pmhip_stats_get(pmhip_engine* e, PMHipKernelStats* out) {
	if (!e || !out) return PMHIP_E_ARG;
	hipSetDevice(e->device);
	int rc = collectStats(e); if (rc) return rc;
	*out = e->stats;
	return 0;
}

// PatchMatchCUDA::EstimateDepthMap(DepthData&), libs/MVS/PatchMatchCUDA.cpp:174-416 -- host buffers in, host buffers out.
int pmhip_estimate_depth_map(pmhip_engine* e, PMHipDepthData* dd, const PMHipParams* p, int nGeometricIter) {
	return pmhip_estimate_depth_map_masked(e, dd, nullptr, 0, p, nGeometricIter);
}
// ... with DepthData::mask (libs/MVS/DepthMap.h:211; filled by DepthEstimator::ImportIgnoreMask, applied in SceneDensify.cpp:656-683)
int pmhip_estimate_depth_map_masked(pmhip_engine* e, PMHipDepthData* dd, const unsigned char* mask, int maskOption, const PMHipParams* p, int nGeometricIter) {
	if (!e || !dd || !p || !dd->views || dd->nViews < 2 || dd->nViews > 1 + PM_MAX_SRC || !dd->depthMap || !dd->normalMap || !dd->confMap) return PMHIP_E_ARG;
	if (!e->inited) { e->err = "pmhip_init not called"; return PMHIP_E_STATE; }
	const int w = dd->views[0].w, h = dd->views[0].h, n = dd->nViews;
	for (int i = 0; i < n; ++i) {
		if (!dd->views[i].image || dd->views[i].w < 3 || dd->views[i].h < 3) return PMHIP_E_ARG;
		if (nGeometricIter >= 0 && i > 0 && !dd->views[i].depth) { e->err = "geometric round needs views[i].depth"; return PMHIP_E_ARG; }
	}
	const int S = (int)p->nSubResolutionLevels;
	if (S > 3) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	// keep device buffers between calls; re-allocate only when geometry changes (reference: PatchMatchCUDA.cpp:264-322)
	if (e->nImages != n || e->w != w || e->h != h || e->nLevels < S) {
		int rc = pmhip_scene_create(e, n, w, h, std::max(S, e->nImages == n && e->w == w && e->h == h ? e->nLevels : 0));
		if (rc) return rc;
	}
	const size_t P0 = (size_t)w * h;
	int32_t nb[PM_MAX_SRC];
	for (int i = 1; i < n; ++i) nb[i - 1] = i;
	for (int i = 0; i < n; ++i) {
		const PMHipView& v = dd->views[i];
		// source views may come in any size (neighbours rescaled by ViewData::ScaleImage, DepthMap.h:194-204): they keep their own pyramid
		int rc = pmhip_scene_set_view_sized(e, i, v.image, v.w, v.h, 0, v.K, v.R, v.C, dd->dMin, dd->dMax, nb, i == 0 ? n - 1 : 0);
		if (rc) return rc;
		e->views[i].id = v.id;
		if (i == 0) continue;
		if (nGeometricIter < 0) { if (e->views[i].sDepth) { rc = pmhip_scene_set_source_depth(e, i, nullptr, 0, 0, nullptr, nullptr, nullptr); if (rc) return rc; } continue; }
		// the known depth-map is read with the camera stored next to it (cameraDepthMap) and has its own size; an all-zero Kd / Rd means
		// "not filled in" and stands for the view's own camera
		bool zero = true; for (int k = 0; k < 9; ++k) zero = zero && v.Kd[k] == 0.0 && v.Rd[k] == 0.0;
		const int dw = v.dw > 0 ? v.dw : v.w, dh = v.dh > 0 ? v.dh : v.h;
		const bool own = zero || (!memcmp(v.Kd, v.K, 72) && !memcmp(v.Rd, v.R, 72) && !memcmp(v.Cd, v.C, 24));
		if (own && dw == w && dh == h && v.w == w && v.h == h) {
			if (e->views[i].sDepth) { rc = pmhip_scene_set_source_depth(e, i, nullptr, 0, 0, nullptr, nullptr, nullptr); if (rc) return rc; }
			HIPCHK(e, hipMemcpyAsync(e->d_snap + P0 * i, v.depth, sizeof(float) * P0, hipMemcpyHostToDevice, e->stream));
		} else {
			rc = pmhip_scene_set_source_depth(e, i, v.depth, dw, dh, zero ? v.K : v.Kd, zero ? v.R : v.Rd, zero ? v.C : v.Cd);
			if (rc) return rc;
		}
	}
	int rc = pmhip_scene_set_maps(e, 0, dd->depthMap, dd->normalMap);
	if (rc) return rc;
	// the engine is reused between calls: install this call's mask state and leave none behind
	rc = pmhip_scene_set_mask(e, 0, mask);
	if (rc) return rc;
	const int savedMode = e->maskMode;
	e->maskMode = mask ? -1 : (maskOption ? 1 : 0);
	const int32_t id0 = 0;
	rc = estimateBatch(e, &id0, 1, *p, nGeometricIter);
	e->maskMode = savedMode;
	if (mask) pmhip_scene_set_mask(e, 0, nullptr);
	if (rc) return rc;
	return pmhip_scene_get_maps(e, 0, dd->depthMap, dd->normalMap, dd->confMap);
}

// in-kernel phase counters (only populated by -DPM_PROFILE builds); reset != 0 clears them after reading
int pmhip_prof_get(pmhip_engine* e, unsigned long long out16[16], int reset) {
	if (!e || !out16) return PMHIP_E_ARG;
#ifdef PM_PROFILE
	HIPCHK(e, hipSetDevice(e->device));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	HIPCHK(e, hipMemcpyFromSymbol(out16, HIP_SYMBOL(pm_prof), sizeof(unsigned long long) * 16));
	if (reset) { unsigned long long z[16] = {0}; HIPCHK(e, hipMemcpyToSymbol(HIP_SYMBOL(pm_prof), z, sizeof(z))); }
	return 0;
#else
	for (int i = 0; i < 16; ++i) out16[i] = 0;
	return 0;
#endif
}

#ifdef PM_PROFILE
// (profile builds only, tools/phase_prof.py) histogram of pm_visit's trips by the number of pixels of the wave that take part
int pmhip_prof_hist(pmhip_engine* e, unsigned long long out17[17], int reset) {
	if (!e || !out17) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	HIPCHK(e, hipMemcpyFromSymbol(out17, HIP_SYMBOL(pm_hist), sizeof(unsigned long long) * 17));
	if (reset) { unsigned long long z[17] = {0}; HIPCHK(e, hipMemcpyToSymbol(HIP_SYMBOL(pm_hist), z, sizeof(z))); }
	return 0;
}
#endif

int pmhip_math_eval(pmhip_engine* e, int kind, const float* a, const float* b, float* out, size_t n) {
	if (!e || !a || !out || n == 0) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	float *da = nullptr, *db = nullptr, *dout = nullptr;
	HIPCHK(e, hipMalloc(&da, n * 4)); HIPCHK(e, hipMalloc(&db, n * 4)); HIPCHK(e, hipMalloc(&dout, n * 4));
	HIPCHK(e, hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
	HIPCHK(e, hipMemcpy(db, b ? b : a, n * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(pm_math_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, e->stream, kind, da, db, dout, n);
	HIPCHK(e, hipStreamSynchronize(e->stream));
	HIPCHK(e, hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
	hipFree(da); hipFree(db); hipFree(dout);
	return 0;
}

int pmhip_resize(pmhip_engine* e, int kind, const float* src, int w, int h, int arg, float* dst) {
	if (!e || !src || !dst || w <= 0 || h <= 0) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	const size_t ns = (size_t)w * h;
	int dw, dh;
	if (kind == 0) { if (arg < 1) return PMHIP_E_SIZE; dw = (int)nearbyint((double)w / arg); dh = (int)nearbyint((double)h / arg); } else { dw = w * 2; dh = h * 2; }
	const size_t nd = (size_t)dw * dh;
	float *ds = nullptr, *dd = nullptr, *dn = nullptr, *dn2 = nullptr, *dp = nullptr; PMUpTask* du = nullptr;
	HIPCHK(e, hipMalloc(&ds, ns * 4)); HIPCHK(e, hipMalloc(&dd, nd * 4));
	HIPCHK(e, hipMemcpy(ds, src, ns * 4, hipMemcpyHostToDevice));
	const int blocks = (int)std::min<size_t>((nd + 255) / 256, 4096);
	if (kind == 0) {
		hipLaunchKernelGGL(pm_area_kernel, dim3(blocks), dim3(256), 0, e->stream, ds, dd, w, h, dw, dh, arg, 1);
	} else if (kind == 1) {
		HIPCHK(e, hipMalloc(&dn, ns * 12)); HIPCHK(e, hipMalloc(&dn2, nd * 12)); HIPCHK(e, hipMalloc(&dp, nd * 4)); HIPCHK(e, hipMalloc(&du, sizeof(PMUpTask)));
		HIPCHK(e, hipMemset(dn, 0, ns * 12));
		PMUpTask u{ds, dn, dd, dn2, dp};
		HIPCHK(e, hipMemcpy(du, &u, sizeof(u), hipMemcpyHostToDevice));
		hipLaunchKernelGGL(pm_upsample_kernel, dim3(blocks, 1), dim3(256), 0, e->stream, du, w, h, dw, dh, 0);
	} else {
		hipLaunchKernelGGL(pm_nearest_up_f_kernel, dim3(blocks), dim3(256), 0, e->stream, ds, dd, w, h, dw, dh);
	}
	HIPCHK(e, hipStreamSynchronize(e->stream));
	HIPCHK(e, hipMemcpy(dst, dd, nd * 4, hipMemcpyDeviceToHost));
	hipFree(ds); hipFree(dd); if (dn) hipFree(dn); if (dn2) hipFree(dn2); if (dp) hipFree(dp); if (du) hipFree(du);
	return 0;
}

// ---- FuseDepthMaps (libs/MVS/SceneDensify.cpp:1372-1650) on the resident scene -------------------------------------------------
static int ensureFuse(pmhip_engine* e) {
	auto& f = e->fu;
	size_t P = (size_t)e->w * e->h; const size_t N = (size_t)e->nImages;
	for (int i = 0; i < e->nImages; ++i) P = std::max(P, e->vpix(i));           // a slab holds the largest image
	if (f.depth && f.slab >= P) return 0;
	if (f.depth) {                                                              // a view grew: start over (the colour images and the output stay)
		HIPCHK(e, hipStreamSynchronize(e->stream));
		void* ptrs[] = {f.depth, f.claimed, f.resv, f.cams, f.recN, f.recColor, f.recX, f.recWeight, f.recNormal, f.recView, f.recProj, f.pend[0], f.pend[1], f.counters, f.nDepthsDev,
		                f.tileSums, f.tileOff, f.normalS, f.confS, f.bgrS, f.dims};
		for (void* q : ptrs) if (q) hipFree(q);
		if (f.pin) hipHostFree(f.pin);
		f.depth = nullptr; f.claimed = f.resv = nullptr; f.cams = nullptr; f.recN = f.recColor = nullptr; f.recX = f.recWeight = f.recNormal = nullptr; f.recView = f.recProj = nullptr;
		f.pend[0] = f.pend[1] = nullptr; f.counters = nullptr; f.nDepthsDev = nullptr; f.tileSums = f.tileOff = nullptr; f.normalS = f.confS = nullptr; f.bgrS = nullptr; f.dims = nullptr; f.pin = nullptr;
	}
	HIPCHK(e, hipMalloc(&f.depth, sizeof(float) * P * N));
	HIPCHK(e, hipMalloc(&f.claimed, sizeof(uint32_t) * P * N));
	HIPCHK(e, hipMalloc(&f.resv, sizeof(uint32_t) * P * N));
	HIPCHK(e, hipMalloc(&f.cams, sizeof(PMFuseCam) * N));
	HIPCHK(e, hipMalloc(&f.dims, sizeof(int) * 2 * N));
	HIPCHK(e, hipMalloc(&f.recN, P)); HIPCHK(e, hipMalloc(&f.recColor, 3 * P));
	HIPCHK(e, hipMalloc(&f.recX, sizeof(float) * 3 * P)); HIPCHK(e, hipMalloc(&f.recNormal, sizeof(float) * 3 * P));
	HIPCHK(e, hipMalloc(&f.recWeight, sizeof(float) * PMFU_MAXV * P));
	HIPCHK(e, hipMalloc(&f.recView, sizeof(uint32_t) * PMFU_MAXV * P)); HIPCHK(e, hipMalloc(&f.recProj, sizeof(uint32_t) * PMFU_MAXV * P));
	HIPCHK(e, hipMalloc(&f.pend[0], sizeof(uint32_t) * P)); HIPCHK(e, hipMalloc(&f.pend[1], sizeof(uint32_t) * P));
	HIPCHK(e, hipMalloc(&f.counters, sizeof(uint32_t) * 8)); HIPCHK(e, hipMalloc(&f.nDepthsDev, sizeof(unsigned long long) * 2));
	const size_t nTiles = (P + PMFU_TILE - 1) / PMFU_TILE;
	HIPCHK(e, hipMalloc(&f.tileSums, sizeof(uint2) * nTiles)); HIPCHK(e, hipMalloc(&f.tileOff, sizeof(uint2) * nTiles));
	HIPCHK(e, hipHostMalloc(&f.pin, sizeof(uint32_t) * 16));
	f.slab = P;
	return 0;
}

int pmhip_scene_set_color(pmhip_engine* e, int idx, const unsigned char* bgr) {
	if (!e || !bgr || idx < 0 || idx >= e->nImages) return PMHIP_E_ARG;
	HIPCHK(e, hipSetDevice(e->device));
	auto& f = e->fu;
	const size_t P = (size_t)e->w * e->h;
	if (f.hasBgr.empty()) f.hasBgr.assign(e->nImages, 0);
	SceneView& v = e->views[idx];
	if (v.sw) {                                                                  // a view with its own size keeps its colour image itself
		if (!v.oBgr) HIPCHK(e, hipMalloc(&v.oBgr, 3 * e->vpix(idx)));
		HIPCHK(e, hipMemcpyAsync(v.oBgr, bgr, 3 * e->vpix(idx), hipMemcpyHostToDevice, e->stream));
		HIPCHK(e, hipStreamSynchronize(e->stream));
		f.hasBgr[idx] = 1;
		return 0;
	}
	if (!f.bgr) HIPCHK(e, hipMalloc(&f.bgr, 3 * P * e->nImages));
	HIPCHK(e, hipMemcpyAsync(f.bgr + 3 * P * idx, bgr, 3 * P, hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	f.hasBgr[idx] = 1;
	return 0;
}

int pmhip_scene_fuse(pmhip_engine* e, const int32_t* order, int nOrder, const PMHipFuseParams* prm, uint64_t* nPoints, uint64_t* nViews, uint64_t* nDepths) {
	if (!e || !order || nOrder <= 0 || !prm || e->nImages < 1) return PMHIP_E_ARG;
	bool mixed = false;
	for (int i = 0; i < e->nImages; ++i) { mixed = mixed || e->views[i].sw; if (e->vw(i) > 65535 || e->vh(i) > 65535) { e->err = "fusion stores projections as 16-bit pixel coordinates"; return PMHIP_E_SIZE; } }
	HIPCHK(e, hipSetDevice(e->device));
	int rc = ensureFuse(e); if (rc) return rc;
	auto& f = e->fu;
	const size_t P = f.slab, N = (size_t)e->nImages, P0 = (size_t)e->w * e->h;
	bool wantColor = prm->bEstimateColor != 0;
	if (wantColor) {
		// colours are read of the fused views and of their neighbours only (SceneDensify.cpp:1455-1560): a source-only slot -- a resampled copy of a neighbour that the estimation
		// read (ViewData::ScaleImage) -- has no depth map, is nobody's neighbour here and needs no colour
		std::vector<unsigned char> need((size_t)e->nImages, 0);
		for (int k = 0; k < nOrder; ++k) {
			const int v = order[k];
			if (v < 0 || v >= e->nImages) continue;
			need[(size_t)v] = 1;
			for (int j = 0; j < e->views[v].nNb; ++j) { const int b = e->views[v].nb[j]; if (b >= 0 && b < e->nImages) need[(size_t)b] = 1; }
		}
		for (int i = 0; i < e->nImages; ++i) if (need[(size_t)i] && e->views[i].set && (f.hasBgr.empty() || !f.hasBgr[i] || !(e->views[i].sw ? (const void*)e->views[i].oBgr : (const void*)f.bgr))) {
			e->err = "bEstimateColor needs pmhip_scene_set_color for every fused view and its neighbours"; return PMHIP_E_STATE; }
	}
	const bool wantNormal = prm->bEstimateNormal != 0;
	// cameras (P composed like Camera::ComposeP)
	std::vector<PMFuseCam> hc(N);
	for (size_t i = 0; i < N; ++i) { memset(&hc[i], 0, sizeof(PMFuseCam)); if (!e->views[i].set) continue; memcpy(hc[i].K, e->views[i].K, 72); memcpy(hc[i].R, e->views[i].R, 72); memcpy(hc[i].C, e->views[i].C, 24); pmfu_composeP(hc[i]); }
	HIPCHK(e, hipMemcpyAsync(f.cams, hc.data(), sizeof(PMFuseCam) * N, hipMemcpyHostToDevice, e->stream));
	// working copies of the depth maps, one slab per image; with views of different sizes also the read-only inputs are gathered into slabs (each image with its own row pitch)
	const bool slabs = mixed || P != P0;
	if (!slabs) HIPCHK(e, hipMemcpyAsync(f.depth, e->d_depth, sizeof(float) * P * N, hipMemcpyDeviceToDevice, e->stream));
	else {
		if (!f.normalS) { HIPCHK(e, hipMalloc(&f.normalS, sizeof(float) * 3 * P * N)); HIPCHK(e, hipMalloc(&f.confS, sizeof(float) * P * N)); }
		if (wantColor && !f.bgrS) HIPCHK(e, hipMalloc(&f.bgrS, 3 * P * N));
		HIPCHK(e, hipMemsetAsync(f.depth, 0, sizeof(float) * P * N, e->stream));
		std::vector<int> hw(N), hh(N);
		for (size_t i = 0; i < N; ++i) {
			const size_t Pi = e->vpix((int)i);
			hw[i] = e->vw((int)i); hh[i] = e->vh((int)i);
			HIPCHK(e, hipMemcpyAsync(f.depth + P * i, e->depthOf((int)i), sizeof(float) * Pi, hipMemcpyDeviceToDevice, e->stream));
			HIPCHK(e, hipMemcpyAsync(f.normalS + 3 * P * i, e->normalOf((int)i), sizeof(float) * 3 * Pi, hipMemcpyDeviceToDevice, e->stream));
			HIPCHK(e, hipMemcpyAsync(f.confS + P * i, e->confOf((int)i), sizeof(float) * Pi, hipMemcpyDeviceToDevice, e->stream));
			if (wantColor && e->views[i].set && f.hasBgr[i]) HIPCHK(e, hipMemcpyAsync(f.bgrS + 3 * P * i, e->views[i].sw ? e->views[i].oBgr : f.bgr + 3 * P0 * i, 3 * Pi, hipMemcpyDeviceToDevice, e->stream));
		}
		HIPCHK(e, hipMemcpyAsync(f.dims, hw.data(), sizeof(int) * N, hipMemcpyHostToDevice, e->stream));          // iw = dims, ih = dims + N
		HIPCHK(e, hipMemcpyAsync(f.dims + N, hh.data(), sizeof(int) * N, hipMemcpyHostToDevice, e->stream));
		HIPCHK(e, hipStreamSynchronize(e->stream));                                 // hw / hh live on this frame
	}
	hipLaunchKernelGGL(pmfu_fill_u32, dim3(2048), dim3(256), 0, e->stream, f.claimed, P * N, PMFU_NO_ID);
	hipLaunchKernelGGL(pmfu_fill_u32, dim3(2048), dim3(256), 0, e->stream, f.resv, P * N, PMFU_FREE);
	HIPCHK(e, hipMemsetAsync(f.counters, 0, sizeof(uint32_t) * 8, e->stream));
	HIPCHK(e, hipMemsetAsync(f.nDepthsDev, 0, sizeof(unsigned long long) * 2, e->stream));
	// output capacity: a point needs a seed and every pixel joins at most one point, so both are bounded by the valid depths
	hipLaunchKernelGGL(pmfu_count_valid, dim3(2048), dim3(256), 0, e->stream, f.depth, P * N, f.nDepthsDev + 1);
	unsigned long long nValid = 0;
	HIPCHK(e, hipMemcpyAsync(&nValid, f.nDepthsDev + 1, sizeof(nValid), hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	if (nValid >= 0xFFFFFFFFull) { e->err = "more than 2^32 depths"; return PMHIP_E_SIZE; }
	if (f.cap < (size_t)nValid + 1 || (wantColor && !f.out.colors) || (wantNormal && !f.out.normals)) {
		freeFuseOut(e);
		const size_t cap = (size_t)nValid + 1;
		HIPCHK(e, hipMalloc(&f.out.points, sizeof(float) * 3 * cap)); HIPCHK(e, hipMalloc(&f.out.viewStart, sizeof(uint32_t) * (cap + 1)));
		HIPCHK(e, hipMalloc(&f.out.views, sizeof(uint32_t) * cap)); HIPCHK(e, hipMalloc(&f.out.weights, sizeof(float) * cap));
		HIPCHK(e, hipMalloc(&f.out.projs, sizeof(uint16_t) * 2 * cap));
		if (wantColor) HIPCHK(e, hipMalloc(&f.out.colors, 3 * cap));
		if (wantNormal) HIPCHK(e, hipMalloc(&f.out.normals, sizeof(float) * 3 * cap));
		f.cap = cap;
	}
	PMFuseOut out = f.out;
	if (!wantColor) out.colors = nullptr;
	if (!wantNormal) out.normals = nullptr;
	const unsigned nMin = std::min<unsigned>(prm->nMinViewsFuse, (unsigned)e->nImages);
	const float normalError = cosf(prm->fNormalDiffThreshold * (3.14159265358979323846f / 180.f));   // COS(FD2R(x)), SceneDensify.cpp:1455
	f.rounds = 0;
	for (int o = 0; o < nOrder; ++o) {
		const int A = order[o];
		if (A < 0 || A >= e->nImages || !e->views[A].set) { e->err = "fuse: view not set"; return PMHIP_E_ARG; }
		const size_t PA = e->vpix(A);                                               // image A's own pixels: seeds, records, compaction
		const unsigned nTiles = (unsigned)((PA + PMFU_TILE - 1) / PMFU_TILE);
		PMFuseCtx c; memset(&c, 0, sizeof(c));
		c.w = e->w; c.h = e->h; c.nImages = e->nImages; c.A = A;
		c.slab = P; if (slabs) { c.iw = f.dims; c.ih = f.dims + N; }
		for (int k = 0; k < e->views[A].nNb && c.nNb < PMFU_MAXNB; ++k) { const int b = e->views[A].nb[k]; if (b >= 0 && b < e->nImages && b != A && e->views[b].set) c.nb[c.nNb++] = b; }
		c.depth = f.depth; c.normal = slabs ? f.normalS : e->d_normal; c.conf = slabs ? f.confS : e->d_conf; c.bgr = slabs ? (wantColor ? f.bgrS : nullptr) : f.bgr; c.claimed = f.claimed; c.resv = f.resv; c.cams = f.cams;
		c.nMinViewsFuse = nMin; c.fDepthDiffThreshold = prm->fDepthDiffThreshold; c.normalError = normalError;
		c.bEstimateColor = wantColor ? 1 : 0; c.bEstimateNormal = wantNormal ? 1 : 0;
		c.recN = f.recN; c.recX = f.recX; c.recView = f.recView; c.recWeight = f.recWeight; c.recProj = f.recProj; c.recColor = f.recColor; c.recNormal = f.recNormal;
		if (prm->nMinViewsFuse < 2) {   // MergeDepthMaps (Scene::DenseReconstruction, SceneDensify.cpp:1695-1698)
			hipLaunchKernelGGL(pmfu_merge_kernel, dim3((unsigned)std::min<size_t>((PA + 255) / 256, 4096)), dim3(256), 0, e->stream, c, f.nDepthsDev);
			hipLaunchKernelGGL(pmfu_tile_sums, dim3(nTiles), dim3(PMFU_TB), 0, e->stream, f.recN, (uint32_t)PA, f.tileSums);
			hipLaunchKernelGGL(pmfu_scan_tiles, dim3(1), dim3(1024), 0, e->stream, f.tileSums, nTiles, f.counters + 2, f.tileOff);
			hipLaunchKernelGGL(pmfu_scatter_kernel, dim3(nTiles), dim3(PMFU_TB), 0, e->stream, c, f.tileOff, out);
			HIPCHK(e, hipGetLastError());
			continue;
		}
		HIPCHK(e, hipMemsetAsync(f.counters, 0, sizeof(uint32_t) * 2, e->stream));
		hipLaunchKernelGGL(pmfu_seed_kernel, dim3((unsigned)std::min<size_t>((PA + 255) / 256, 4096)), dim3(256), 0, e->stream, c, f.pend[0], f.counters, f.nDepthsDev);
		HIPCHK(e, hipMemcpyAsync(f.pin, f.counters, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
		HIPCHK(e, hipStreamSynchronize(e->stream));
		uint32_t n = f.pin[0]; int cur = 0;
		while (n) {
			++f.rounds;
			HIPCHK(e, hipMemsetAsync(f.counters + 1, 0, sizeof(uint32_t), e->stream));
			const unsigned gb = (n + 255) / 256;
			hipLaunchKernelGGL(pmfu_reserve_kernel, dim3(gb), dim3(256), 0, e->stream, c, f.pend[cur], n);
			hipLaunchKernelGGL(pmfu_commit_kernel, dim3(gb), dim3(256), 0, e->stream, c, f.pend[cur], n, f.pend[cur ^ 1], f.counters + 1);
			HIPCHK(e, hipMemcpyAsync(f.pin, f.counters + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
			HIPCHK(e, hipStreamSynchronize(e->stream));
			if (f.pin[0] >= n) { e->err = "fuse: no progress in a reservation round"; return PMHIP_E_STATE; }
			n = f.pin[0]; cur ^= 1;
		}
		hipLaunchKernelGGL(pmfu_tile_sums, dim3(nTiles), dim3(PMFU_TB), 0, e->stream, f.recN, (uint32_t)PA, f.tileSums);
		hipLaunchKernelGGL(pmfu_scan_tiles, dim3(1), dim3(1024), 0, e->stream, f.tileSums, nTiles, f.counters + 2, f.tileOff);
		hipLaunchKernelGGL(pmfu_scatter_kernel, dim3(nTiles), dim3(PMFU_TB), 0, e->stream, c, f.tileOff, out);
		HIPCHK(e, hipGetLastError());
	}
	unsigned long long nd = 0;
	HIPCHK(e, hipMemcpyAsync(f.pin, f.counters + 2, sizeof(uint32_t) * 2, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipMemcpyAsync(&nd, f.nDepthsDev, sizeof(nd), hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	f.nPoints = f.pin[0]; f.nViews = f.pin[1]; f.nDepths = nd; f.haveColor = wantColor; f.haveNormal = wantNormal;
	const uint32_t last = (uint32_t)f.nViews;
	HIPCHK(e, hipMemcpyAsync(f.out.viewStart + f.nPoints, &last, sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	if (nPoints) *nPoints = f.nPoints; if (nViews) *nViews = f.nViews; if (nDepths) *nDepths = f.nDepths;
	return 0;
}

int pmhip_scene_fuse_get(pmhip_engine* e, float* points, uint32_t* viewStart, uint32_t* views, float* weights, uint16_t* projs, unsigned char* colors, float* normals) {
	if (!e || !e->fu.out.points) return PMHIP_E_STATE;
	HIPCHK(e, hipSetDevice(e->device));
	auto& f = e->fu;
	if ((colors && !f.haveColor) || (normals && !f.haveNormal)) { e->err = "fuse_get: colours / normals were not estimated"; return PMHIP_E_STATE; }
	const size_t n = (size_t)f.nPoints, v = (size_t)f.nViews;
	if (points && n) HIPCHK(e, hipMemcpyAsync(points, f.out.points, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, e->stream));
	if (viewStart) HIPCHK(e, hipMemcpyAsync(viewStart, f.out.viewStart, sizeof(uint32_t) * (n + 1), hipMemcpyDeviceToHost, e->stream));
	if (views && v) HIPCHK(e, hipMemcpyAsync(views, f.out.views, sizeof(uint32_t) * v, hipMemcpyDeviceToHost, e->stream));
	if (weights && v) HIPCHK(e, hipMemcpyAsync(weights, f.out.weights, sizeof(float) * v, hipMemcpyDeviceToHost, e->stream));
	if (projs && v) HIPCHK(e, hipMemcpyAsync(projs, f.out.projs, sizeof(uint16_t) * 2 * v, hipMemcpyDeviceToHost, e->stream));
	if (colors && n) HIPCHK(e, hipMemcpyAsync(colors, f.out.colors, 3 * n, hipMemcpyDeviceToHost, e->stream));
	if (normals && n) HIPCHK(e, hipMemcpyAsync(normals, f.out.normals, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

uint64_t pmhip_scene_fuse_rounds(pmhip_engine* e) { return e ? e->fu.rounds : 0; }

} // extern "C"
