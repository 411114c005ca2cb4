/* mvsfront.h -- the host-side steps in front of the depth-map estimator, as a dependency-free C library (libmvsfront.so):
 *   - the MVSI scene archive reader (MVS::Interface, libs/MVS/Interface.h:215-275,360-760 in /root/reference);
 *   - per-image pixel cameras (Platform::GetCamera, libs/MVS/Platform.cpp:44-54; Camera::GetK / ScaleK, libs/MVS/Camera.h:146-200);
 *   - neighbour-view selection (Scene::SelectNeighborViews, libs/MVS/Scene.cpp:801-934; FilterNeighborViews, :953-968; the score cut of
 *     DepthMapsData::InitViews, libs/MVS/SceneDensify.cpp:333-340);
 *   - the depth range and sparse seed maps of InitViews (SceneDensify.cpp:418-460; TriangulatePoints2DepthMap with bInitSparse,
 *     libs/MVS/DepthMap.cpp:1117-1157, with an own Delaunay triangulation instead of CGAL's).
 * A C++ host that does not link OpenMVS gets from `scene.mvs` to the arguments of pmhip_scene_set_view / pmhip_scene_set_maps with these
 * calls; openmvs_amd/mvsi.py + views.py are the same logic in numpy and the two are tested against each other (tests/test_mvsfront.py).
 * All functions return 0 on success, < 0 on error (-1 argument, -2 file / format, -3 not enough data for this image).
 */
#ifndef MVSFRONT_H_
#define MVSFRONT_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvsf_scene mvsf_scene;

/* ViewScore, libs/MVS/Image.h (ID, points, scale, angle, area, score) */
typedef struct MVSFViewScore { uint32_t ID, points; float scale, angle, area, score; } MVSFViewScore;

/* the OPTDENSE values these steps read (defaults of libs/MVS/DepthMap.cpp:69-90 via mvsf_default_options) */
typedef struct MVSFOptions {
	uint32_t nMinViews, nMaxViews, nMinViewsTrustPoint, nNumViews, nPointInsideROI;
	float fViewMinScore, fViewMinScoreRatio, fMinArea, fMinAngle, fOptimAngle, fMaxAngle;   /* angles in degrees */
} MVSFOptions;
void mvsf_default_options(MVSFOptions* o);

int mvsf_load(const char* path, mvsf_scene** out);
void mvsf_free(mvsf_scene* s);
int mvsf_version(const mvsf_scene* s);
int mvsf_num_images(const mvsf_scene* s);
int mvsf_num_points(const mvsf_scene* s);
/* name: image file name as stored (relative to the archive); w, h: the camera's resolution (0 if the archive stores none); valid: calibrated */
int mvsf_image_info(const mvsf_scene* s, int idx, char* name, int nameCap, int* w, int* h, int* valid);
/* point i: position and the (ascending) image indices that see it; views may be NULL to query the count */
int mvsf_point(const mvsf_scene* s, int i, float X[3], uint32_t* views, int viewsCap, int* nViews);
/* pixel camera of image idx at resolution w x h (0, 0 = the archive's): K, R row-major, C */
int mvsf_camera(const mvsf_scene* s, int idx, int w, int h, double K[9], double R[9], double C[3]);

/* DepthMapsData::SelectViews + the score cut of InitViews for image idx at the working resolutions sizes[2*i], sizes[2*i+1] of every image
 * (NULL = the archive's).  neighbors: up to cap entries in decreasing score; points: indices of the sparse points seen by idx in at least
 * max(2, nMinViewsTrustPoint) views.  Returns -3 if the image cannot be densified (too few points / neighbours). */
int mvsf_select_views(const mvsf_scene* s, int idx, const int* sizes, const MVSFOptions* opt, MVSFViewScore* neighbors, int cap, int* nNeighbors,
                      uint32_t* points, int pointsCap, int* nPoints, float* avgDepth);
/* The unfiltered list of Scene::SelectNeighborViews (all candidates, sorted), for inspection / tests. */
int mvsf_select_neighbor_views(const mvsf_scene* s, int idx, const int* sizes, uint32_t nMinViews, uint32_t nMinPointViews, float fOptimAngleDeg, uint32_t nInsideROI,
                               MVSFViewScore* neighbors, int cap, int* nNeighbors, uint32_t* points, int pointsCap, int* nPoints, float* avgDepth);
/* Neighbour lists the scene already carries.  An archive of version > 6 stores every image's view scores (MVS::Interface::Image::viewScores, libs/MVS/Interface.h:527-577;
 * Scene::LoadInterface copies them into Image::neighbors, libs/MVS/Scene.cpp:158), and `--view-neighbors-file` replaces them (Scene::LoadViewNeighbors, Scene.cpp:413-457:
 * one line per image, "<id> <neighbour-0> <neighbour-1> ...", best first; every listed neighbour becomes ViewScore{ID, 0 points, scale 1, angle 15 deg, area 0.5, score 3}).
 * DepthMapsData::SelectViews takes such a list as it is instead of scoring the views again (libs/MVS/SceneDensify.cpp:278-281) -- mvsf_select_views does the same; the image
 * then has no seed points, so InitViews starts its depth map from random values in [0.1, 100] (:418-427; mvsf_init_depth_map with nPoints = 0).
 * mvsf_load_view_neighbors: -2 = unreadable file or an image ID outside the scene (the reference asserts); lines starting with '#' and lines with fewer than two
 * numbers are skipped as there.  mvsf_save_view_neighbors writes Scene::SaveViewNeighbors' format (Scene.cpp:458-480): every image, "<id> <n0> <n1> ...\n". */
int mvsf_get_neighbors(const mvsf_scene* s, int idx, MVSFViewScore* neighbors, int cap, int* nNeighbors);
int mvsf_set_neighbors(mvsf_scene* s, int idx, const MVSFViewScore* neighbors, int nNeighbors);
int mvsf_load_view_neighbors(mvsf_scene* s, const char* path);
int mvsf_save_view_neighbors(const mvsf_scene* s, const char* path);
/* depth statistics an archive of version > 6 stores with the image (Interface.h:551-553; avgDepth is what the SGM path's corner support points use) */
int mvsf_image_depths(const mvsf_scene* s, int idx, float* minDepth, float* avgDepth, float* maxDepth);

/* Depth range [dMin, dMax] and seed maps (w*h depth, w*h*3 normal, zero where unknown) of image idx from `points`:
 * nMinViewsTrustPoint < 2: 5x5 splats with zero normals; else 2x2 splats with the area-weighted vertex normals of the Delaunay mesh of the
 * projections.  No points: dMin 0.1, dMax 100, empty maps. */
int mvsf_init_depth_map(const mvsf_scene* s, int idx, int w, int h, const uint32_t* points, int nPoints, uint32_t nMinViewsTrustPoint,
                        float* depthMap, float* normalMap, float* dMin, float* dMax);

/* The same with OPTDENSE::bInitSparse = 0 (DepthMap.cpp:1158-1190): every face of the Delaunay mesh rasterised (TImage::RasterizeTriangleBary,
 * libs/Common/Types.inl:2629-2669) with perspective-correct depth and interpolated vertex normal; pixels outside the mesh stay zero. */
int mvsf_init_depth_map_dense(const mvsf_scene* s, int idx, int w, int h, const uint32_t* points, int nPoints,
                              float* depthMap, float* normalMap, float* dMin, float* dMax);

/* The depth-only TriangulatePoints2DepthMap (DepthMap.cpp:1194-1251) that seeds the SGM path (SemiGlobalMatcher.cpp:608-618: corners, dense, at half the
 * first level's resolution).  addCorners: the four image corners join the mesh at a depth extrapolated from the faces next to them (DepthMap.cpp:1050-1107;
 * avgDepth is the image's average depth from mvsf_select_neighbor_views), so that the mesh covers the whole image.  dMin / dMax: depth bounds of the points. */
int mvsf_triangulate_depth_map(const mvsf_scene* s, int idx, int w, int h, const uint32_t* points, int nPoints, int addCorners, float avgDepth, int sparseOnly,
                               float* depthMap, float* dMin, float* dMax);

/* Normals of a depth map that has none (MVS::EstimateNormalMap, libs/MVS/DepthMap.cpp:1522-1613: what InitViews does with a .dmap stored without normals,
 * SceneDensify.cpp:411-414, FuseDepthMaps with such a map, :1427, and the SGM fuse mode, :2055): per pixel the least-squares depth gradient over the 8-neighbourhood
 * (neighbours with a depth within 3 % of the pixel's, at least three), then normalize((K00 dx, K11 dy, (K02 - x) dx + (K12 - y) dy - d)); zero where that fails.
 * K: 9 doubles row-major (used as floats, like the reference's Matrix3x3f); depth: w*h; normal: w*h*3. */
int mvsf_estimate_normal_map(const double K[9], const float* depth, int w, int h, float* normal);

#ifdef __cplusplus
}
#endif
#endif /* MVSFRONT_H_ */
