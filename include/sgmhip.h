/* sgmhip.h -- C ABI of the MI355X-native semi-global-matching kernels (libsgmhip.so).
 *
 * Boundary: the private hot method of the reference's SGM,
 *     void SemiGlobalMatcher::Match(const ViewData& left, const ViewData& right,
 *                                   DisparityMap& disparityMap, AccumCostMap& costMap)
 * (libs/MVS/SemiGlobalMatcher.h:170, body libs/MVS/SemiGlobalMatcher.cpp:863-1302), which works on
 * the members imagePixels / imageCosts / imageAccumCosts / maxNumDisp (SemiGlobalMatcher.h:198-201)
 * prepared by its caller Match(scene, idxImage, numNeighbors) (:530-737, reached from
 * SceneDensify.cpp:2048).  Those members become plain arguments here; everything around the call
 * (rectification, pyramid, range maps, cross-check, sub-pixel refinement) stays with the caller.
 * Integer outputs are bit-identical to the reference algorithm (u8 costs, u16 sums, i16 disparities).
 */
#ifndef SGMHIP_H_
#define SGMHIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SGMHIP_OK = 0, SGMHIP_E_ARG = -1, SGMHIP_E_HIP = -3, SGMHIP_E_NODEVICE = -5 };

typedef struct sgmhip_engine sgmhip_engine;

/* SemiGlobalMatcher::PixelData (SemiGlobalMatcher.h:79-82): 16 bytes, one per pixel of the valid
 * grid (w-6) x (h-6), row-major; range.minDisp >= range.maxDisp marks an invalid pixel. */
typedef struct SGMHipPixelData {
	uint64_t idx;            /* offset of this pixel's first cost in the ragged cost/accum arrays */
	int16_t minDisp, maxDisp;
	int32_t pad_;
} SGMHipPixelData;

int sgmhip_create(int device, sgmhip_engine** out);
void sgmhip_destroy(sgmhip_engine* e);
const char* sgmhip_last_error(sgmhip_engine* e);

/* GenerateP2s (SemiGlobalMatcher.cpp:518-524): P2s[i] = round(P2*(1+alpha*exp(-i^2/(2 beta^2)))). */
int sgmhip_generate_p2s(uint16_t P2, float alpha, float beta, uint16_t out256[256]);

/* Upload one stereo problem (host pointers): left colour BGR u8 (w*h*3), left/right gray float
 * (w*h), pixel table ((w-6)*(h-6)), total number of costs, max disparities per pixel. */
int sgmhip_set_problem(sgmhip_engine* e, const uint8_t* leftBGR, const float* leftGray, const float* rightGray,
                       int w, int h, const SGMHipPixelData* pixels, uint64_t numCosts, int maxNumDisp);
/* SemiGlobalMatcher::Match(ViewData,ViewData,...): cost volume + 8-path aggregation + WTA on the
 * resident problem.  Asynchronous unless sync != 0.  P1 / P2s as in the SemiGlobalMatcher ctor. */
int sgmhip_match(sgmhip_engine* e, uint16_t P1, const uint16_t P2s[256], int sync);
/* Download results (any pointer may be NULL): disparity i16 and cost u16 per valid-grid pixel;
 * the raw cost volume (u8) and the 8-path sums (u16), numCosts entries each. */
int sgmhip_get_results(sgmhip_engine* e, int16_t* disparity, uint16_t* cost, uint8_t* costs, uint16_t* accums);
int sgmhip_sync(sgmhip_engine* e);
/* Kernel mapping of sgmhip_match: 0 (default) = one wavefront per pixel / line, one lane per disparity (plain SGM, D = 64 .. 256);
 * 8 / 16 / 32 (1 = 16) = sub-groups of that many lanes with a loop over the range, for the narrow ragged ranges of the tSGM loop
 * (csrc/sgm_kernels_sub.hip).  Same results. */
int sgmhip_set_sub_group_kernels(sgmhip_engine* e, int lanes);

/* ---- the steps of the tSGM loop around Match (SemiGlobalMatcher.cpp:1449-1811); disparity maps int16 with NO_DISP = 32767,
 * masks uint8 with INVALID = 0 / VALID = 255, all host pointers, row-major ------------------------------------------------ */
/* ConsistencyCrossCheck(l2r, r2l, thCross) (:1449-1489): l2r (wl x h) is filtered in place against r2l (wr x h). */
int sgmhip_consistency_cross_check(sgmhip_engine* e, int16_t* l2r, const int16_t* r2l, int wl, int h, int wr, int thCross);
/* FilterByCost (:1491-1514): disparities whose cost exceeds th become NO_DISP. */
int sgmhip_filter_by_cost(sgmhip_engine* e, int16_t* disparity, const uint16_t* cost, int w, int h, uint16_t th);
/* ExtractMask (:1516-1573, thValid default 3): per row, from both ends, pixels are marked INVALID until thValid valid disparities were met.
 * initValid != 0: the mask starts all VALID (the reference creates it when its size differs), else `mask` is in/out. */
int sgmhip_extract_mask(sgmhip_engine* e, const int16_t* disparity, uint8_t* mask, int w, int h, int thValid, int initValid);
/* UpscaleMask (:1657-1690): mask (w x h) -> mask2x (w2 x h2): pixel (r,c) covers the 2x2 block at (2r+3, 2c+3); the rest is INVALID. */
int sgmhip_upscale_mask(sgmhip_engine* e, const uint8_t* mask, int w, int h, uint8_t* mask2x, int w2, int h2);
/* FlipDirection (:1628-1655): r2l(r, c+d-1 .. c+d+1) = -l2r(r,c), later columns overwrite earlier ones; elsewhere NO_DISP. */
int sgmhip_flip_direction(sgmhip_engine* e, const int16_t* l2r, int w, int h, int16_t* r2l);
/* RefineDisparityMap (:1693-1811) on the resident result of the last sgmhip_match, using its 8-path sums in place on the device:
 * subpixelMode 0 NA, 1 LINEAR, 2 POLY4, 3 PARABOLA, 4 SINE, 5 COSINE, 6 LC_BLEND (the reference's default with subpixelSteps 4).
 * Fetch the result with sgmhip_get_results.  cos/sin come from csrc/pm_math.h (Cephes kernels), not libm. */
int sgmhip_refine_disparity(sgmhip_engine* e, int subpixelMode, int subpixelSteps);
/* Replace the resident disparity map of the last sgmhip_match (valid-grid size) with a post-processed one -- the tSGM loop refines the
 * cross-checked map, not the raw winner-take-all result (:693-699). */
int sgmhip_set_disparity(sgmhip_engine* e, const int16_t* disparity);

/* Disparity2RangeMap (:1350-1444): from the previous level's disparity map (w x h) and the 2x mask (w2 x h2, w2 > 2w+3, h2 >= 2h+3), the pixel
 * table of the next level: per 2x pixel the search range around the median of the valid disparities in a 7x7 window (41x41 at holes) and the
 * running cost index.  minNumDisp / minNumDispInvalid: 11 / 33 on the first level, 5 / 7 afterwards (:621,642).  pixels: w2*h2 entries. */
int sgmhip_disparity2range_map(sgmhip_engine* e, const int16_t* disparity, int w, int h, const uint8_t* mask2x, int w2, int h2,
                               int minNumDisp, int minNumDispInvalid, SGMHipPixelData* pixels, uint64_t* numCosts, int* maxNumDisp);
/* Depth2DisparityMap (:1836-1860): depth map of the un-rectified image (dw x dh) -> disparity map of the rectified image (valid size w x h);
 * invH 3x3 and invQ 4x4 row-major doubles (Image::StereoRectifyImages' H and Q, inverted). */
int sgmhip_depth2disparity_map(sgmhip_engine* e, const float* depthMap, int dw, int dh, const double invH[9], const double invQ[16], int subpixelSteps,
                               int16_t* disparity, int w, int h);
/* Disparity2DepthMap (:1862-1923): disparity (+ optional cost) map of the rectified image (w x h) -> depth (+ confidence = 1/(cost+1)) map of the
 * un-rectified image (dw x dh). */
int sgmhip_disparity2depth_map(sgmhip_engine* e, const int16_t* disparity, const uint16_t* cost, int w, int h, const double H[9], const double Q[16],
                               int subpixelSteps, float* depthMap, float* confMap, int dw, int dh);

/* ProjectDisparity2DepthMap (:1925-2039): forward-projects every valid disparity into the un-rectified image (dw x dh), keeps per pixel and quadrant
 * the nearest projection and blends the (up to four) agreeing ones; depthRangeMap: 2 floats per pixel (depth at disparity -1 / +1), confMap only
 * with a cost map.  Where depthMap is 0 the other two maps are 0 (uninitialised in the reference).  *anyDepth = the reference's return value. */
int sgmhip_project_disparity2depth_map(sgmhip_engine* e, const int16_t* disparity, const uint16_t* cost, int w, int h, const double Q[16], int subpixelSteps,
                                       float* depthMap, float* depthRangeMap, float* confMap, int dw, int dh, int* anyDepth);
/* The per-pixel fusion of SemiGlobalMatcher::Fuse (:797-849) over nPairs (<= 32) pair maps produced by the call above: depths whose trust ranges
 * contain each other are clustered, the largest cluster (>= minViews members) is averaged. */
int sgmhip_fuse_pairs(sgmhip_engine* e, const float* const* depthMaps, const float* const* depthRangeMaps, const float* const* confMaps, int nPairs,
                      int dw, int dh, unsigned minViews, float* depthMap, float* confMap);

/* SemiGlobalMatcher::Fuse (:738-859) in one call for nPairs <= 16 pairs that have the reference image on the left: every pair's disparity / cost maps (valid grid
 * widths[p] x heights[p], Q = Qs + 16 p, subpixelSteps[p] -- the content of its .dimap) are projected into the reference image (dw x dh) and fused per pixel;
 * pairs that produce no depth are dropped; nUsed (nullable) = pairs that took part.  The per-pair depth / range / confidence maps stay on the device. */
int sgmhip_fuse_disparities(sgmhip_engine* e, int nPairs, const int16_t* const* disparities, const uint16_t* const* costs, const int* widths, const int* heights,
                            const double* Qs, const int* subpixelSteps, int dw, int dh, unsigned minViews, float* depthMap, float* confMap, int* nUsed);

/* cv::filterSpeckles(disparityMap, NO_DISP, maxSpeckleSize, maxDiff) as the tSGM loop applies it on the first level (:687-688; OPTDENSE::nSpeckleSize,
 * 5): 4-connected regions of disparities differing by <= maxDiff step to step that have at most maxSpeckleSize pixels become NO_DISP. */
int sgmhip_filter_speckles(sgmhip_engine* e, int16_t* disparity, int w, int h, int maxSpeckleSize, int maxDiff);

/* HIP-event timing since the last reset: milliseconds in the cost-volume, aggregation (8 path
 * kernels) and WTA kernels, number of match calls. */
typedef struct SGMHipStats { double costMs, aggrMs, wtaMs; uint64_t calls, aggrLaunches; } SGMHipStats;
/* The tSGM coarse-to-fine loop of SemiGlobalMatcher::Match(scene, ...) (libs/MVS/SemiGlobalMatcher.cpp:577-706) for one rectified pair in one call, resident
 * on the device: per level the image pyramids (INTER_AREA from the full-resolution images), FlipDirection + Disparity2RangeMap, Match right->left and
 * left->right, ConsistencyCrossCheck; on the first level also filterSpeckles(nSpeckleSize, 5) and ExtractMask; finally RefineDisparityMap.  Equal to driving the
 * single steps above as openmvs_amd/tsgm.py does.  Images: w x h (a multiple of 2^levels), BGR 8-bit and gray float of the rectified pair, their 8-bit validity
 * masks (Image::StereoRectifyImages); initLeftDisparity: nullable, the half-resolution map of the first level (Depth2DisparityMap of the sparse-point depth map,
 * :608-625), size (round(w_l/2)-6) x (round(h_l/2)-6) with w_l, h_l the coarsest level's size.  Outputs on the valid grid (w-6) x (h-6). */
int sgmhip_tsgm_match(sgmhip_engine* e, const uint8_t* leftBGR, const uint8_t* rightBGR, const float* leftGray, const float* rightGray,
                      const uint8_t* leftMask, const uint8_t* rightMask, int w, int h, unsigned minResolution, const int16_t* initLeftDisparity,
                      int nSpeckleSize, int subpixelMode, int subpixelSteps, uint16_t P1, const uint16_t P2s[256], int16_t* disparity, uint16_t* cost, int* numLevels);

int sgmhip_stats_reset(sgmhip_engine* e, int enable);
int sgmhip_stats_get(sgmhip_engine* e, SGMHipStats* out);

#ifdef __cplusplus
}
#endif
#endif /* SGMHIP_H_ */
