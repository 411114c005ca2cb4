// ref_sgm_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  The reference's SemiGlobalMatcher::Match(ViewData, ViewData, ...) --
// cost volume, 8-path aggregation (the threaded variant, which is the parity target) and winner-take-all, SemiGlobalMatcher.cpp:863-1302 -- cut
// verbatim from /root/reference by oracle/ref/build_ref.py and compiled against oracle/ref/shim.  The worker-thread pool is replaced by a stand-in
// that runs every queued job at once on the calling thread (one "worker"): the jobs of a pass only add into disjoint or commutative sums.
#include <functional>
#include "seacave_min.h"
#define DECLARE_NO_INDEX(T) std::numeric_limits<T>::max()
namespace SEACAVE {                          // (ROUND2INT / Round2Int: the reference's own text since round 3, seacave_min.h <- Types.h:916-963)
template <typename TYPE> class TPixel { public: union { struct { TYPE b, g, r; }; TYPE c[3]; };             // Types.h:1873-1895 (_COLORMODE BGR)
	inline const TYPE& operator[](size_t i) const { return c[i]; } inline TYPE& operator[](size_t i) { return c[i]; } };
typedef TPixel<uint8_t> Pixel8U;
typedef TImage<Pixel8U> Image8U3;
class Event { public: Event(uint32_t) {} virtual ~Event() {} virtual bool Run(void* = NULL) { return true; } };
struct Semaphore {};
class EventThreadPool {                    // one worker that runs a job the moment it is queued
public:
	typedef size_t size_type;
	size_t n = 1;
	bool empty() const { return n == 0; } bool IsEmpty() const { return true; } size_t size() const { return n; }
	void AddEvent(Event* e) { e->Run(NULL); delete e; }
};
inline Thread::safe_t safeDecShim(volatile Thread::safe_t& v) { return --v; }
}
namespace MVS { class Scene; class PointCloud;
template <int nTexels> struct WeightedPatchFix { struct Pixel { float weight; float tempWeight; }; Pixel weights[nTexels]; float sumWeights; float normSq0; WeightedPatchFix() : normSq0(0) {} };   // DepthMap.h:143-153
#include "snip/sgm_h_defines.inc"          // SemiGlobalMatcher.h:44-46
namespace STEREO {
#include "snip/sgm_h_class.inc"            // SemiGlobalMatcher.h:57-204: class SemiGlobalMatcher
} }
using namespace MVS; using namespace MVS::STEREO;
struct ThreadX : Thread { static inline safe_t safeDec(volatile safe_t& v) { return --v; } };
#define Thread ThreadX
#include "snip/sgm_cpp_events.inc"         // SemiGlobalMatcher.cpp:436-492: EVTPixelProcess, EVTPixelAccumInc, EVTPixelAccumDec
#include "snip/sgm_cpp_ctor.inc"           // SemiGlobalMatcher.cpp:513-529: constructor, destructor, GenerateP2s
#include "snip/sgm_cpp_match.inc"          // SemiGlobalMatcher.cpp:863-1302: Match
#include "snip/sgm_cpp_post.inc"           // SemiGlobalMatcher.cpp:1446-1811: ConsistencyCrossCheck, FilterByCost, ExtractMask, FlipDirection, UpscaleMask, RefineDisparityMap
#include "snip/image_cpp_disp2depth.inc"   // Image.cpp:372-412: TDisparity2Depth, Image::Disparity2Depth (four overloads)
#include "snip/image_cpp_depth2disp.inc"   // Image.cpp:423-433: Image::Depth2Disparity
#include "snip/sgm_cpp_range.inc"          // SemiGlobalMatcher.cpp:1350-1444: Disparity2RangeMap
#include "snip/sgm_cpp_conv.inc"           // SemiGlobalMatcher.cpp:1836-2039: Depth2DisparityMap, Disparity2DepthMap, ProjectDisparity2DepthMap
#undef Thread
EventThreadPool SemiGlobalMatcher::threads;
Semaphore SemiGlobalMatcher::sem;
void SemiGlobalMatcher::WaitThreadWorkers(unsigned) {}

namespace {
struct Access : SemiGlobalMatcher {       // reaches the protected members Match() works on
	using SemiGlobalMatcher::Match; using SemiGlobalMatcher::imagePixels; using SemiGlobalMatcher::imageCosts; using SemiGlobalMatcher::imageAccumCosts;
	using SemiGlobalMatcher::maxNumDisp; using SemiGlobalMatcher::P1; using SemiGlobalMatcher::P2s; using SemiGlobalMatcher::GenerateP2s;
	using SemiGlobalMatcher::ConsistencyCrossCheck; using SemiGlobalMatcher::FilterByCost; using SemiGlobalMatcher::ExtractMask; using SemiGlobalMatcher::UpscaleMask;
	using SemiGlobalMatcher::FlipDirection; using SemiGlobalMatcher::RefineDisparityMap;
	using SemiGlobalMatcher::Disparity2RangeMap; using SemiGlobalMatcher::Depth2DisparityMap; using SemiGlobalMatcher::Disparity2DepthMap; using SemiGlobalMatcher::ProjectDisparity2DepthMap;
	Access(SgmSubpixelMode m = SUBPIXEL_LC_BLEND, Disparity steps = 4) : SemiGlobalMatcher(m, steps) {}
};
template <typename T> static TImage<T> imageOf(const T* p, int w, int h) { TImage<T> m; m.create(cv::Size(w, h)); memcpy(m.data(), p, sizeof(T) * (size_t)w * h); return m; }
}
extern "C" {
void ref_sgm_generate_p2s(uint16_t P2, float alpha, float beta, uint16_t* out256) {
	const auto p = Access::GenerateP2s(P2, alpha, beta);
	for (int i = 0; i < 256; ++i) out256[i] = p[i];
}
// The steps around Match (SemiGlobalMatcher.cpp:1446-1811), same arguments as the orc_sgm_* functions of oracle/sgm_post_oracle.cpp
void ref_sgm_cross_check(int16_t* l2r, const int16_t* r2l, int wl, int h, int wr, int thCross) {
	SemiGlobalMatcher::DisparityMap a = imageOf(l2r, wl, h), b = imageOf(r2l, wr, h);
	Access::ConsistencyCrossCheck(a, b, (SemiGlobalMatcher::Disparity)thCross);
	memcpy(l2r, a.data(), sizeof(int16_t) * (size_t)wl * h);
}
void ref_sgm_filter_by_cost(int16_t* disp, const uint16_t* cost, int w, int h, uint16_t th) {
	SemiGlobalMatcher::DisparityMap a = imageOf(disp, w, h); SemiGlobalMatcher::AccumCostMap c = imageOf(cost, w, h);
	Access::FilterByCost(a, c, th);
	memcpy(disp, a.data(), sizeof(int16_t) * (size_t)w * h);
}
void ref_sgm_extract_mask(const int16_t* disp, uint8_t* mask, int w, int h, int thValid, int initValid) {
	SemiGlobalMatcher::DisparityMap a = imageOf(disp, w, h);
	SemiGlobalMatcher::MaskMap m; if (!initValid) m = imageOf(mask, w, h);      // an empty mask is created all-VALID by the function (:1521-1525)
	Access::ExtractMask(a, m, thValid);
	memcpy(mask, m.data(), (size_t)w * h);
}
void ref_sgm_upscale_mask(const uint8_t* mask, int w, int h, uint8_t* mask2x, int w2, int h2) {
	SemiGlobalMatcher::MaskMap m = imageOf(mask, w, h);
	Access::UpscaleMask(m, cv::Size(w2, h2));
	memcpy(mask2x, m.data(), (size_t)w2 * h2);
}
void ref_sgm_flip_direction(const int16_t* l2r, int w, int h, int16_t* r2l) {
	SemiGlobalMatcher::DisparityMap a = imageOf(l2r, w, h), b;
	Access::FlipDirection(a, b);
	memcpy(r2l, b.data(), sizeof(int16_t) * (size_t)w * h);
}
// RefineDisparityMap works on the matcher's pixel table and 8-path sums: pixels = {u64 idx; i16 min, max; pad} per valid-grid pixel (row-major vw x vh), accums = the sums
void ref_sgm_refine_wh(int16_t* disp, const void* pixels, const uint16_t* accums, uint64_t numCosts, int vw, int vh, int mode, int steps) {
	Access m((SemiGlobalMatcher::SgmSubpixelMode)mode, (SemiGlobalMatcher::Disparity)steps);
	m.imagePixels.Resize((SemiGlobalMatcher::Index)vw * vh);
	memcpy(m.imagePixels.Begin(), pixels, 16 * (size_t)vw * vh);
	m.imageAccumCosts.Resize(numCosts); memcpy(m.imageAccumCosts.Begin(), accums, numCosts * 2);
	SemiGlobalMatcher::DisparityMap a = imageOf(disp, vw, vh);
	m.RefineDisparityMap(a);
	memcpy(disp, a.data(), sizeof(int16_t) * (size_t)vw * vh);
}
// Disparity2RangeMap, Depth2DisparityMap, Disparity2DepthMap, ProjectDisparity2DepthMap (SemiGlobalMatcher.cpp:1350-1444, :1836-2039): same arguments as the orc_sgm_* functions
unsigned long long ref_sgm_disparity2range_map(const int16_t* disparityMap, int cols, int rows, const uint8_t* maskMap, int w2, int h2, int minNumDisp, int minNumDispInvalid,
		void* imagePixels, int* outMaxNumDisp) {
	Access m;
	const SemiGlobalMatcher::DisparityMap d = imageOf(disparityMap, cols, rows); const SemiGlobalMatcher::MaskMap k = imageOf(maskMap, w2, h2);
	const SemiGlobalMatcher::Index n = m.Disparity2RangeMap(d, k, (SemiGlobalMatcher::Disparity)minNumDisp, (SemiGlobalMatcher::Disparity)minNumDispInvalid);
	memcpy(imagePixels, m.imagePixels.Begin(), 16 * (size_t)w2 * h2);
	*outMaxNumDisp = m.maxNumDisp;
	return n;
}
// SemiGlobalMatcher::Fuse, the per-pixel cluster fusion (SemiGlobalMatcher.cpp:795-848 verbatim; the part before it loads the pair files and calls ProjectDisparity2DepthMap,
// which ref_sgm_project_disparity2depth_map covers): same arguments as orc_sgm_fuse_pairs (oracle/sgm_post_oracle.cpp)
void ref_sgm_fuse_pairs(const float* const* depthMaps, const float* const* rangeMaps, const float* const* confMaps, int nPairs, int dw, int dh, unsigned minViews,
		float* outDepth, float* outConf) {
	typedef SemiGlobalMatcher::DepthRange DepthRange; typedef SemiGlobalMatcher::DepthRangeMap DepthRangeMap;
	#include "snip/sgm_cpp_fuse_pairdata.inc"   // :744-749
	CLISTDEFIDX(PairData,IIndex) pairs;
	pairs.reserve((IIndex)nPairs);
	for (int p = 0; p < nPairs; ++p) {
		PairData& pair = pairs.emplace_back(cv::Size(dw, dh));
		memcpy(pair.depthMap.data(), depthMaps[p], sizeof(float) * (size_t)dw * dh);
		pair.depthRangeMap.create(cv::Size(dw, dh)); memcpy(pair.depthRangeMap.data(), rangeMaps[p], sizeof(float) * 2 * (size_t)dw * dh);
		pair.confMap.create(cv::Size(dw, dh)); memcpy(pair.confMap.data(), confMaps[p], sizeof(float) * (size_t)dw * dh);
	}
	struct { struct { cv::Size sz; cv::Size size() const { return sz; } } image; } leftImage; leftImage.image.sz = cv::Size(dw, dh);
	DepthMap depthMap; ConfidenceMap confMap;
	#include "snip/sgm_cpp_fuse_loop.inc"       // :795-848
	memcpy(outDepth, depthMap.data(), sizeof(float) * (size_t)dw * dh); memcpy(outConf, confMap.data(), sizeof(float) * (size_t)dw * dh);
}
static Matrix3x3 m33(const double* p) { Matrix3x3 M; for (int i = 0; i < 9; ++i) M.val[i] = p[i]; return M; }
static Matrix4x4 m44(const double* p) { Matrix4x4 M; for (int i = 0; i < 16; ++i) M.val[i] = p[i]; return M; }
void ref_sgm_depth2disparity_map(const float* depthMap, int dw, int dh, const double* invH, const double* invQ, int subpixelSteps, int16_t* disparityMap, int w, int h) {
	const DepthMap dm = imageOf(depthMap, dw, dh);
	SemiGlobalMatcher::DisparityMap out; out.create(cv::Size(w, h));
	Access::Depth2DisparityMap(dm, m33(invH), m44(invQ), (SemiGlobalMatcher::Disparity)subpixelSteps, out);
	memcpy(disparityMap, out.data(), sizeof(int16_t) * (size_t)w * h);
}
void ref_sgm_disparity2depth_map(const int16_t* disparityMap, const uint16_t* costMap, int w, int h, const double* H, const double* Q, int subpixelSteps, float* depthMap, float* confMap, int dw, int dh) {
	const SemiGlobalMatcher::DisparityMap d = imageOf(disparityMap, w, h);
	SemiGlobalMatcher::AccumCostMap c; if (costMap) c = imageOf(costMap, w, h);
	DepthMap dm; dm.create(cv::Size(dw, dh)); ConfidenceMap cm;
	Access::Disparity2DepthMap(d, c, m33(H), m44(Q), (SemiGlobalMatcher::Disparity)subpixelSteps, dm, cm);
	memcpy(depthMap, dm.data(), sizeof(float) * (size_t)dw * dh);
	if (costMap) memcpy(confMap, cm.data(), sizeof(float) * (size_t)dw * dh);
}
int ref_sgm_project_disparity2depth_map(const int16_t* disparityMap, const uint16_t* costMap, int w, int h, const double* Q, int subpixelSteps, float* depthMap, float* depthRangeMap, float* confMap, int dw, int dh) {
	const SemiGlobalMatcher::DisparityMap d = imageOf(disparityMap, w, h);
	SemiGlobalMatcher::AccumCostMap c; if (costMap) c = imageOf(costMap, w, h);
	DepthMap dm; dm.create(cv::Size(dw, dh)); SemiGlobalMatcher::DepthRangeMap rm; ConfidenceMap cm;
	const bool ok = Access::ProjectDisparity2DepthMap(d, c, m44(Q), (SemiGlobalMatcher::Disparity)subpixelSteps, dm, rm, cm);
	memcpy(depthMap, dm.data(), sizeof(float) * (size_t)dw * dh);
	memcpy(depthRangeMap, rm.data(), sizeof(float) * 2 * (size_t)dw * dh);
	if (costMap) memcpy(confMap, cm.data(), sizeof(float) * (size_t)dw * dh);
	return ok ? 1 : 0;
}
// same arguments as orc_sgm_match (oracle/sgm_oracle.cpp); pixels: (w-6)*(h-6) entries {u64 idx; i16 min, max; pad}
int ref_sgm_match(const uint8_t* colorL, const float* grayL, const float* grayR, int w, int h, const void* pixels, uint64_t numCosts, int maxNumDisp, uint16_t P1, const uint16_t* P2s,
		int16_t* disparity, uint16_t* cost, uint8_t* costsOut, uint16_t* accumsOut) {
	struct PD { uint64_t idx; int16_t mn, mx; };
	static_assert(sizeof(PD) == 16 && sizeof(SemiGlobalMatcher::PixelData) == 16, "PixelData layout");
	Access m;
	m.P1 = P1; for (int i = 0; i < 256; ++i) m.P2s[i] = P2s[i];
	const int vw = w - 6, vh = h - 6;
	m.imagePixels.Resize((SemiGlobalMatcher::Index)vw * vh);
	memcpy(m.imagePixels.Begin(), pixels, sizeof(PD) * (size_t)vw * vh);
	m.imageCosts.Resize(numCosts); m.imageAccumCosts.Resize(numCosts); m.maxNumDisp = (SemiGlobalMatcher::Disparity)maxNumDisp;
	SemiGlobalMatcher::ViewData L, R;
	L.imageColor.create(cv::Size(w, h)); memcpy(L.imageColor.data(), colorL, (size_t)w * h * 3);
	L.imageGray.create(cv::Size(w, h)); memcpy(L.imageGray.data(), grayL, sizeof(float) * (size_t)w * h);
	R.imageGray.create(cv::Size(w, h)); memcpy(R.imageGray.data(), grayR, sizeof(float) * (size_t)w * h);
	SemiGlobalMatcher::DisparityMap d; SemiGlobalMatcher::AccumCostMap c;
	m.Match(L, R, d, c);
	memcpy(disparity, d.data(), sizeof(int16_t) * (size_t)vw * vh); memcpy(cost, c.data(), sizeof(uint16_t) * (size_t)vw * vh);
	if (costsOut) memcpy(costsOut, m.imageCosts.Begin(), numCosts);
	if (accumsOut) memcpy(accumsOut, m.imageAccumCosts.Begin(), numCosts * 2);
	return 0;
}
}
