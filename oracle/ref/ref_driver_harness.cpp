// ref_driver_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  The reference's own multi-level driver: DepthMapsData::ScaleDepthData and
// DepthMapsData::EstimateDepthMap (libs/MVS/SceneDensify.cpp:578-601, :616-805) cut VERBATIM from /root/reference by oracle/ref/build_ref.py, compiled against
// oracle/ref/shim together with the verbatim estimator (DepthMap.cpp:325-972) and pass bodies (SceneDensify.cpp:489-576), and driven through a C interface
// shaped like oracle/pm_oracle.cpp's orc_estimate_depth_map(_masked).  What the reference's text decides here: the level loop, the threshold schedule
// (:774-776), the prior hand-off (:660-664), nearest / linear choice (:661), the thread scaffolding (one estimator per thread on the shared idxPixel,
// :631-750; scene.nMaxThreads threads -- this is also the multi-threaded CPU baseline of bench.py, cpu_baseline.kind = "reference").
// What is NOT the reference's: cv::resize.  OpenCV is an un-vendored, unpinned dependency; the shim's cv::resize below dispatches to the
// oracle's restatement of OpenCV's resamplers (orc_resize_* of libpm_oracle.so: integer-factor INTER_AREA, INTER_LINEAR) and to OpenCV's INTER_NEAREST
// index rule written out here.  The mask file loader (DepthEstimator::ImportIgnoreMask, DepthMap.cpp:300-323) is replaced by a loader from the caller's array.
#define REF_DRIVER 1
#include "seacave_min.h"
#include "snip/depthmap_h.inc"            // libs/MVS/DepthMap.h:41-468 (opens namespace MVS; closed right below)
} // namespace MVS

namespace MVS { namespace OPTDENSE {      // libs/MVS/DepthMap.cpp:69-114 (defaults); set per call from the caller's options
unsigned nResolutionLevel = 1, nMaxResolution = 2560, nMinResolution = 640, nSubResolutionLevels = 2, nMinViews = 2, nMaxViews = 12, nMinViewsFuse = 2, nMinViewsFilter = 2,
	nMinViewsFilterAdjust = 1, nMinViewsTrustPoint = 2, nNumViews = 5, nPointInsideROI = 2;
bool bFilterAdjust = true, bAddCorners = false, bInitSparse = true, bRemoveDmaps = false;
float fViewMinScore = 2.f, fViewMinScoreRatio = 0.03f, fMinArea = 0.05f, fMinAngle = 3.f, fOptimAngle = 12.f, fMaxAngle = 65.f, fDescriptorMinMagnitudeThreshold = 0.02f,
	fDepthDiffThreshold = 0.01f, fNormalDiffThreshold = 25.f, fPairwiseMul = 0.3f, fOptimizerEps = 0.001f;
int nOptimizerMaxIters = 80;
unsigned nSpeckleSize = 100, nIpolGapSize = 7;
int nIgnoreMaskLabel = -1;
unsigned nOptimize = 7, nEstimateColors = 2, nEstimateNormals = 2;
float fNCCThresholdKeep = 0.9f;
unsigned nEstimationIters = 3, nEstimationGeometricIters = 2;
float fEstimationGeometricWeight = 0.1f;
unsigned nRandomIters = 6, nRandomMaxScale = 2;
float fRandomDepthRatio = 0.003f, fRandomAngle1Range = 16.f, fRandomAngle2Range = 10.f, fRandomSmoothDepth = 0.02f, fRandomSmoothNormal = 13.f, fRandomSmoothBonus = 0.93f;
} }
using namespace MVS;

// ---- cv::resize: OpenCV's dispatch (imgproc/resize.cpp: dsize from the factors with cvRound when empty, factors from dsize otherwise; a fresh destination, so
// in-place calls are fine) onto the oracle's restated resamplers -----------------------------------------------------------------------------------------------
extern "C" {
void orc_resize_area(const float* s, int w, int h, int f, float* o);
void orc_resize_linear(const float* s, int w, int h, int nw, int nh, float* o);
int orc_scaled_size(int n, int f);
}
namespace cv {
template <class A, class B> void resize(const A& src, B& dst, Size dsize, double fx, double fy, int interpolation) {
	typedef typename A::Type T;
	const int sw = src.cols, sh = src.rows;
	if (dsize.empty()) dsize = Size(saturate_cast<int>(sw * fx), saturate_cast<int>(sh * fy));
	else { fx = (double)dsize.width / sw; fy = (double)dsize.height / sh; }
	B out; out.create(dsize);
	if (interpolation == INTER_NEAREST) {
		// resizeNN: sx = min(cvFloor(x * ifx), ssize.width - 1), ifx = 1 / fx
		const double ifx = 1. / fx, ify = 1. / fy;
		for (int y = 0; y < dsize.height; ++y) {
			const int sy = std::min((int)std::floor(y * ify), sh - 1);
			for (int x = 0; x < dsize.width; ++x) { const int sx = std::min((int)std::floor(x * ifx), sw - 1); out(y, x) = src(sy, sx); }
		}
	} else if constexpr (std::is_same<T, float>::value) {
		if (interpolation == INTER_AREA) {
			// the reference only shrinks by 1 / 2^k (ScaleDepthData): the integer-factor INTER_AREA of the oracle; its size rule must agree with OpenCV's cvRound above
			const int f = (int)std::lrint(1. / fx);
			if (f < 1 || orc_scaled_size(sw, f) != dsize.width || orc_scaled_size(sh, f) != dsize.height) abort();
			orc_resize_area(const_cast<A&>(src).data(), sw, sh, f, out.data());
		} else if (interpolation == INTER_LINEAR) orc_resize_linear(const_cast<A&>(src).data(), sw, sh, dsize.width, dsize.height, out.data());
		else abort();
	} else abort();
	dst = out;
}
}

// the level-0 ignore mask of the call in flight (nullable; 0 = ignored pixel), read by the stand-in for the mask file loader
static const unsigned char* g_ignoreMask = nullptr; static int g_maskW = 0, g_maskH = 0;
#include "snip/depthmap_cpp_copy.inc"     // libs/MVS/DepthMap.cpp:121-133: DepthData's copy constructor
#include "snip/depthmap_cpp_applymask.inc" // libs/MVS/DepthMap.cpp:214-231: DepthData::ApplyIgnoreMask
// DepthEstimator::ImportIgnoreMask (DepthMap.cpp:300-323) with the PNG loader replaced: cv::resize(mask, size, INTER_NEAREST), then a bit per pixel != label
bool DepthEstimator::ImportIgnoreMask(const Image&, const Image8U::Size& size, uint16_t, BitMatrix& bmask, Image8U*) {
	if (!g_ignoreMask) return false;
	Image8U m0(cv::Size(g_maskW, g_maskH)); memcpy(m0.data(), g_ignoreMask, (size_t)g_maskW * g_maskH);
	Image8U m; cv::resize(m0, m, size, 0, 0, cv::INTER_NEAREST);
	bmask.create(size.width, size.height);
	for (int r = 0; r < size.height; ++r) for (int c = 0; c < size.width; ++c) bmask.bits[(size_t)r * size.width + c] = m(r, c) ? 1 : 0;
	return true;
}
#include "snip/depthmap_cpp.inc"          // libs/MVS/DepthMap.cpp:325-972: MapMatrix2ZigzagIdx, the constructor, PreparePixelPatch ... InitPlane

namespace MVS {
class DepthMapsData {                     // SceneDensify.h:54-93: the members the driver touches
public:
	bool EstimateDepthMap(IIndex idxImage, int nGeometricIter);
	static DepthData ScaleDepthData(const DepthData& inputDeptData, float scale);
	static void* STCALL ScoreDepthMapTmp(void*);
	static void* STCALL EstimateDepthMapTmp(void*);
	static void* STCALL EndDepthMapTmp(void*);
	struct { unsigned nMaxThreads; } scene;
	DepthDataArr arrDepthData;
	Image8U::Size prevDepthMapSize;
	DepthEstimator::MapRefArr coords;
};
}
#define TD_TIMER_STARTD() ((void)0)
#define TD_TIMER_GET_FMT() String()
#define DEBUG_EXTRA(...) ((void)0)
#include "snip/scenedensify_cpp.inc"      // libs/MVS/SceneDensify.cpp:489-576: the three pass bodies
#include "snip/scenedensify_scale.inc"    // libs/MVS/SceneDensify.cpp:578-601: ScaleDepthData
#include "snip/scenedensify_estimate.inc" // libs/MVS/SceneDensify.cpp:616-805: EstimateDepthMap

extern "C" {
struct OrcView {                          // same layout as oracle/pm_oracle.cpp's
	const float* image; int w, h;
	double K[9], R[9], C[3];
	const float* depth;
	double Kd[9], Rd[9], Cd[3];
	int dw, dh;
};
struct OrcOpt {
	uint32_t nSubResolutionLevels, nEstimationIters, nEstimationGeometricIters, nRandomIters;
	float fEstimationGeometricWeight, fRandomDepthRatio, fRandomAngle1Range, fRandomAngle2Range;
	float fRandomSmoothDepth, fRandomSmoothNormal, fRandomSmoothBonus, fNCCThresholdKeep, fDescriptorMinMagnitudeThreshold;
	uint32_t seed, viewID; int32_t rngMode, nThreads;
};
static void setCamera(Camera& c, const double* K, const double* R, const double* C) {
	for (int i = 0; i < 9; ++i) { c.K.val[i] = K[i]; c.R.val[i] = R[i]; }
	c.C.x = C[0]; c.C.y = C[1]; c.C.z = C[2];
}
// One DepthMapsData::EstimateDepthMap call (the reference's text) for views[0] with sources views[1..]: depth / normal in and out (zero = unset), conf out.
// nGeometricIter < 0: photometric pass over the pyramid; >= 0: one geometric round (views[i].depth = the neighbours' maps).  opt->nThreads = scene.nMaxThreads.
// mask (nullable, image resolution, 0 = ignore) with maskMode != 0 = OPTDENSE::nIgnoreMaskLabel >= 0.
int ref_estimate_depth_map(const OrcView* views, int nViews, float* depth, float* normal, float* conf, float dMin, float dMax, const OrcOpt* opt,
		int nGeometricIter, const unsigned char* mask, int maskMode) {
	if (nViews < 2) return -1;
	OPTDENSE::nRandomIters = opt->nRandomIters; OPTDENSE::fEstimationGeometricWeight = opt->fEstimationGeometricWeight;
	OPTDENSE::fRandomDepthRatio = opt->fRandomDepthRatio; OPTDENSE::fRandomAngle1Range = opt->fRandomAngle1Range; OPTDENSE::fRandomAngle2Range = opt->fRandomAngle2Range;
	OPTDENSE::fRandomSmoothDepth = opt->fRandomSmoothDepth; OPTDENSE::fRandomSmoothNormal = opt->fRandomSmoothNormal; OPTDENSE::fRandomSmoothBonus = opt->fRandomSmoothBonus;
	OPTDENSE::fNCCThresholdKeep = opt->fNCCThresholdKeep; OPTDENSE::fDescriptorMinMagnitudeThreshold = opt->fDescriptorMinMagnitudeThreshold;
	OPTDENSE::nEstimationIters = opt->nEstimationIters; OPTDENSE::nEstimationGeometricIters = opt->nEstimationGeometricIters;
	OPTDENSE::nSubResolutionLevels = opt->nSubResolutionLevels;
	OPTDENSE::nIgnoreMaskLabel = maskMode ? 0 : -1;
	const int w = views[0].w, h = views[0].h;
	g_ignoreMask = maskMode ? mask : nullptr; g_maskW = w; g_maskH = h;
	const cv::Size size(w, h);
	std::vector<Image> images((size_t)nViews);
	DepthMapsData dm;
	dm.scene.nMaxThreads = (unsigned)std::max(1, opt->nThreads);
	dm.arrDepthData.resize(1);
	DepthData& depthData = dm.arrDepthData[0];
	depthData.images.Resize((IIndex)nViews);
	for (int i = 0; i < nViews; ++i) {
		DepthData::ViewData& v = depthData.images[(IIndex)i]; const OrcView& s = views[i];
		Image& im = images[(size_t)i];
		im.ID = (uint32_t)i; setCamera(im.camera, s.K, s.R, s.C); im.size = cv::Size(s.w, s.h);
		v.scale = 1.f; v.pImageData = &im;
		v.camera = im.camera;
		v.image.create(cv::Size(s.w, s.h)); memcpy(v.image.data(), s.image, sizeof(float) * (size_t)s.w * s.h);
		if (i > 0 && s.depth) {
			const int dw = s.dw > 0 ? s.dw : s.w, dh = s.dh > 0 ? s.dh : s.h;
			v.depthMap.create(cv::Size(dw, dh)); memcpy(v.depthMap.data(), s.depth, sizeof(float) * (size_t)dw * dh);
			setCamera(v.cameraDepthMap, s.Kd, s.Rd, s.Cd);
		}
	}
	for (DepthData::ViewData& v : depthData.images) v.Init(depthData.images.First().camera);   // DepthMapsData::InitViews, SceneDensify.cpp:395-397
	depthData.dMin = dMin; depthData.dMax = dMax;
	depthData.depthMap.create(size); memcpy(depthData.depthMap.data(), depth, sizeof(float) * (size_t)w * h);
	depthData.normalMap.create(size); memcpy(depthData.normalMap.data(), normal, sizeof(float) * 3 * (size_t)w * h);
	const bool ok = dm.EstimateDepthMap(0, nGeometricIter);
	g_ignoreMask = nullptr; OPTDENSE::nIgnoreMaskLabel = -1;
	if (!ok) return 1;
	DepthData& out = dm.arrDepthData[0];
	if (out.depthMap.size() != size || out.normalMap.size() != size || out.confMap.size() != size) return 2;
	memcpy(depth, out.depthMap.data(), sizeof(float) * (size_t)w * h);
	memcpy(normal, out.normalMap.data(), sizeof(float) * 3 * (size_t)w * h);
	memcpy(conf, out.confMap.data(), sizeof(float) * (size_t)w * h);
	return 0;
}
// DepthMapsData::ScaleDepthData (SceneDensify.cpp:578-601) of one view: image, camera K and the optional depth map with its camera, scaled by 1 / f
int ref_scale_view(const OrcView* view, int f, float* outImage, double* outK, float* outDepth, double* outKd, int* outWH) {
	Image im; im.ID = 0; setCamera(im.camera, view->K, view->R, view->C); im.size = cv::Size(view->w, view->h);
	DepthData dd; dd.images.Resize(1);
	DepthData::ViewData& v = dd.images[0];
	v.scale = 1.f; v.pImageData = &im; v.camera = im.camera;
	v.image.create(im.size); memcpy(v.image.data(), view->image, sizeof(float) * (size_t)view->w * view->h);
	if (view->depth) { v.depthMap.create(im.size); memcpy(v.depthMap.data(), view->depth, sizeof(float) * (size_t)view->w * view->h); setCamera(v.cameraDepthMap, view->Kd, view->Rd, view->Cd); }
	const DepthData r(DepthMapsData::ScaleDepthData(dd, 1.f / (float)f));
	const DepthData::ViewData& o = r.images[0];
	outWH[0] = o.image.cols; outWH[1] = o.image.rows;
	memcpy(outImage, const_cast<Image32F&>(o.image).data(), sizeof(float) * (size_t)o.image.cols * o.image.rows);
	for (int i = 0; i < 9; ++i) outK[i] = o.camera.K.val[i];
	if (view->depth) { memcpy(outDepth, const_cast<DepthMap&>(o.depthMap).data(), sizeof(float) * (size_t)o.depthMap.cols * o.depthMap.rows); for (int i = 0; i < 9; ++i) outKd[i] = o.cameraDepthMap.K.val[i]; }
	return 0;
}
const char* ref_driver_math_kind() {
#ifdef REF_MATH_PM
	return "pm_math";
#else
	return "libm";
#endif
}
}
