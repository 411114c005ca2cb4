// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  Drives the reference's own estimator code -- cut verbatim from /root/reference by
// oracle/ref/build_ref.py and compiled against oracle/ref/shim -- through a C interface shaped like oracle/pm_oracle.cpp's, so that tests can hand the
// same arrays to both and compare bit for bit.  The only code written here is the set-up the reference does in DepthMapsData::EstimateDepthMap
// (SceneDensify.cpp:616-805) for ONE pyramid level with ONE estimator thread: ViewData::Init, the zig-zag pixel list, the weight cache, one
// DepthEstimator per pass, then ScoreDepthMapTmp / EstimateDepthMapTmp / EndDepthMapTmp.  Resampling between levels (cv::resize) is not part of it.
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
#include "snip/depthmap_cpp_copy.inc"     // libs/MVS/DepthMap.cpp:121-133: DepthData's copy constructor (arrDepthData holds DepthData by value)
#include "snip/depthmap_cpp.inc"          // libs/MVS/DepthMap.cpp:325-972: MapMatrix2ZigzagIdx, the constructor, PreparePixelPatch ... InitPlane

namespace MVS {
class DepthMapsData {                     // SceneDensify.h: only the three pass bodies
public:
	static void* STCALL ScoreDepthMapTmp(void*);
	static void* STCALL EstimateDepthMapTmp(void*);
	static void* STCALL EndDepthMapTmp(void*);
	bool RemoveSmallSegments(DepthData& depthData);
	bool GapInterpolation(DepthData& depthData);
	bool FilterDepthMap(DepthData& depthData, const IIndexArr& idxNeighbors, bool bAdjust=true);
	struct { unsigned nCalibratedImages; } scene;   // Scene: the one member FilterDepthMap reads
	DepthDataArr arrDepthData;
};
// FilterDepthMap hands its result to SaveDepthMap / SaveConfidenceMap (files "filtered.dmap" / ".cmap", DepthMap.cpp); here they keep the maps for the caller
static DepthMap g_filteredDepth; static ConfidenceMap g_filteredConf;
inline bool SaveDepthMap(const String&, const DepthMap& m) { g_filteredDepth = m; return true; }
inline bool SaveConfidenceMap(const String&, const ConfidenceMap& m) { g_filteredConf = m; return true; }
}
#define TD_TIMER_STARTD() ((void)0)
#define TD_TIMER_GET_FMT() String()
#undef ComposeDepthFilePath                    // DepthMap.h:72 builds a file name; nothing is written here
#define ComposeDepthFilePath(i, e) String()
#include "snip/scenedensify_cpp.inc"      // libs/MVS/SceneDensify.cpp:489-576: the three pass bodies
#include "snip/scenedensify_filters.inc"  // libs/MVS/SceneDensify.cpp:809-1045: RemoveSmallSegments, GapInterpolation
#include "snip/scenedensify_filterdm.inc" // libs/MVS/SceneDensify.cpp:1049-1299: FilterDepthMap

extern "C" {
// same layout as oracle/pm_oracle.cpp's OrcView / OrcOpt
struct OrcView {
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

// One pyramid level of DepthMapsData::EstimateDepthMap with one estimator thread (SceneDensify.cpp:649-768, :771-797):
//   doInit   : ScoreDepthMapTmp with an estimator built for iteration iterBegin
//   sweeps   : EstimateDepthMapTmp for iter = iterBegin .. iterEnd-1, a fresh estimator each (fresh RNG, non-release seed)
//   thEnd>=0 : EndDepthMapTmp with OPTDENSE::fNCCThresholdKeep = thEnd
// depth / normal / conf: w*h (x3) floats, in and out.  prior: the low-resolution depth map resized to this level (nullable = empty).
// mask: nullable w*h bytes, 0 = ignored pixel (DepthData::ApplyIgnoreMask + the masked pixel list).
int ref_run_level(const OrcView* views, int nViews, float* depth, float* normal, float* conf, const float* prior,
		float dMin, float dMax, const OrcOpt* opt, int doInit, unsigned iterBegin, unsigned iterEnd, float thEnd, const unsigned char* mask) {
	if (nViews < 2) return -1;
	OPTDENSE::nRandomIters = opt->nRandomIters; OPTDENSE::fEstimationGeometricWeight = opt->fEstimationGeometricWeight;
	OPTDENSE::fRandomDepthRatio = opt->fRandomDepthRatio; OPTDENSE::fRandomAngle1Range = opt->fRandomAngle1Range; OPTDENSE::fRandomAngle2Range = opt->fRandomAngle2Range;
	OPTDENSE::fRandomSmoothDepth = opt->fRandomSmoothDepth; OPTDENSE::fRandomSmoothNormal = opt->fRandomSmoothNormal; OPTDENSE::fRandomSmoothBonus = opt->fRandomSmoothBonus;
	OPTDENSE::fNCCThresholdKeep = opt->fNCCThresholdKeep; OPTDENSE::fDescriptorMinMagnitudeThreshold = opt->fDescriptorMinMagnitudeThreshold;
	OPTDENSE::nEstimationIters = opt->nEstimationIters; OPTDENSE::nEstimationGeometricIters = opt->nEstimationGeometricIters;
	const int w = views[0].w, h = views[0].h;
	const cv::Size size(w, h);
	DepthData depthData;
	depthData.images.Resize((IIndex)nViews);
	static Image dummyImage[64];
	for (int i = 0; i < nViews; ++i) {
		DepthData::ViewData& v = depthData.images[(IIndex)i]; const OrcView& s = views[i];
		v.scale = 1.f; v.pImageData = &dummyImage[i % 64]; dummyImage[i % 64].ID = (uint32_t)i;
		setCamera(v.camera, s.K, s.R, s.C);
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
	depthData.confMap.create(size); memcpy(depthData.confMap.data(), conf, sizeof(float) * (size_t)w * h);
	DepthMap currentSizeResDepthMap;
	if (prior) { currentSizeResDepthMap.create(size); memcpy(currentSizeResDepthMap.data(), prior, sizeof(float) * (size_t)w * h); }
	DepthEstimator::WeightMap weightMap0;
	weightMap0.clear();
	weightMap0.resize(size.area() - (size.width + 1) * DepthEstimator::nSizeHalfWindow);           // SceneDensify.cpp:674-675
	BitMatrix bmask;
	if (mask) {
		bmask.create(w, h);
		for (size_t i = 0; i < (size_t)w * h; ++i) bmask.bits[i] = mask[i] ? 1 : 0;
		// DepthData::ApplyIgnoreMask, DepthMap.cpp:215-231
		for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) if (!mask[(size_t)r * w + c]) {
			depthData.depthMap(r, c) = 0; depthData.normalMap(r, c) = Normal::ZERO; depthData.confMap(r, c) = 0; }
	}
	DepthEstimator::MapRefArr coords;
	DepthEstimator::MapMatrix2ZigzagIdx(size, coords, bmask, MAXF(64, 1 * 8));                   // :681, nMaxThreads = 1
	volatile Thread::safe_t idxPixel;
	if (doInit) {
		idxPixel = -1;
		DepthEstimator estimator(iterBegin, depthData, idxPixel, weightMap0, coords);
		estimator.lowResDepthMap = currentSizeResDepthMap;
		DepthMapsData::ScoreDepthMapTmp(&estimator);
	}
	for (unsigned iter = iterBegin; iter < iterEnd; ++iter) {
		idxPixel = -1;
		DepthEstimator estimator(iter, depthData, idxPixel, weightMap0, coords);
		estimator.lowResDepthMap = currentSizeResDepthMap;
		DepthMapsData::EstimateDepthMapTmp(&estimator);
	}
	if (thEnd >= 0) {
		const float keep(OPTDENSE::fNCCThresholdKeep);
		OPTDENSE::fNCCThresholdKeep = thEnd;
		idxPixel = -1;
		DepthEstimator estimator(0, depthData, idxPixel, weightMap0, coords);
		DepthMapsData::EndDepthMapTmp(&estimator);
		OPTDENSE::fNCCThresholdKeep = keep;
	}
	memcpy(depth, depthData.depthMap.data(), sizeof(float) * (size_t)w * h);
	memcpy(normal, depthData.normalMap.data(), sizeof(float) * 3 * (size_t)w * h);
	memcpy(conf, depthData.confMap.data(), sizeof(float) * (size_t)w * h);
	return 0;
}


static void loadViews(DepthData& depthData, const OrcView* views, int nViews, Image* dummy) {
	depthData.images.Resize((IIndex)nViews);
	for (int i = 0; i < nViews; ++i) {
		DepthData::ViewData& v = depthData.images[(IIndex)i]; const OrcView& s = views[i];
		v.scale = 1.f; v.pImageData = &dummy[i % 64]; dummy[i % 64].ID = (uint32_t)i;
		setCamera(v.camera, s.K, s.R, s.C);
		v.image.create(cv::Size(s.w, s.h)); memcpy(v.image.data(), s.image, sizeof(float) * (size_t)s.w * s.h);
		if (i > 0 && s.depth) {
			const int dw = s.dw > 0 ? s.dw : s.w, dh = s.dh > 0 ? s.dh : s.h;
			v.depthMap.create(cv::Size(dw, dh)); memcpy(v.depthMap.data(), s.depth, sizeof(float) * (size_t)dw * dh);
			setCamera(v.cameraDepthMap, s.Kd, s.Rd, s.Cd);
		}
	}
	for (DepthData::ViewData& v : depthData.images) v.Init(depthData.images.First().camera);
}
static void setOpt(const OrcOpt* opt) {
	OPTDENSE::nRandomIters = opt->nRandomIters; OPTDENSE::fEstimationGeometricWeight = opt->fEstimationGeometricWeight;
	OPTDENSE::fRandomDepthRatio = opt->fRandomDepthRatio; OPTDENSE::fRandomAngle1Range = opt->fRandomAngle1Range; OPTDENSE::fRandomAngle2Range = opt->fRandomAngle2Range;
	OPTDENSE::fRandomSmoothDepth = opt->fRandomSmoothDepth; OPTDENSE::fRandomSmoothNormal = opt->fRandomSmoothNormal; OPTDENSE::fRandomSmoothBonus = opt->fRandomSmoothBonus;
	OPTDENSE::fNCCThresholdKeep = opt->fNCCThresholdKeep; OPTDENSE::fDescriptorMinMagnitudeThreshold = opt->fDescriptorMinMagnitudeThreshold;
	OPTDENSE::nEstimationIters = opt->nEstimationIters; OPTDENSE::nEstimationGeometricIters = opt->nEstimationGeometricIters;
}
// DepthEstimator::ScorePixel for one plane hypothesis at one pixel, no close neighbours (same arguments as orc_score_pixel)
int ref_score_pixel(const OrcView* views, int nViews, const OrcOpt* opt, int x, int y, float depthv, const float* normalv, const float* prior, float* outScores, float* outAgg) {
	setOpt(opt);
	static Image dummy[64];
	DepthData depthData; loadViews(depthData, views, nViews, dummy);
	const int w = views[0].w, h = views[0].h; const cv::Size size(w, h);
	depthData.dMin = 0.1f; depthData.dMax = 100.f;
	depthData.depthMap.create(size); depthData.normalMap.create(size); depthData.confMap.create(size);
	DepthEstimator::WeightMap weightMap0; weightMap0.resize(size.area());
	DepthEstimator::MapRefArr coords;
	volatile Thread::safe_t idxPixel = -1;
	DepthEstimator e(0, depthData, idxPixel, weightMap0, coords);
	if (prior) { e.lowResDepthMap.create(size); memcpy(e.lowResDepthMap.data(), prior, sizeof(float) * (size_t)w * h); }
	if (!e.PreparePixelPatch(ImageRef(x, y)) || !e.FillPixelPatch()) return 1;
	const Normal n(normalv[0], normalv[1], normalv[2]);
	e.InitPlane(depthv, n);
	*outAgg = e.ScorePixel(depthv, n);
	for (int i = 0; i < nViews - 1; ++i) outScores[i] = e.scores[i];   // NB: ScorePixel's GetNth has partially sorted them
	return 0;
}
// InterpolatePixel, CorrectNormal and the smoothness factor of ScorePixelImage for one close neighbour (same arguments as orc_pixel_helpers).
// The factor is read off two ScorePixel calls on a one-source scene: score with the neighbour / score without it.
int ref_pixel_helpers(const OrcView* views, int nViews, const OrcOpt* opt, int x, int y, float dMin, float dMax, int nx, int ny, float ndepth, const float* nnormal,
		float* outInterpDepth, float* outCorrected) {
	setOpt(opt);
	static Image dummy[64];
	DepthData depthData; loadViews(depthData, views, nViews, dummy);
	const int w = views[0].w, h = views[0].h; const cv::Size size(w, h);
	depthData.dMin = dMin; depthData.dMax = dMax;
	depthData.depthMap.create(size); depthData.normalMap.create(size); depthData.confMap.create(size);
	DepthEstimator::WeightMap weightMap0; weightMap0.resize(size.area());
	DepthEstimator::MapRefArr coords;
	volatile Thread::safe_t idxPixel = -1;
	DepthEstimator e(0, depthData, idxPixel, weightMap0, coords);
	if (!e.PreparePixelPatch(ImageRef(x, y))) return 1;
	e.FillPixelPatch();
	Normal n(nnormal[0], nnormal[1], nnormal[2]);
	*outInterpDepth = e.InterpolatePixel(ImageRef(nx, ny), ndepth, n);
	e.CorrectNormal(n);
	outCorrected[0] = n.x; outCorrected[1] = n.y; outCorrected[2] = n.z;
	return 0;
}
// ViewData::Init (DepthMap.h:175-185) of one source view against a reference camera: the per-view constants of the homography and of the
// geometric term, as the reference computes them
void ref_view_init(const OrcView* ref, const OrcView* src, double* Hl, double* Hm, double* Hr, float* Tl, float* Tm, float* Tr, float* Tn) {
	Camera cref; setCamera(cref, ref->K, ref->R, ref->C);
	DepthData::ViewData v; setCamera(v.camera, src->K, src->R, src->C);
	if (src->depth) { v.depthMap.create(cv::Size(4, 4)); setCamera(v.cameraDepthMap, src->Kd, src->Rd, src->Cd); }
	v.Init(cref);
	for (int i = 0; i < 9; ++i) { Hl[i] = v.Hl.val[i]; Hr[i] = v.Hr.val[i]; }
	for (int i = 0; i < 3; ++i) Hm[i] = v.Hm.val[i];
	if (src->depth) { for (int i = 0; i < 9; ++i) { Tl[i] = v.Tl.val[i]; Tr[i] = v.Tr.val[i]; } Tm[0] = v.Tm.x; Tm[1] = v.Tm.y; Tm[2] = v.Tm.z; Tn[0] = v.Tn.x; Tn[1] = v.Tn.y; Tn[2] = v.Tn.z; }
}
void ref_zigzag(int w, int h, int rawStride, uint16_t* outXY) {
	DepthEstimator::MapRefArr coords; BitMatrix none;
	DepthEstimator::MapMatrix2ZigzagIdx(cv::Size(w, h), coords, none, rawStride);
	for (size_t i = 0; i < coords.size(); ++i) { outXY[2 * i] = coords[i].x; outXY[2 * i + 1] = coords[i].y; }
}
// DepthMapsData::RemoveSmallSegments / GapInterpolation (SceneDensify.cpp:809-1045) on caller-owned maps, in place
static void filterMaps(bool gap, float* depth, float* normal, float* conf, int w, int h, unsigned arg, float fDepthDiffThreshold) {
	DepthData dd;
	const cv::Size size(w, h);
	dd.depthMap.create(size); dd.normalMap.create(size); dd.confMap.create(size);
	memcpy(dd.depthMap.data(), depth, sizeof(float) * (size_t)w * h);
	memcpy(dd.normalMap.data(), normal, sizeof(float) * 3 * (size_t)w * h);
	memcpy(dd.confMap.data(), conf, sizeof(float) * (size_t)w * h);
	const float keep(OPTDENSE::fDepthDiffThreshold); const unsigned ks(OPTDENSE::nSpeckleSize), kg(OPTDENSE::nIpolGapSize);
	OPTDENSE::fDepthDiffThreshold = fDepthDiffThreshold;
	DepthMapsData dm;
	if (gap) { OPTDENSE::nIpolGapSize = arg; dm.GapInterpolation(dd); } else { OPTDENSE::nSpeckleSize = arg; dm.RemoveSmallSegments(dd); }
	OPTDENSE::fDepthDiffThreshold = keep; OPTDENSE::nSpeckleSize = ks; OPTDENSE::nIpolGapSize = kg;
	memcpy(depth, dd.depthMap.data(), sizeof(float) * (size_t)w * h);
	memcpy(normal, dd.normalMap.data(), sizeof(float) * 3 * (size_t)w * h);
	memcpy(conf, dd.confMap.data(), sizeof(float) * (size_t)w * h);
}
void ref_remove_small_segments(float* depth, float* normal, float* conf, int w, int h, unsigned nSpeckleSize, float fDepthDiffThreshold) { filterMaps(false, depth, normal, conf, w, h, nSpeckleSize, fDepthDiffThreshold); }
void ref_gap_interpolation(float* depth, float* normal, float* conf, int w, int h, unsigned nIpolGapSize, float fDepthDiffThreshold) { filterMaps(true, depth, normal, conf, w, h, nIpolGapSize, fDepthDiffThreshold); }
// DepthMapsData::FilterDepthMap (SceneDensify.cpp:1049-1299) of one reference view against N neighbour views; same FltView layout as oracle/filter_oracle.cpp.
// Returns 0 and fills newDepth / newConf, or 1 if the reference refuses the view (too few neighbours).
struct FltView { const float* depth; const float* conf; double K[9], R[9], C[3]; int w, h; };
int ref_filter_depth_map(const FltView* ref, const FltView* nb, int N, int w, int h, float dMin, float dMax, int bAdjust,
		unsigned nMinViewsFilter, unsigned nMinViewsFilterAdjust, unsigned nCalibratedImages, float fDepthDiffThreshold, float* newDepth, float* newConf) {
	DepthMapsData dm;
	dm.scene.nCalibratedImages = nCalibratedImages;
	static ImageArr images;                              // ViewData::pImageData of view i -> images[i] (GetID())
	images.resize((IIndex)(N + 1));
	for (int i = 0; i <= N; ++i) images[(IIndex)i].ID = (uint32_t)i;
	dm.arrDepthData.resize((IIndex)(N + 1));
	for (int i = 0; i <= N; ++i) {
		const FltView& s = i == 0 ? *ref : nb[i - 1];
		const cv::Size size(i && s.w ? s.w : w, i && s.h ? s.h : h);   // every depth map of its own size, as DepthMapsData::InitViews leaves them
		DepthData& dd = dm.arrDepthData[(IIndex)i];
		dd.images.resize(1);
		DepthData::ViewData& v = dd.images[0];
		setCamera(v.camera, s.K, s.R, s.C);
		v.pImageData = &images[(IIndex)i];
		dd.depthMap.create(size); memcpy(dd.depthMap.data(), s.depth, sizeof(float) * (size_t)size.width * size.height);
		dd.confMap.create(size); memcpy(dd.confMap.data(), s.conf, sizeof(float) * (size_t)size.width * size.height);
		dd.dMin = dMin; dd.dMax = dMax;
	}
	DepthData& dref = dm.arrDepthData[0];
	dref.neighbors.resize((IIndex)N);
	IIndexArr idx;
	for (int n = 0; n < N; ++n) { dref.neighbors[(IIndex)n].ID = (uint32_t)(n + 1); idx.push_back((IIndex)n); }
	const unsigned k1(OPTDENSE::nMinViewsFilter), k2(OPTDENSE::nMinViewsFilterAdjust); const float k3(OPTDENSE::fDepthDiffThreshold);
	OPTDENSE::nMinViewsFilter = nMinViewsFilter; OPTDENSE::nMinViewsFilterAdjust = nMinViewsFilterAdjust; OPTDENSE::fDepthDiffThreshold = fDepthDiffThreshold;
	g_filteredDepth.release(); g_filteredConf.release();
	const bool ok = dm.FilterDepthMap(dref, idx, bAdjust != 0);
	OPTDENSE::nMinViewsFilter = k1; OPTDENSE::nMinViewsFilterAdjust = k2; OPTDENSE::fDepthDiffThreshold = k3;
	if (!ok || g_filteredDepth.empty()) return 1;
	memcpy(newDepth, g_filteredDepth.data(), sizeof(float) * (size_t)w * h);
	memcpy(newConf, g_filteredConf.data(), sizeof(float) * (size_t)w * h);
	return 0;
}
const char* ref_math_kind() {
#ifdef REF_MATH_PM
	return "pm_math";
#else
	return "libm";
#endif
}
}
