// pm_oracle.cpp -- CPU restatement of OpenMVS's CPU PatchMatch depth-map estimator.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under openmvs_amd/ (the product) may
// include, link, import or execute this file; only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
// baseline ("kind": "port").
//
// PARITY PINNED BY THE REFERENCE'S OWN CODE (round 3): the reference ships no golden depth maps or numeric asserts for this path
// and its whole tree cannot be built here (OpenCV / Eigen / Boost / CGAL absent), but the estimator itself can: oracle/ref/build_ref.py cuts
// DepthEstimator (DepthMap.h:276-468, DepthMap.cpp:325-971), the pass bodies (SceneDensify.cpp:489-576) and the helpers they call out of
// /root/reference verbatim and compiles them against a minimal OpenCV / Eigen stand-in (oracle/_ref).  tests/test_ref_pinning.py runs the same
// arrays through that code and through this file (orc_run_level, rngMode 2) and compares bit for bit: init pass, sweeps in both directions,
// geometric rounds, prior, masks, 1-4 sources, option sets, noise starts, and InterpolatePixel / CorrectNormal / ScorePixel on random planes.
// The first such comparison found one real deviation -- norm(Point2f) of the geometric term binds to SEACAVE::norm (float), not cv::norm
// (double) -- which is fixed here and in the kernels.  (paths below are relative to /root/reference)
//
// Deliberate, documented differences from the reference (DESIGN.md "Oracle"):
//   * RNG: default mode is a counter-based Philox4x32-10 keyed by (seed, pass, pixel,
//     attempt) so a parallel schedule can reproduce it; mode 1 keeps the reference's
//     per-estimator std::mt19937 stream in traversal order (Random.h:102-137).
//   * exp/acos/atan2/sin/cos come from csrc/pm_math.h (same polynomial kernels as the
//     GPU) instead of libm (the reference's side of the pinning tests is routed to the same header; a second build of oracle/_ref with
//     libm bounds what that changes: tests/test_ref_pinning.py::test_libm_build_bounds_the_pm_math_substitution).
//     CorrectNormal's rotation is evaluated in float -- as the reference does (Rotation.inl:701-728 with TYPE = float; checked by the pinning tests).
//   * third-party arithmetic (cv::resize, cv::Matx::inv) is restated from OpenCV's
//     published algorithms (integer-factor INTER_AREA incl. its border rule for non-divisible sizes).
#include "../openmvs_amd/csrc/pm_math.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <random>
#include <thread>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------
// small containers
struct ImgF {
	int w = 0, h = 0;
	std::vector<float> d;
	void create(int W, int H) { w = W; h = H; d.assign((size_t)W * H, 0.f); }
	bool empty() const { return d.empty(); }
	float& operator()(int y, int x) { return d[(size_t)y * w + x]; }
	float operator()(int y, int x) const { return d[(size_t)y * w + x]; }
};
struct ImgN { // 3 floats / pixel
	int w = 0, h = 0;
	std::vector<float> d;
	void create(int W, int H) { w = W; h = H; d.assign((size_t)W * H * 3, 0.f); }
	bool empty() const { return d.empty(); }
	float* at(int y, int x) { return &d[((size_t)y * w + x) * 3]; }
	const float* at(int y, int x) const { return &d[((size_t)y * w + x) * 3]; }
};
struct Cam { double K[9], R[9], C[3]; };

// cv::Matx product convention: c(i,j) = sum_k a(i,k)*b(k,j), accumulated left to right from 0
static void mul33(const double* a, const double* b, double* c) {
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
		double s = 0; for (int k = 0; k < 3; ++k) s += a[i*3+k] * b[k*3+j]; c[i*3+j] = s; }
}
static void mul31(const double* a, const double* v, double* c) {
	for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += a[i*3+k] * v[k]; c[i] = s; }
}
static void transpose33(const double* a, double* t) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[j*3+i] = a[i*3+j]; }
// cv::Matx<double,3,3>::inv() closed form (OpenCV matx.hpp Matx_FastInvOp<_Tp,3,3>)
static void inv33(const double* a, double* b) {
	double d = a[0]*(a[4]*a[8] - a[7]*a[5]) - a[1]*(a[3]*a[8] - a[6]*a[5]) + a[2]*(a[3]*a[7] - a[6]*a[4]);
	d = 1 / d;
	b[0] = (a[4]*a[8] - a[5]*a[7]) * d; b[1] = (a[2]*a[7] - a[1]*a[8]) * d; b[2] = (a[1]*a[5] - a[2]*a[4]) * d;
	b[3] = (a[5]*a[6] - a[3]*a[8]) * d; b[4] = (a[0]*a[8] - a[2]*a[6]) * d; b[5] = (a[2]*a[3] - a[0]*a[5]) * d;
	b[6] = (a[3]*a[7] - a[4]*a[6]) * d; b[7] = (a[1]*a[6] - a[0]*a[7]) * d; b[8] = (a[0]*a[4] - a[1]*a[3]) * d;
}
// Camera::InvK, libs/MVS/Camera.h:176-185
static void invK(const double* K, double* o) {
	for (int i = 0; i < 9; ++i) o[i] = (i % 4 == 0) ? 1.0 : 0.0;
	o[0] = 1.0 / K[0]; o[4] = 1.0 / K[4]; o[2] = -K[2] * o[0]; o[5] = -K[5] * o[4];
}
// Camera::ScaleK(K,size,newSize), libs/MVS/Camera.h:160-170
static void scaleK(const double* K, int w, int h, int nw, int nh, double* o) {
	const double sx = (double)nw / (double)w, sy = (double)nh / (double)h;
	o[0] = K[0]*sx; o[1] = K[1]*sx; o[2] = (K[2]+0.5)*sx-0.5;
	o[3] = 0;       o[4] = K[4]*sy; o[5] = (K[5]+0.5)*sy-0.5;
	o[6] = 0; o[7] = 0; o[8] = 1;
}

// ---------------------------------------------------------------------------
// third-party resampling (OpenCV cv::resize), restated; see header note
// cv::resize(src, dst, Size(), fx, fy, ...) output size: saturate_cast<int>(ssize*fx) == cvRound (ties to even)
static inline int cvRoundHalfEven(double v) { return (int)nearbyint(v); }
static inline int scaledSize(int n, int f) { return cvRoundHalfEven((double)n / (double)f); }
// INTER_AREA with integer factor f = 1/fx (scale_x == scale_y == f exactly -> OpenCV's "area fast" path):
// a destination pixel whose f x f block lies inside the source is, for f==2, ((a+b)+(c+d))*0.25 (ResizeAreaFastVec),
// otherwise the running row-major sum * (1/f^2) (ResizeAreaFast_Invoker); blocks cut by the right/bottom border
// (sizes not divisible by f) average the available pixels, (float)sum/count, and every pixel of a cut bottom row
// takes that path (w = 0 in ResizeAreaFast_Invoker).
static void resizeArea(const ImgF& s, int f, ImgF& o) {
	o.create(scaledSize(s.w, f), scaledSize(s.h, f));
	const float scale = 1.f / (float)(f * f);
	const int fullCols = s.w / f;
	for (int y = 0; y < o.h; ++y) {
		const int sy0 = y * f;
		const bool rowFull = sy0 + f <= s.h;
		for (int x = 0; x < o.w; ++x) {
			const int sx0 = x * f;
			if (sy0 >= s.h || sx0 >= s.w) { o(y,x) = 0; continue; }
			if (rowFull && x < fullCols) {
				if (f == 2) {
					o(y,x) = ((s(sy0,sx0) + s(sy0,sx0+1)) + (s(sy0+1,sx0) + s(sy0+1,sx0+1))) * 0.25f;
				} else {
					float sum = 0;
					for (int j = 0; j < f; ++j) for (int i = 0; i < f; ++i) sum += s(sy0+j, sx0+i);
					o(y,x) = sum * scale;
				}
			} else {
				float sum = 0; int count = 0;
				for (int j = 0; j < f && sy0 + j < s.h; ++j) for (int i = 0; i < f && sx0 + i < s.w; ++i) { sum += s(sy0+j, sx0+i); ++count; }
				o(y,x) = sum / (float)count;
			}
		}
	}
}
// INTER_NEAREST down by fx = 1/f (ifx = 1/fx = f exactly): sx = min(dx*f, ssize-1)
static void resizeNearestDown(const ImgF& s, int f, ImgF& o) {
	o.create(scaledSize(s.w, f), scaledSize(s.h, f));
	for (int y = 0; y < o.h; ++y) { const int sy = std::min(y * f, s.h - 1);
		for (int x = 0; x < o.w; ++x) o(y,x) = s(sy, std::min(x * f, s.w - 1)); }
}
static void resizeNearestDownN(const ImgN& s, int f, ImgN& o) {
	o.create(scaledSize(s.w, f), scaledSize(s.h, f));
	for (int y = 0; y < o.h; ++y) { const int sy = std::min(y * f, s.h - 1);
		for (int x = 0; x < o.w; ++x) { const float* p = s.at(sy, std::min(x * f, s.w - 1)); float* q = o.at(y,x); q[0]=p[0]; q[1]=p[1]; q[2]=p[2]; } }
}
// INTER_NEAREST: sx = min(floor(dx*ssize/dsize), ssize-1)
static void resizeNearest(const ImgF& s, int nw, int nh, ImgF& o) {
	o.create(nw, nh);
	const double ifx = (double)s.w / nw, ify = (double)s.h / nh;
	for (int y = 0; y < nh; ++y) { const int sy = std::min((int)floor(y * ify), s.h - 1);
		for (int x = 0; x < nw; ++x) { const int sx = std::min((int)floor(x * ifx), s.w - 1); o(y,x) = s(sy,sx); } }
}
static void resizeNearestN(const ImgN& s, int nw, int nh, ImgN& o) {
	o.create(nw, nh);
	const double ifx = (double)s.w / nw, ify = (double)s.h / nh;
	for (int y = 0; y < nh; ++y) { const int sy = std::min((int)floor(y * ify), s.h - 1);
		for (int x = 0; x < nw; ++x) { const int sx = std::min((int)floor(x * ifx), s.w - 1);
			const float* p = s.at(sy,sx); float* q = o.at(y,x); q[0]=p[0]; q[1]=p[1]; q[2]=p[2]; } }
}
// INTER_LINEAR (float): half-pixel centres, edge clamp, horizontal pass then vertical
static void linearCoef(int dn, int sn, std::vector<int>& idx, std::vector<float>& a) {
	idx.resize(dn); a.resize(dn);
	const double scale = (double)sn / dn;
	for (int d = 0; d < dn; ++d) {
		float f = (float)((d + 0.5) * scale - 0.5);
		int s = (int)floorf(f);
		f -= s;
		if (s < 0) { f = 0; s = 0; }
		if (s >= sn - 1) { f = 0; s = sn - 1; }
		idx[d] = s; a[d] = f;
	}
}
static void resizeLinear(const ImgF& s, int nw, int nh, ImgF& o) {
	o.create(nw, nh);
	std::vector<int> xi, yi; std::vector<float> xa, ya;
	linearCoef(nw, s.w, xi, xa); linearCoef(nh, s.h, yi, ya);
	for (int y = 0; y < nh; ++y) {
		const int y0 = yi[y], y1 = std::min(y0 + 1, s.h - 1);
		const float b1 = ya[y], b0 = 1.f - b1;
		for (int x = 0; x < nw; ++x) {
			const int x0 = xi[x], x1 = std::min(x0 + 1, s.w - 1);
			const float a1 = xa[x], a0 = 1.f - a1;
			const float t0 = s(y0,x0)*a0 + s(y0,x1)*a1;
			const float t1 = s(y1,x0)*a0 + s(y1,x1)*a1;
			o(y,x) = t0*b0 + t1*b1;
		}
	}
}

// ---------------------------------------------------------------------------
// DepthData::ViewData, libs/MVS/DepthMap.h:158-205
struct ViewData {
	Cam camera;
	ImgF image;
	ImgF depthMap;         // known depth-map of this source view (geometric pass only)
	Cam cameraDepthMap;
	double Hl[9], Hm[3], Hr[9];
	float Tl[9], Tm[3], Tr[9], Tn[3];
	// ViewData::Init, DepthMap.h:175-185
	void Init(const Cam& ref) {
		double KR[9], RrT[9], dC[3];
		mul33(camera.K, camera.R, KR);
		transpose33(ref.R, RrT);
		mul33(KR, RrT, Hl);
		for (int i = 0; i < 3; ++i) dC[i] = ref.C[i] - camera.C[i];
		mul31(KR, dC, Hm);
		inv33(ref.K, Hr);
		if (!depthMap.empty()) {
			double KdRd[9], t[9], v[3], RdT[9], KR0[9], iKd[9], t2[9];
			mul33(cameraDepthMap.K, cameraDepthMap.R, KdRd);
			mul33(KdRd, RrT, t);
			for (int i = 0; i < 9; ++i) Tl[i] = (float)t[i];
			for (int i = 0; i < 3; ++i) dC[i] = ref.C[i] - cameraDepthMap.C[i];
			mul31(KdRd, dC, v);
			for (int i = 0; i < 3; ++i) Tm[i] = (float)v[i];
			mul33(ref.K, ref.R, KR0);
			transpose33(cameraDepthMap.R, RdT);
			mul33(KR0, RdT, t);
			invK(cameraDepthMap.K, iKd);
			mul33(t, iKd, t2);
			for (int i = 0; i < 9; ++i) Tr[i] = (float)t2[i];
			for (int i = 0; i < 3; ++i) dC[i] = cameraDepthMap.C[i] - ref.C[i];
			mul31(KR0, dC, v);
			for (int i = 0; i < 3; ++i) Tn[i] = (float)v[i];
		}
	}
};
struct DepthData {
	std::vector<ViewData> images; // [0] = reference
	ImgF depthMap; ImgN normalMap; ImgF confMap;
	float dMin = 0, dMax = 0;
};

// OPTDENSE subset, defaults = libs/MVS/DepthMap.cpp:69-114
struct Opt {
	unsigned nSubResolutionLevels = 2;
	unsigned nEstimationIters = 3;
	unsigned nEstimationGeometricIters = 2;
	float fEstimationGeometricWeight = 0.1f;
	unsigned nRandomIters = 6;
	float fRandomDepthRatio = 0.003f;
	float fRandomAngle1Range = 16.f;
	float fRandomAngle2Range = 10.f;
	float fRandomSmoothDepth = 0.02f;
	float fRandomSmoothNormal = 13.f;
	float fRandomSmoothBonus = 0.93f;
	float fNCCThresholdKeep = 0.9f;
	float fDescriptorMinMagnitudeThreshold = 0.02f;
	// oracle controls
	uint32_t seed = 0;       // RNG seed
	uint32_t viewID = 0;     // mixed into the Philox key
	int rngMode = 0;         // 0 = Philox per (pixel,attempt); 1 = mt19937 in traversal order (theta drawn before phi); 2 = the same with the two-draw
	                         // expressions evaluated right to left, as GCC compiles the reference (DepthMap.h:441, DepthMap.cpp:836) -- what oracle/_ref is compared with
	int nThreads = 1;        // 1 = sequential parity oracle; >1 = reference threading model (timing baseline)
	// Tiled sweeps (the engine's opt-in PMHipParams::tileW / tileH; 0 = the reference's one sweep over the whole map): the pixels that take part in the estimation
	// (x, y >= HW) are cut into tileW x tileH tiles; a sweep runs inside every tile in the reference's order, and a neighbour in ANOTHER tile is read as the previous sweep
	// left it (depth, normal and confidence as they were when this sweep started).  The tiles of a sweep are then independent of each other -- which is all the engine
	// gains from it -- and the result is a function of the inputs alone, so that this restatement and the device agree bit for bit.  NOT the reference's result.
	int tileW = 0, tileH = 0;
};

// WeightedPatchFix<25>, DepthMap.h:145-155
struct Weight { float weight[25], tempWeight[25]; float sumWeights; float normSq0 = 0; };

enum { STREAM_INIT = 0, STREAM_RAND = 1, STREAM_REFINE = 2 };

// SEACAVE::Random restated (Random.h:102-137) with a second, counter-based mode
struct Rng {
	int mode; std::mt19937 mt; uint32_t k0, k1; PmPhilox4 cur;
	Rng(int m, uint32_t key0, uint32_t key1) : mode(m), mt(std::mt19937::default_seed), k0(key0), k1(key1) {}
	void attempt(int x, int y, int stream, int it) { if (mode == 0) cur = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(stream * 256 + it), 0u, k0, k1); }
	float unit(int slot) { return mode == 0 ? pm_u32_to_unit(cur.v[slot]) : pm_u32_to_unit((uint32_t)mt()); }
};
// one Philox key per (seed, view, pass); pass = level*64 + kind*32 + iter
static inline void passKey(uint32_t seed, uint32_t viewID, uint32_t pass, uint32_t& k0, uint32_t& k1) { k0 = seed; k1 = viewID * 0x9E3779B1u + pass; }

#define FD2R(d) ((d) * (PM_PI_F / 180.f))
#define SQ(x) ((x) * (x))

// ---------------------------------------------------------------------------
// DepthEstimator, libs/MVS/DepthMap.h:276-468 + DepthMap.cpp:361-971
struct NeighborEstimate { float depth; float normal[3]; float X[3]; };
struct DepthEstimator {
	enum { HW = 4, STEP = 2, NT = 25 };
	enum { LT2RB = 0, RB2LT = 1 };
	Rng rnd;
	std::atomic<long>& idxPixel;
	int nbX[2], nbY[2], nNb = 0;           // "neighbors"
	NeighborEstimate close[4]; int nClose = 0; // "neighborsClose"
	double X0[3]; int x0x, x0y; float normSq0;
	std::vector<float> scores;
	float planeN[3], planeD;
	ImgF& depthMap0; ImgN& normalMap0; ImgF& confMap0;
	std::vector<Weight>& weightMap0;
	const ImgF* lowResDepthMap = nullptr;  // prior (nullptr == empty)
	// tiled sweeps (Opt::tileW): the maps as the sweep found them, read for neighbours that lie in another tile
	const ImgF* oldDepth = nullptr; const ImgN* oldNormal = nullptr; const ImgF* oldConf = nullptr;
	bool otherTile(int nx, int ny) const { return opt.tileW > 0 && oldDepth && ((nx - HW) / opt.tileW != (x0x - HW) / opt.tileW || (ny - HW) / opt.tileH != (x0y - HW) / opt.tileH); }
	float nbDepth(int nx, int ny) const { return otherTile(nx, ny) ? (*oldDepth)(ny, nx) : depthMap0(ny, nx); }
	const float* nbNormal(int nx, int ny) const { return otherTile(nx, ny) ? oldNormal->at(ny, nx) : normalMap0.at(ny, nx); }
	float nbConf(int nx, int ny) const { return otherTile(nx, ny) ? (*oldConf)(ny, nx) : confMap0(ny, nx); }
	const unsigned nIteration;
	const std::vector<ViewData>& views;    // images[0] = reference, images[1..] = sources
	const ViewData& image0;
	const std::vector<std::pair<uint16_t,uint16_t>>& coords;
	const int W, H;
	const float dMin, dMax, dMinSqr, dMaxSqr;
	const int dir;
	const unsigned idxScore;
	const Opt& opt;
	const float smoothBonusDepth, smoothBonusNormal, smoothSigmaDepth, smoothSigmaNormal;
	const float thMagnitudeSq, angle1Range, angle2Range, thConfSmall, thConfBig, thConfRand, thRobust;
	const float geoWeight;

	// ctor: DepthMap.cpp:361-412
	DepthEstimator(unsigned nIter, DepthData& dd, std::atomic<long>& idx, std::vector<Weight>& wm,
			const std::vector<std::pair<uint16_t,uint16_t>>& c, const Opt& o, uint32_t k0, uint32_t k1)
		: rnd(o.rngMode, k0, k1), idxPixel(idx), scores(dd.images.size() - 1),
		  depthMap0(dd.depthMap), normalMap0(dd.normalMap), confMap0(dd.confMap), weightMap0(wm),
		  nIteration(nIter), views(dd.images), image0(dd.images[0]), coords(c),
		  W(dd.images[0].image.w), H(dd.images[0].image.h),
		  dMin(dd.dMin), dMax(dd.dMax), dMinSqr(pm_sqrtf(dd.dMin)), dMaxSqr(pm_sqrtf(dd.dMax)),
		  dir(nIter % 2 ? RB2LT : LT2RB),
		  idxScore(dd.images.size() <= 2 ? 0u : 1u), opt(o),
		  smoothBonusDepth(1.f - o.fRandomSmoothBonus), smoothBonusNormal((1.f - o.fRandomSmoothBonus) * 0.96f),
		  smoothSigmaDepth(-1.f / (2.f * SQ(o.fRandomSmoothDepth))),
		  smoothSigmaNormal(-1.f / (2.f * SQ(FD2R(o.fRandomSmoothNormal)))),
		  thMagnitudeSq(o.fDescriptorMinMagnitudeThreshold > 0 ? SQ(o.fDescriptorMinMagnitudeThreshold) : -1.f),
		  angle1Range(FD2R(o.fRandomAngle1Range)), angle2Range(FD2R(o.fRandomAngle2Range)),
		  thConfSmall(o.fNCCThresholdKeep * 0.66f), thConfBig(o.fNCCThresholdKeep * 0.9f),
		  thConfRand(o.fNCCThresholdKeep * 1.1f), thRobust(o.fNCCThresholdKeep * 4.f / 3.f),
		  geoWeight(o.fEstimationGeometricWeight) {}

	// DepthMap.cpp:415-420
	bool PreparePixelPatch(int x, int y) {
		x0x = x; x0y = y;
		return x - HW >= 0 && y - HW >= 0 && x + HW < W && y + HW < H;
	}
	// GetWeight, DepthMap.h:403-412
	float GetWeight(int j, int i, float center) const {
		const float sigmaColor = -1.f / (2.f * SQ(0.1f));
		const float wColor = SQ(image0.image(x0y + i, x0x + j) - center) * sigmaColor;
		const float sigmaSpatial = -1.f / (2.f * (float)SQ((int)HW - 1));
		const float wSpatial = (float)(SQ(j) + SQ(i)) * sigmaSpatial;
		return pm_expf(wColor + wSpatial);
	}
	// DepthMap.cpp:422-462 (DENSE_NCC_WEIGHTED branch)
	bool FillPixelPatch() {
		Weight& w = weightMap0[(size_t)x0y * W + x0x];
		if (w.normSq0 == 0) {
			w.sumWeights = 0;
			int n = 0;
			const float colCenter = image0.image(x0y, x0x);
			for (int i = -HW; i <= HW; i += STEP) for (int j = -HW; j <= HW; j += STEP) {
				w.tempWeight[n] = image0.image(x0y + i, x0x + j);
				w.weight[n] = GetWeight(j, i, colCenter);
				w.normSq0 += w.tempWeight[n] * w.weight[n];
				w.sumWeights += w.weight[n];
				++n;
			}
			const float tm = w.normSq0 / w.sumWeights;
			w.normSq0 = 0;
			n = 0;
			do {
				const float t = w.tempWeight[n] - tm;
				w.tempWeight[n] = w.weight[n] * t;
				w.normSq0 += w.tempWeight[n] * t;
			} while (++n < NT);
		}
		normSq0 = w.normSq0;
		if (normSq0 < thMagnitudeSq && (lowResDepthMap == nullptr || (*lowResDepthMap)(x0y, x0x) <= 0))
			return false;
		// Camera::TransformPointI2C, Camera.h:331-336
		const double* K = image0.camera.K;
		X0[0] = ((double)x0x - K[2]) / K[0]; X0[1] = ((double)x0y - K[5]) / K[4]; X0[2] = 1.0;
		return true;
	}
	// ComputeHomographyMatrix, DepthMap.h:414-423
	void ComputeHomographyMatrix(const ViewData& img, float depth, const float* normal, float* H) const {
		const double n[3] = {(double)normal[0], (double)normal[1], (double)normal[2]};
		double ndx = 0; for (int i = 0; i < 3; ++i) ndx += n[i] * X0[i];
		const double den = ndx * (double)depth;
		const double inv = (den == 0.0) ? 1e+14 : 1.0 / den; // INVERT, Types.h:1234 (INV_ZERO 1e14)
		double M[9];
		for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i*3+j] = img.Hl[i*3+j] + img.Hm[i] * (n[j] * inv);
		double Hd[9]; mul33(M, img.Hr, Hd);
		for (int i = 0; i < 9; ++i) H[i] = (float)Hd[i];
	}
	// one factor of the smoothness product, DepthMap.cpp:524-533 (the current plane must have been set with InitPlane)
	float SmoothFactor(const NeighborEstimate& nb, float depth, const float* normal) const {
		// Planef::Distance (Eigen 3-vector dot: e0 + (e1 + e2)), Plane.inl:187-190
		const float dist = (planeN[0]*nb.X[0] + (planeN[1]*nb.X[1] + planeN[2]*nb.X[2])) + planeD;
		const float factorDepth = pm_expf(SQ(dist / depth) * smoothSigmaDepth);
		// ComputeAngle, Util.inl:544-546
		const float ca = pm_clampf((normal[0]*nb.normal[0] + normal[1]*nb.normal[1] + normal[2]*nb.normal[2]) /
			pm_sqrtf((normal[0]*normal[0] + normal[1]*normal[1] + normal[2]*normal[2]) *
			         (nb.normal[0]*nb.normal[0] + nb.normal[1]*nb.normal[1] + nb.normal[2]*nb.normal[2])), -1.f, 1.f);
		const float factorNormal = pm_expf(SQ(pm_acosf(ca)) * smoothSigmaNormal);
		return (1.f - smoothBonusDepth * factorDepth) * (1.f - smoothBonusNormal * factorNormal);
	}
	// InitPlane, DepthMap.cpp:963-971
	void InitPlane(float depth, const float* normal) {
		planeN[0] = normal[0]; planeN[1] = normal[1]; planeN[2] = normal[2];
		const float vx = (float)X0[0], vy = (float)X0[1], vz = (float)X0[2];
		planeD = -depth * (normal[0]*vx + normal[1]*vy + normal[2]*vz);
	}
	// TImage::sample, libs/Common/Types.inl:2273-2281
	static float sample(const ImgF& im, float px, float py) {
		const int lx = (int)px, ly = (int)py;
		const float x = px - lx, x1 = 1.f - x;
		const float y = py - ly, y1 = 1.f - y;
		return (im(ly,lx)*x1 + im(ly,lx+1)*x)*y1 + (im(ly+1,lx)*x1 + im(ly+1,lx+1)*x)*y;
	}
	// isInsideWithBorder<float,1>, libs/Common/Types.h:1649-1651
	static bool insideBorder1(const ImgF& im, float px, float py) {
		return px >= 1.f && py >= 1.f && px <= (float)(im.w - 2) && py <= (float)(im.h - 2);
	}
	// TImage::sample with validity functor, Types.inl:2299-2314; functor = IsDepthSimilar(z, d, 0.03)
	static bool sampleDepth(const ImgF& im, float px, float py, float z, float& v) {
		const int lx = (int)px, ly = (int)py;
		const float x = px - lx, x1 = 1.f - x;
		const float y = py - ly, y1 = 1.f - y;
		const float x0y0 = im(ly,lx), x1y0 = im(ly,lx+1), x0y1 = im(ly+1,lx), x1y1 = im(ly+1,lx+1);
		// IsDepthSimilar(d0=z, d1=d, th): |d0-d1|/d0 < th, Util.inl:798-809
		const bool b00 = pm_fabsf(z - x0y0) / z < 0.03f, b10 = pm_fabsf(z - x1y0) / z < 0.03f;
		const bool b01 = pm_fabsf(z - x0y1) / z < 0.03f, b11 = pm_fabsf(z - x1y1) / z < 0.03f;
		if (!b00 && !b10 && !b01 && !b11) return false;
		v = y1*(x1*(b00 ? x0y0 : (b10 ? x1y0 : (b01 ? x0y1 : x1y1))) + x*(b10 ? x1y0 : (b00 ? x0y0 : (b11 ? x1y1 : x0y1)))) +
		    y *(x1*(b01 ? x0y1 : (b11 ? x1y1 : (b00 ? x0y0 : x1y0))) + x*(b11 ? x1y1 : (b01 ? x0y1 : (b10 ? x1y0 : x0y0))));
		return true;
	}
	// ScorePixelImage, DepthMap.cpp:465-564 (WEIGHTED / SMOOTHNESS_PLANE branches)
	float ScorePixelImage(const ViewData& image1, float depth, const float* normal) {
		float Hm[9]; ComputeHomographyMatrix(image1, depth, normal, Hm);
		float X[3], baseX[3];
		const float px = (float)(x0x - HW), py = (float)(x0y - HW);
		for (int i = 0; i < 3; ++i) X[i] = Hm[i*3+0]*px + Hm[i*3+1]*py + Hm[i*3+2]; // ProjectVertex_3x3_2_3, Util.inl:382-386
		baseX[0] = X[0]; baseX[1] = X[1]; baseX[2] = X[2];
		for (int i = 0; i < 9; ++i) Hm[i] *= (float)STEP;
		int n = 0;
		float sum = 0, sumSq = 0, num = 0;
		const Weight& w = weightMap0[(size_t)x0y * W + x0x];
		for (int i = -HW; i <= HW; i += STEP) {
			for (int j = -HW; j <= HW; j += STEP) {
				const float ptx = X[0] / X[2], pty = X[1] / X[2]; // TPoint2(Point3), Types.h:1291
				if (!insideBorder1(image1.image, ptx, pty))
					return thRobust;
				const float v = sample(image1.image, ptx, pty);
				const float vw = v * w.weight[n];
				sum += vw;
				sumSq += v * vw;
				num += v * w.tempWeight[n];
				++n;
				X[0] += Hm[0]; X[1] += Hm[3]; X[2] += Hm[6];
			}
			baseX[0] += Hm[1]; baseX[1] += Hm[4]; baseX[2] += Hm[7];
			X[0] = baseX[0]; X[1] = baseX[1]; X[2] = baseX[2];
		}
		const float normSq1 = sumSq - SQ(sum) / w.sumWeights;
		const float nrmSq = normSq0 * normSq1;
		if (nrmSq <= 1e-16f)
			return thRobust;
		const float ncc = pm_clampf(num / pm_sqrtf(nrmSq), -1.f, 1.f);
		float score = 1.f - ncc;
		// encourage smoothness, :524-533
		for (int k = 0; k < nClose; ++k)
			score *= SmoothFactor(close[k], depth, normal);
		// geometric consistency, :535-551
		if (!image1.depthMap.empty()) {
			float consistency = 4.f;
			const float Xc[3] = {(float)X0[0] * depth, (float)X0[1] * depth, depth};
			float X1[3];
			for (int i = 0; i < 3; ++i) X1[i] = (image1.Tl[i*3]*Xc[0] + image1.Tl[i*3+1]*Xc[1] + image1.Tl[i*3+2]*Xc[2]) + image1.Tm[i];
			if (X1[2] > 0) {
				const float x1x = X1[0] / X1[2], x1y = X1[1] / X1[2];
				if (insideBorder1(image1.depthMap, x1x, x1y)) {
					float depth1;
					if (sampleDepth(image1.depthMap, x1x, x1y, X1[2], depth1)) {
						const float Xd[3] = {x1x * depth1, x1y * depth1, depth1};
						float Xb[3];
						for (int i = 0; i < 3; ++i) Xb[i] = (image1.Tr[i*3]*Xd[0] + image1.Tr[i*3+1]*Xd[1] + image1.Tr[i*3+2]*Xd[2]) + image1.Tn[i];
						const float xbx = Xb[0] / Xb[2], xby = Xb[1] / Xb[2];
						const float dx = (float)x0x - xbx, dy = (float)x0y - xby;
						const float dist = pm_sqrtf(dx * dx + dy * dy); // norm(Point2f) binds to SEACAVE::norm(const TPoint2<float>&) (Types.inl:1021-1024): float, not cv::norm's double -- pinned by oracle/_ref
						consistency = pm_minf(pm_sqrtf(dist * (dist + 2.f)), consistency);
					}
				}
			}
			score += geoWeight * consistency;
		}
		// low-resolution depth prior, :553-561
		if (lowResDepthMap != nullptr) {
			const float d0 = (*lowResDepthMap)(x0y, x0x);
			if (d0 > 0) {
				const float deltaDepth = pm_minf(pm_fabsf(d0 - depth) / d0, 0.5f); // DepthSimilarity, Util.inl:790-797
				const float sigma = -1.f / (1.f * 0.02f);
				const float factorDeltaDepth = pm_expf(normSq0 * sigma);
				score = (1.f - factorDeltaDepth) * score + factorDeltaDepth * deltaDepth;
			}
		}
		return pm_minf(2.f, score);
	}
	// ScorePixel, DepthMap.cpp:567-626 (AGGNCC_MINMEAN branch :594-611)
	float ScorePixel(float depth, const float* normal) {
		const size_t N = views.size() - 1;
		for (size_t i = 0; i < N; ++i)
			scores[i] = ScorePixelImage(views[i + 1], depth, normal);
		if (idxScore == 0)
			return *std::min_element(scores.begin(), scores.end());
		// GetNth(1) == nth_element: [0] <= [1] <= rest; two smallest as a multiset
		float s0 = scores[0], s1 = scores[1];
		if (s1 < s0) std::swap(s0, s1);
		for (size_t i = 2; i < N; ++i) {
			const float s = scores[i];
			if (s < s0) { s1 = s0; s0 = s; } else if (s < s1) s1 = s;
		}
		int n = 1; float score = s0;
		if (!(s1 >= thRobust)) { score += s1; ++n; }
		return score / (float)n;
	}
	// Normal2Dir / Dir2Normal, libs/Common/Util.inl:754-766
	static void Normal2Dir(const float* d, float* p) { p[0] = pm_atan2f(d[1], d[0]); p[1] = pm_acosf(pm_clampf(d[2], -1.f, 1.f)); }
	static void Dir2Normal(const float* p, float* d) {
		float sx, cx, sy, cy; pm_sincosf(p[0], &sx, &cx); pm_sincosf(p[1], &sy, &cy);
		d[0] = cx * sy; d[1] = sx * sy; d[2] = cy;
	}
	// RandomDepth / RandomNormal, DepthMap.h:435-444 (draw order fixed: depth, theta, phi)
	float RandomDepth() { const float u = rnd.unit(0); const float r = dMinSqr + (dMaxSqr - dMinSqr) * u; return r * r; }
	void RandomNormal(const float* viewRay, float* normal) {
		const float a0 = FD2R(0.f), a1 = FD2R(180.f), b0 = FD2R(90.f), b1 = FD2R(180.f);
		float p[2];
		if (rnd.mode == 2) { p[1] = b0 + (b1 - b0) * rnd.unit(2); p[0] = a0 + (a1 - a0) * rnd.unit(1); }   // GCC evaluates Point2f(a, b)'s arguments right to left
		else { p[0] = a0 + (a1 - a0) * rnd.unit(1); p[1] = b0 + (b1 - b0) * rnd.unit(2); }
		Dir2Normal(p, normal);
		if (normal[0]*viewRay[0] + normal[1]*viewRay[1] + normal[2]*viewRay[2] > 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
	}
	// CorrectNormal, DepthMap.h:447-453 + TRMatrixBase::Set(axis,angle), Rotation.inl:701-728 (float restatement)
	void CorrectNormal(float* normal) const {
		const float v[3] = {(float)X0[0], (float)X0[1], (float)X0[2]};
		const float cosAngLen = normal[0]*v[0] + normal[1]*v[1] + normal[2]*v[2];
		if (cosAngLen >= 0) {
			const float nv = pm_sqrtf(v[0]*v[0] + v[1]*v[1] + v[2]*v[2]);
			const float phi = pm_minf((pm_acosf(pm_clampf(cosAngLen / nv, -1.f, 1.f)) - FD2R(90.f)) * 1.01f, -0.001f);
			float a[3] = {normal[1]*v[2] - normal[2]*v[1], normal[2]*v[0] - normal[0]*v[2], normal[0]*v[1] - normal[1]*v[0]};
			const float an = pm_sqrtf(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]);
			const float ia = 1.f / an;
			a[0] *= ia; a[1] *= ia; a[2] *= ia;
			float s, c; pm_sincosf(phi, &s, &c);
			const float O[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
			float OO[9];
			for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float t = 0; for (int k = 0; k < 3; ++k) t += O[i*3+k]*O[k*3+j]; OO[i*3+j] = t; }
			float Rm[9];
			for (int i = 0; i < 9; ++i) Rm[i] = ((i % 4 == 0 ? 1.f : 0.f) + O[i]*s) + OO[i]*(1.f - c);
			float r[3];
			for (int i = 0; i < 3; ++i) r[i] = Rm[i*3]*normal[0] + Rm[i*3+1]*normal[1] + Rm[i*3+2]*normal[2];
			normal[0] = r[0]; normal[1] = r[1]; normal[2] = r[2];
		}
	}
	// InterpolatePixel, DepthMap.cpp:915-959
	float InterpolatePixel(int nx, int ny, float depth, const float* normal) const {
		const double* K = image0.camera.K;
		float depthNew;
		if (x0x == nx) {
			const float nx1 = (float)(((double)x0y - K[5]) / K[4]);
			const float denom = normal[2] + nx1 * normal[1];
			if (pm_fabsf(denom) < 0.0001f) return depth; // ISZERO, Types.h:1222
			const float x1 = (float)(((double)ny - K[5]) / K[4]);
			const float nom = depth * (normal[2] + x1 * normal[1]);
			depthNew = nom / denom;
		} else {
			const float nx1 = (float)(((double)x0x - K[2]) / K[0]);
			const float denom = normal[2] + nx1 * normal[0];
			if (pm_fabsf(denom) < 0.0001f) return depth;
			const float x1 = (float)(((double)nx - K[2]) / K[0]);
			const float nom = depth * (normal[2] + x1 * normal[0]);
			depthNew = nom / denom;
		}
		return (dMin <= depthNew && depthNew < dMax) ? depthNew : depth; // ISINSIDE, Types.h:1193
	}
	void addClose(int nx, int ny, float ndepth) {
		NeighborEstimate& ne = close[nClose++];
		ne.depth = ndepth;
		const float* nn = nbNormal(nx, ny);
		ne.normal[0] = nn[0]; ne.normal[1] = nn[1]; ne.normal[2] = nn[2];
		// TransformPointI2C(Point3(nx, ndepth)) in double, then Cast<float>; Camera.h:338-344
		const double* K = image0.camera.K;
		const double z = (double)ndepth;
		ne.X[0] = (float)(((double)nx - K[2]) * z / K[0]);
		ne.X[1] = (float)(((double)ny - K[5]) * z / K[4]);
		ne.X[2] = (float)z;
	}
	// ProcessPixel, DepthMap.cpp:630-852 (DENSE_REFINE_ITER branch)
	void ProcessPixel(size_t idx) {
		const std::pair<uint16_t,uint16_t>& c = (dir == LT2RB ? coords[idx] : coords[coords.size() - 1 - idx]);
		if (!PreparePixelPatch(c.first, c.second) || !FillPixelPatch())
			return;
		nNb = 0; nClose = 0;
		const int x = x0x, y = x0y;
		if (dir == LT2RB) {
			if (x > HW)      { const float nd = nbDepth(x-1, y); if (nd > 0) { nbX[nNb] = x-1; nbY[nNb] = y; ++nNb; addClose(x-1, y, nd); } }
			if (y > HW)      { const float nd = nbDepth(x, y-1); if (nd > 0) { nbX[nNb] = x; nbY[nNb] = y-1; ++nNb; addClose(x, y-1, nd); } }
			if (x < W - HW)  { const float nd = nbDepth(x+1, y); if (nd > 0) addClose(x+1, y, nd); }
			if (y < H - HW)  { const float nd = nbDepth(x, y+1); if (nd > 0) addClose(x, y+1, nd); }
		} else {
			if (x < W - HW)  { const float nd = nbDepth(x+1, y); if (nd > 0) { nbX[nNb] = x+1; nbY[nNb] = y; ++nNb; addClose(x+1, y, nd); } }
			if (y < H - HW)  { const float nd = nbDepth(x, y+1); if (nd > 0) { nbX[nNb] = x; nbY[nNb] = y+1; ++nNb; addClose(x, y+1, nd); } }
			if (x > HW)      { const float nd = nbDepth(x-1, y); if (nd > 0) addClose(x-1, y, nd); }
			if (y > HW)      { const float nd = nbDepth(x, y-1); if (nd > 0) addClose(x, y-1, nd); }
		}
		float& conf = confMap0(y, x);
		float& depth = depthMap0(y, x);
		float* normal = normalMap0.at(y, x);
		const float viewDir[3] = {(float)X0[0], (float)X0[1], (float)X0[2]};
		// propagate, :775-799
		for (int n = 0; n < nNb; ++n) {
			if (nbConf(nbX[n], nbY[n]) >= opt.fNCCThresholdKeep)
				continue;
			NeighborEstimate nb = close[n];
			nb.depth = InterpolatePixel(nbX[n], nbY[n], nb.depth, nb.normal);
			CorrectNormal(nb.normal);
			InitPlane(nb.depth, nb.normal);
			const float nconf = ScorePixel(nb.depth, nb.normal);
			if (conf > nconf) { conf = nconf; depth = nb.depth; normal[0] = nb.normal[0]; normal[1] = nb.normal[1]; normal[2] = nb.normal[2]; }
		}
		// refine, :801-852
		static const float scaleRanges[12] = {1.f, 0.5f, 0.25f, 0.125f, 0.0625f, 0.03125f, 0.015625f, 0.0078125f, 0.00390625f, 0.001953125f, 0.0009765625f, 0.00048828125f};
		unsigned idxScaleRange = 0;
	RefineIters:
		if (conf <= thConfSmall)
			idxScaleRange = 2;
		else if (conf <= thConfBig)
			idxScaleRange = 1;
		else if (conf >= thConfRand) {
			nClose = 0;
			for (unsigned iter = 0; iter < opt.nRandomIters; ++iter) {
				rnd.attempt(x, y, STREAM_RAND, (int)iter);
				const float ndepth = RandomDepth();
				float nnormal[3]; RandomNormal(viewDir, nnormal);
				const float nconf = ScorePixel(ndepth, nnormal);
				if (conf > nconf) {
					conf = nconf; depth = ndepth; normal[0] = nnormal[0]; normal[1] = nnormal[1]; normal[2] = nnormal[2];
					if (conf < thConfRand)
						goto RefineIters;
				}
			}
			return;
		}
		float scaleRange = scaleRanges[idxScaleRange];
		const float depthRange = depth * opt.fRandomDepthRatio; // MaxDepthDifference, Util.inl:782-789
		float p[2]; Normal2Dir(normal, p);
		float nnormal[3];
		for (unsigned iter = 0; iter < opt.nRandomIters; ++iter) {
			rnd.attempt(x, y, STREAM_REFINE, (int)iter);
			// randomMeanRange(mean, delta) = mean + delta*(2*U-1), Random.h:137-140
			const float ndepth = depth + (depthRange * scaleRange) * (2.f * rnd.unit(0) - 1.f);
			if (!(dMin <= ndepth && ndepth < dMax))
				continue;
			float np[2];
			if (rnd.mode == 2) { np[1] = p[1] + (angle2Range * scaleRange) * (2.f * rnd.unit(2) - 1.f); np[0] = p[0] + (angle1Range * scaleRange) * (2.f * rnd.unit(1) - 1.f); }
			else { np[0] = p[0] + (angle1Range * scaleRange) * (2.f * rnd.unit(1) - 1.f); np[1] = p[1] + (angle2Range * scaleRange) * (2.f * rnd.unit(2) - 1.f); }
			Dir2Normal(np, nnormal);
			if (nnormal[0]*viewDir[0] + nnormal[1]*viewDir[1] + nnormal[2]*viewDir[2] >= 0)
				continue;
			InitPlane(ndepth, nnormal);
			const float nconf = ScorePixel(ndepth, nnormal);
			if (conf > nconf) {
				conf = nconf; depth = ndepth; normal[0] = nnormal[0]; normal[1] = nnormal[1]; normal[2] = nnormal[2];
				p[0] = np[0]; p[1] = np[1];
				scaleRange = scaleRanges[++idxScaleRange];
			}
		}
	}
};

// MapMatrix2ZigzagIdx, DepthMap.cpp:329-356; mask (nullable, w*hTotal bytes): only pixels with a non-zero entry are listed (:346-347)
static void MapMatrix2ZigzagIdx(int w, int hTotal, std::vector<std::pair<uint16_t,uint16_t>>& coords, int rawStride, const unsigned char* mask = nullptr) {
	const int w1 = w - 1;
	coords.clear(); coords.reserve((size_t)w * hTotal);
	for (int dy = 0, h = rawStride; dy < hTotal; dy += h) {
		if (h * 2 > hTotal - dy)
			h = hTotal - dy;
		int lastX = 0;
		int xx = 0, xy = 0;
		for (int i = 0, ei = w * h; i < ei; ++i) {
			if (!mask || mask[(size_t)(xy + dy) * w + xx])
				coords.emplace_back((uint16_t)xx, (uint16_t)(xy + dy));
			if (xx-- == 0 || ++xy == h) {
				if (++lastX < w) { xx = lastX; xy = 0; }
				else { xx = w1; xy = lastX - w1; }
			}
		}
	}
}

// ScoreDepthMapTmp, SceneDensify.cpp:490-517
static void ScoreDepthMapTmp(DepthEstimator& e) {
	long idx;
	while ((idx = ++e.idxPixel) < (long)e.coords.size()) {
		const int x = e.coords[idx].first, y = e.coords[idx].second;
		if (!e.PreparePixelPatch(x, y) || !e.FillPixelPatch()) {
			e.depthMap0(y,x) = 0; float* n = e.normalMap0.at(y,x); n[0] = n[1] = n[2] = 0; e.confMap0(y,x) = 2.f;
			continue;
		}
		float& depth = e.depthMap0(y,x);
		float* normal = e.normalMap0.at(y,x);
		const float viewDir[3] = {(float)e.X0[0], (float)e.X0[1], (float)e.X0[2]};
		e.rnd.attempt(x, y, STREAM_INIT, 0);
		if (!(e.dMin <= depth && depth < e.dMax)) {
			depth = e.RandomDepth();
			e.RandomNormal(viewDir, normal);
		} else if (normal[0]*viewDir[0] + normal[1]*viewDir[1] + normal[2]*viewDir[2] >= 0) {
			e.RandomNormal(viewDir, normal);
		}
		e.nClose = 0;
		e.confMap0(y,x) = e.ScorePixel(depth, normal);
	}
}
// EstimateDepthMapTmp, SceneDensify.cpp:519-526
static void EstimateDepthMapTmp(DepthEstimator& e) {
	long idx;
	while ((idx = ++e.idxPixel) < (long)e.coords.size())
		e.ProcessPixel((size_t)idx);
}
// EndDepthMapTmp, SceneDensify.cpp:528-576
static void EndDepthMap(DepthData& dd, float thKeep) {
	for (int y = 0; y < dd.depthMap.h; ++y) for (int x = 0; x < dd.depthMap.w; ++x) {
		float& depth = dd.depthMap(y,x); float& conf = dd.confMap(y,x);
		if (depth <= 0 || conf >= thKeep) { conf = 0; depth = 0; float* n = dd.normalMap.at(y,x); n[0] = n[1] = n[2] = 0; }
		else conf = conf >= 1.f ? 0.f : 1.f - conf;
	}
}

// ScaleDepthData, SceneDensify.cpp:578-601 (integer factor f = 2^scaleNumber)
static void ScaleDepthData(const DepthData& in, int f, DepthData& out) {
	out.dMin = in.dMin; out.dMax = in.dMax;
	out.images.resize(in.images.size());
	for (size_t i = 0; i < in.images.size(); ++i) {
		const ViewData& s = in.images[i]; ViewData& v = out.images[i];
		resizeArea(s.image, f, v.image);
		v.camera = s.camera;
		scaleK(s.camera.K, s.image.w, s.image.h, v.image.w, v.image.h, v.camera.K);
		if (!s.depthMap.empty()) {
			resizeArea(s.depthMap, f, v.depthMap);
			v.cameraDepthMap = s.cameraDepthMap;
			scaleK(s.cameraDepthMap.K, s.depthMap.w, s.depthMap.h, v.image.w, v.image.h, v.cameraDepthMap.K);
		}
	}
	for (size_t i = 0; i < out.images.size(); ++i) out.images[i].Init(out.images[0].camera);
	if (!in.depthMap.empty()) resizeNearestDown(in.depthMap, f, out.depthMap);
	if (!in.normalMap.empty()) resizeNearestDownN(in.normalMap, f, out.normalMap);
}

template <typename F>
static void runThreads(int T, std::vector<DepthEstimator*>& est, F fn) {
	if (T <= 1) { fn(*est[0]); return; }
	std::vector<std::thread> th;
	for (int i = 1; i < T; ++i) th.emplace_back([&, i]() { fn(*est[i]); });
	fn(*est[0]);
	for (auto& t : th) t.join();
}

// DepthMapsData::EstimateDepthMap, SceneDensify.cpp:616-805
// nGeometricIter < 0: photometric pass (pyramid); >= 0: one geometric-consistency round.
// ignoreMask (nullable): W0*H0 bytes at image resolution, 0 = ignore (the BitMatrix of ImportIgnoreMask, DepthMap.cpp:296-323);
// maskMode: OPTDENSE::nIgnoreMaskLabel >= 0 (selects INTER_NEAREST for the depth hand-off, SceneDensify.cpp:661).
static int EstimateDepthMap(DepthData& full, const Opt& opt, int nGeometricIter,
		void (*levelHook)(void*, int, int, const DepthData&), void* hookArg, const unsigned char* ignoreMask = nullptr, bool maskMode = false) {
	const int T = std::max(1, opt.nThreads);
	const unsigned iterBegin = nGeometricIter < 0 ? 0u : opt.nEstimationIters + (unsigned)nGeometricIter;
	const unsigned iterEnd = nGeometricIter < 0 ? opt.nEstimationIters : iterBegin + 1;
	const unsigned totalScaleNumber = nGeometricIter < 0 ? opt.nSubResolutionLevels : 0u;
	const int W0 = full.images[0].image.w, H0 = full.images[0].image.h;
	if (scaledSize(W0, 1 << totalScaleNumber) < 2 * DepthEstimator::HW + 1 || scaledSize(H0, 1 << totalScaleNumber) < 2 * DepthEstimator::HW + 1) return -2;
	for (auto& v : full.images) v.Init(full.images[0].camera);
	ImgF lowResDepthMap; ImgN lowResNormalMap;
	std::vector<Weight> weightMap0;
	ImgF currentSizeResDepthMap;
	std::vector<std::pair<uint16_t,uint16_t>> coords;
	std::atomic<long> idxPixel;
	for (unsigned scaleNumber = totalScaleNumber + 1; scaleNumber-- > 0; ) {
		DepthData currentDepthData;
		if (scaleNumber > 0) ScaleDepthData(full, 1 << scaleNumber, currentDepthData);
		DepthData& dd = scaleNumber == 0 ? full : currentDepthData;
		const int w = dd.images[0].image.w, h = dd.images[0].image.h;
		if (scaleNumber != totalScaleNumber) {
			if (maskMode) resizeNearest(lowResDepthMap, w, h, dd.depthMap); else resizeLinear(lowResDepthMap, w, h, dd.depthMap);
			resizeNearestN(lowResNormalMap, w, h, dd.normalMap);
			currentSizeResDepthMap = dd.depthMap;
		} else if (totalScaleNumber > 0) {
			full.depthMap.d.clear(); full.normalMap.d.clear(); full.confMap.d.clear();
		}
		if (dd.depthMap.empty()) dd.depthMap.create(w, h);   // InitViews creates a zero depth map, SceneDensify.cpp:420
		if (dd.normalMap.empty()) dd.normalMap.create(w, h);
		dd.confMap.create(w, h);
		weightMap0.assign((size_t)w * h, Weight());
		std::vector<unsigned char> levelMask;
		if (ignoreMask) {   // cv::resize(mask, size, INTER_NEAREST), then DepthData::ApplyIgnoreMask (DepthMap.cpp:215-231)
			levelMask.resize((size_t)w * h);
			const double ifx = (double)W0 / w, ify = (double)H0 / h;
			for (int y = 0; y < h; ++y) { const int sy = std::min((int)floor(y * ify), H0 - 1);
				for (int x = 0; x < w; ++x) { const int sx = std::min((int)floor(x * ifx), W0 - 1); levelMask[(size_t)y * w + x] = ignoreMask[(size_t)sy * W0 + sx]; } }
			for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) if (!levelMask[(size_t)y * w + x]) {
				dd.depthMap(y, x) = 0; float* n = dd.normalMap.at(y, x); n[0] = n[1] = n[2] = 0; dd.confMap(y, x) = 0;
			}
		}
		MapMatrix2ZigzagIdx(w, h, coords, std::max(64, T * 8), ignoreMask ? levelMask.data() : nullptr);
		const ImgF* prior = currentSizeResDepthMap.empty() ? nullptr : &currentSizeResDepthMap;
		const uint32_t level = scaleNumber;
		auto makeEstimators = [&](unsigned iter, uint32_t pass, std::vector<std::unique_ptr<DepthEstimator>>& own, std::vector<DepthEstimator*>& ptrs) {
			uint32_t k0, k1; passKey(opt.seed, opt.viewID, pass, k0, k1);
			for (int t = 0; t < T; ++t) {
				own.emplace_back(new DepthEstimator(iter, dd, idxPixel, weightMap0, coords, opt, k0, k1));
				own.back()->lowResDepthMap = prior;
				ptrs.push_back(own.back().get());
			}
		};
		{ // pass A: init scores
			idxPixel = -1;
			std::vector<std::unique_ptr<DepthEstimator>> own; std::vector<DepthEstimator*> est;
			makeEstimators(iterBegin, level * 64 + 32 + (nGeometricIter < 0 ? 0 : 16 + nGeometricIter), own, est);
			runThreads(T, est, ScoreDepthMapTmp);
		}
		if (levelHook) levelHook(hookArg, (int)scaleNumber, -1, dd);
		ImgF oldDepth, oldConf; ImgN oldNormal;
		for (unsigned iter = iterBegin; iter < iterEnd; ++iter) { // pass B: sweeps
			idxPixel = -1;
			std::vector<std::unique_ptr<DepthEstimator>> own; std::vector<DepthEstimator*> est;
			makeEstimators(iter, level * 64 + iter, own, est);
			if (opt.tileW > 0 && opt.tileH > 0) {   // tiled sweeps: what the sweep finds, for the reads across tile borders
				oldDepth = dd.depthMap; oldNormal = dd.normalMap; oldConf = dd.confMap;
				for (DepthEstimator* e : est) { e->oldDepth = &oldDepth; e->oldNormal = &oldNormal; e->oldConf = &oldConf; }
			}
			runThreads(T, est, EstimateDepthMapTmp);
			if (levelHook) levelHook(hookArg, (int)scaleNumber, (int)iter, dd);
		}
		if (scaleNumber > 0) { lowResDepthMap = dd.depthMap; lowResNormalMap = dd.normalMap; }
	}
	// pass C: finalize, SceneDensify.cpp:771-797
	float th = opt.fNCCThresholdKeep;
	if (nGeometricIter < 0 && opt.nEstimationGeometricIters) th *= 1.333f;
	EndDepthMap(full, th);
	return 0;
}

} // namespace orc

// ---------------------------------------------------------------------------
// C interface for tests / bench (ctypes).  Layout mirrors include/pmhip.h's PODs on purpose
// so the same seeded inputs can be handed to both sides.
extern "C" {

struct OrcView {
	const float* image; int w, h;
	double K[9], R[9], C[3];
	const float* depth;          // nullable; source view's known depth-map (geometric pass)
	double Kd[9], Rd[9], Cd[3];
	int dw, dh;                  // size of that depth-map (0, 0 = the image's size): it is read with cameraDepthMap, not with the image's camera (DepthMap.h:170-171)
};
struct OrcOpt {
	uint32_t nSubResolutionLevels, nEstimationIters, nEstimationGeometricIters, nRandomIters;
	float fEstimationGeometricWeight, fRandomDepthRatio, fRandomAngle1Range, fRandomAngle2Range;
	float fRandomSmoothDepth, fRandomSmoothNormal, fRandomSmoothBonus, fNCCThresholdKeep, fDescriptorMinMagnitudeThreshold;
	uint32_t seed, viewID; int32_t rngMode, nThreads;
	int32_t tileW, tileH;        // tiled sweeps (orc::Opt::tileW); 0 = off
};
static orc::Opt toOpt(const OrcOpt* o) {
	orc::Opt r;
	r.nSubResolutionLevels = o->nSubResolutionLevels; r.nEstimationIters = o->nEstimationIters;
	r.nEstimationGeometricIters = o->nEstimationGeometricIters; r.nRandomIters = o->nRandomIters;
	r.fEstimationGeometricWeight = o->fEstimationGeometricWeight; r.fRandomDepthRatio = o->fRandomDepthRatio;
	r.fRandomAngle1Range = o->fRandomAngle1Range; r.fRandomAngle2Range = o->fRandomAngle2Range;
	r.fRandomSmoothDepth = o->fRandomSmoothDepth; r.fRandomSmoothNormal = o->fRandomSmoothNormal;
	r.fRandomSmoothBonus = o->fRandomSmoothBonus; r.fNCCThresholdKeep = o->fNCCThresholdKeep;
	r.fDescriptorMinMagnitudeThreshold = o->fDescriptorMinMagnitudeThreshold;
	r.seed = o->seed; r.viewID = o->viewID; r.rngMode = o->rngMode; r.nThreads = o->nThreads;
	r.tileW = o->tileW; r.tileH = o->tileH;
	return r;
}
void orc_default_opt(OrcOpt* o) {
	orc::Opt d;
	o->nSubResolutionLevels = d.nSubResolutionLevels; o->nEstimationIters = d.nEstimationIters;
	o->nEstimationGeometricIters = d.nEstimationGeometricIters; o->nRandomIters = d.nRandomIters;
	o->fEstimationGeometricWeight = d.fEstimationGeometricWeight; o->fRandomDepthRatio = d.fRandomDepthRatio;
	o->fRandomAngle1Range = d.fRandomAngle1Range; o->fRandomAngle2Range = d.fRandomAngle2Range;
	o->fRandomSmoothDepth = d.fRandomSmoothDepth; o->fRandomSmoothNormal = d.fRandomSmoothNormal;
	o->fRandomSmoothBonus = d.fRandomSmoothBonus; o->fNCCThresholdKeep = d.fNCCThresholdKeep;
	o->fDescriptorMinMagnitudeThreshold = d.fDescriptorMinMagnitudeThreshold;
	o->seed = 0; o->viewID = 0; o->rngMode = 0; o->nThreads = 1; o->tileW = 0; o->tileH = 0;
}

static void loadDepthData(const OrcView* views, int nViews, const float* depth, const float* normal, float dMin, float dMax, orc::DepthData& dd) {
	dd.images.resize(nViews);
	for (int i = 0; i < nViews; ++i) {
		orc::ViewData& v = dd.images[i]; const OrcView& s = views[i];
		v.image.create(s.w, s.h); memcpy(v.image.d.data(), s.image, sizeof(float) * s.w * s.h);
		memcpy(v.camera.K, s.K, 72); memcpy(v.camera.R, s.R, 72); memcpy(v.camera.C, s.C, 24);
		if (i > 0 && s.depth) {
			const int dw = s.dw > 0 ? s.dw : s.w, dh = s.dh > 0 ? s.dh : s.h;
			v.depthMap.create(dw, dh); memcpy(v.depthMap.d.data(), s.depth, sizeof(float) * dw * dh);
			memcpy(v.cameraDepthMap.K, s.Kd, 72); memcpy(v.cameraDepthMap.R, s.Rd, 72); memcpy(v.cameraDepthMap.C, s.Cd, 24);
		}
	}
	const int w = views[0].w, h = views[0].h;
	if (depth) { dd.depthMap.create(w, h); memcpy(dd.depthMap.d.data(), depth, sizeof(float) * w * h); }
	if (normal) { dd.normalMap.create(w, h); memcpy(dd.normalMap.d.data(), normal, sizeof(float) * w * h * 3); }
	dd.dMin = dMin; dd.dMax = dMax;
}

struct HookCtx { float* levelDump; size_t cap; size_t used; };
static void dumpHook(void* a, int level, int iter, const orc::DepthData& dd) {
	// appends [level, iter, w, h] + depth + normal + conf (cost) for debugging / per-stage parity
	HookCtx* c = (HookCtx*)a;
	const size_t n = (size_t)dd.depthMap.w * dd.depthMap.h;
	if (c->used + 4 + n * 5 > c->cap) return;
	float* p = c->levelDump + c->used;
	p[0] = (float)level; p[1] = (float)iter; p[2] = (float)dd.depthMap.w; p[3] = (float)dd.depthMap.h;
	memcpy(p + 4, dd.depthMap.d.data(), n * 4); memcpy(p + 4 + n, dd.normalMap.d.data(), n * 12); memcpy(p + 4 + n * 4, dd.confMap.d.data(), n * 4);
	c->used += 4 + n * 5;
}

// One EstimateDepthMap call (SceneDensify.cpp:616): depth/normal are in/out (may be zero-filled = "unset"),
// conf is out.  stageDump (nullable) receives per-level/per-iteration snapshots, stageCap floats.
int orc_estimate_depth_map(const OrcView* views, int nViews, float* depth, float* normal, float* conf,
		float dMin, float dMax, const OrcOpt* opt, int nGeometricIter, float* stageDump, size_t stageCap, size_t* stageUsed) {
	if (nViews < 2) return -1;
	orc::DepthData dd; loadDepthData(views, nViews, depth, normal, dMin, dMax, dd);
	orc::Opt o = toOpt(opt);
	HookCtx ctx{stageDump, stageCap, 0};
	const int rc = orc::EstimateDepthMap(dd, o, nGeometricIter, stageDump ? dumpHook : nullptr, &ctx);
	if (rc) return rc;
	const size_t n = (size_t)views[0].w * views[0].h;
	memcpy(depth, dd.depthMap.d.data(), n * 4); memcpy(normal, dd.normalMap.d.data(), n * 12); memcpy(conf, dd.confMap.d.data(), n * 4);
	if (stageUsed) *stageUsed = ctx.used;
	return 0;
}

// Same with OPTDENSE::nIgnoreMaskLabel >= 0: mask (nullable) = the reference view's ignore mask at image resolution, 0 = ignore.
int orc_estimate_depth_map_masked(const OrcView* views, int nViews, float* depth, float* normal, float* conf,
		float dMin, float dMax, const OrcOpt* opt, int nGeometricIter, const unsigned char* mask, int maskMode) {
	if (nViews < 2) return -1;
	orc::DepthData dd; loadDepthData(views, nViews, depth, normal, dMin, dMax, dd);
	orc::Opt o = toOpt(opt);
	const int rc = orc::EstimateDepthMap(dd, o, nGeometricIter, nullptr, nullptr, mask, maskMode != 0);
	if (rc) return rc;
	const size_t n = (size_t)views[0].w * views[0].h;
	memcpy(depth, dd.depthMap.d.data(), n * 4); memcpy(normal, dd.normalMap.d.data(), n * 12); memcpy(conf, dd.confMap.d.data(), n * 4);
	return 0;
}

// The small per-pixel helpers of ProcessPixel at pixel (x, y), for the second-reading tests: InterpolatePixel of the neighbour estimate
// (nx, ny, ndepth, nnormal) to (x, y), CorrectNormal of that normal, and the smoothness factor of the hypothesis plane (hypDepth, hypNormal)
// with respect to that neighbour.
// One pyramid level with one estimator thread, pass by pass -- the unit oracle/_ref (the reference's own code, oracle/ref/ref_harness.cpp:ref_run_level)
// exposes, with the same arguments: init pass (ScoreDepthMapTmp) if doInit, sweeps iterBegin..iterEnd-1, EndDepthMapTmp with threshold thEnd if thEnd >= 0.
// pass keys of the Philox mode as in EstimateDepthMap at level 0; with rngMode 1 / 2 every estimator restarts its mt19937 from the default seed.
int orc_run_level(const OrcView* views, int nViews, float* depth, float* normal, float* conf, const float* prior,
		float dMin, float dMax, const OrcOpt* opt, int doInit, unsigned iterBegin, unsigned iterEnd, float thEnd, const unsigned char* mask) {
	if (nViews < 2) return -1;
	orc::DepthData dd; loadDepthData(views, nViews, depth, normal, dMin, dMax, dd);
	const orc::Opt o = toOpt(opt);
	const int w = views[0].w, h = views[0].h;
	for (auto& v : dd.images) v.Init(dd.images[0].camera);
	dd.confMap.create(w, h); memcpy(dd.confMap.d.data(), conf, sizeof(float) * (size_t)w * h);
	orc::ImgF pr; if (prior) { pr.create(w, h); memcpy(pr.d.data(), prior, sizeof(float) * (size_t)w * h); }
	std::vector<orc::Weight> weightMap0((size_t)w * h);
	if (mask) for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) if (!mask[(size_t)y * w + x]) {
		dd.depthMap(y, x) = 0; float* n = dd.normalMap.at(y, x); n[0] = n[1] = n[2] = 0; dd.confMap(y, x) = 0; }
	std::vector<std::pair<uint16_t,uint16_t>> coords;
	orc::MapMatrix2ZigzagIdx(w, h, coords, 64, mask);
	std::atomic<long> idxPixel;
	auto run = [&](unsigned iter, uint32_t pass, void (*fn)(orc::DepthEstimator&)) {
		uint32_t k0, k1; orc::passKey(o.seed, o.viewID, pass, k0, k1);
		idxPixel = -1;
		orc::DepthEstimator e(iter, dd, idxPixel, weightMap0, coords, o, k0, k1);
		e.lowResDepthMap = prior ? &pr : nullptr;
		fn(e);
	};
	if (doInit) run(iterBegin, 32, orc::ScoreDepthMapTmp);
	for (unsigned iter = iterBegin; iter < iterEnd; ++iter) run(iter, iter, orc::EstimateDepthMapTmp);
	if (thEnd >= 0) {
		// EndDepthMapTmp walks the pixel list (SceneDensify.cpp:533): pixels outside it are left as they are
		for (const auto& c : coords) {
			const int x = c.first, y = c.second;
			float& d = dd.depthMap(y, x); float& cf = dd.confMap(y, x);
			if (d <= 0 || cf >= thEnd) { cf = 0; d = 0; float* n = dd.normalMap.at(y, x); n[0] = n[1] = n[2] = 0; }
			else cf = cf >= 1.f ? 0.f : 1.f - cf;
		}
	}
	const size_t n = (size_t)w * h;
	memcpy(depth, dd.depthMap.d.data(), n * 4); memcpy(normal, dd.normalMap.d.data(), n * 12); memcpy(conf, dd.confMap.d.data(), n * 4);
	return 0;
}

int orc_pixel_helpers(const OrcView* views, int nViews, const OrcOpt* opt, int x, int y, float dMin, float dMax, int nx, int ny, float ndepth, const float* nnormal,
		float hypDepth, const float* hypNormal, float* outInterpDepth, float* outCorrected, float* outSmoothFactor) {
	orc::DepthData dd; loadDepthData(views, nViews, nullptr, nullptr, dMin, dMax, dd);
	for (auto& v : dd.images) v.Init(dd.images[0].camera);
	const int w = views[0].w, h = views[0].h;
	dd.depthMap.create(w, h); dd.normalMap.create(w, h); dd.confMap.create(w, h);
	float* nn = dd.normalMap.at(ny, nx); nn[0] = nnormal[0]; nn[1] = nnormal[1]; nn[2] = nnormal[2];
	orc::Opt o = toOpt(opt);
	std::vector<orc::Weight> wm((size_t)w * h);
	std::vector<std::pair<uint16_t,uint16_t>> coords;
	std::atomic<long> idx(-1);
	orc::DepthEstimator e(0, dd, idx, wm, coords, o, 0, 0);
	if (!e.PreparePixelPatch(x, y)) return 1;
	e.FillPixelPatch();                      // sets X0 (its texture verdict does not matter here)
	*outInterpDepth = e.InterpolatePixel(nx, ny, ndepth, nnormal);
	outCorrected[0] = nnormal[0]; outCorrected[1] = nnormal[1]; outCorrected[2] = nnormal[2];
	e.CorrectNormal(outCorrected);
	e.nClose = 0; e.addClose(nx, ny, ndepth);
	e.InitPlane(hypDepth, hypNormal);
	*outSmoothFactor = e.SmoothFactor(e.close[0], hypDepth, hypNormal);
	return 0;
}

// Score one plane hypothesis at one pixel at full resolution, no neighbours (known-answer tests).
int orc_score_pixel(const OrcView* views, int nViews, const OrcOpt* opt, int x, int y, float depthv, const float* normalv,
		const float* prior, float* outScores, float* outAgg) {
	orc::DepthData dd; loadDepthData(views, nViews, nullptr, nullptr, 0.1f, 100.f, dd);
	for (auto& v : dd.images) v.Init(dd.images[0].camera);
	const int w = views[0].w, h = views[0].h;
	dd.depthMap.create(w, h); dd.normalMap.create(w, h); dd.confMap.create(w, h);
	orc::Opt o = toOpt(opt);
	std::vector<orc::Weight> wm((size_t)w * h);
	std::vector<std::pair<uint16_t,uint16_t>> coords;
	std::atomic<long> idx(-1);
	orc::DepthEstimator e(0, dd, idx, wm, coords, o, 0, 0);
	orc::ImgF pr;
	if (prior) { pr.create(w, h); memcpy(pr.d.data(), prior, sizeof(float) * w * h); e.lowResDepthMap = &pr; }
	if (!e.PreparePixelPatch(x, y) || !e.FillPixelPatch()) return 1;
	e.InitPlane(depthv, normalv);
	*outAgg = e.ScorePixel(depthv, normalv);
	for (int i = 0; i < nViews - 1; ++i) outScores[i] = e.scores[i];
	return 0;
}

void orc_zigzag(int w, int h, int rawStride, uint16_t* outXY) {
	std::vector<std::pair<uint16_t,uint16_t>> c; orc::MapMatrix2ZigzagIdx(w, h, c, rawStride);
	for (size_t i = 0; i < c.size(); ++i) { outXY[2*i] = c[i].first; outXY[2*i+1] = c[i].second; }
}
void orc_resize_area(const float* s, int w, int h, int f, float* o) {
	orc::ImgF a, b; a.create(w, h); memcpy(a.d.data(), s, sizeof(float) * w * h); orc::resizeArea(a, f, b); memcpy(o, b.d.data(), sizeof(float) * b.w * b.h);
}
void orc_resize_nearest_down(const float* s, int w, int h, int f, float* o) {
	orc::ImgF a, b; a.create(w, h); memcpy(a.d.data(), s, sizeof(float) * w * h); orc::resizeNearestDown(a, f, b); memcpy(o, b.d.data(), sizeof(float) * b.w * b.h);
}
int orc_scaled_size(int n, int f) { return orc::scaledSize(n, f);
}
void orc_resize_linear(const float* s, int w, int h, int nw, int nh, float* o) {
	orc::ImgF a, b; a.create(w, h); memcpy(a.d.data(), s, sizeof(float) * w * h); orc::resizeLinear(a, nw, nh, b); memcpy(o, b.d.data(), sizeof(float) * nw * nh);
}
void orc_resize_nearest(const float* s, int w, int h, int nw, int nh, float* o) {
	orc::ImgF a, b; a.create(w, h); memcpy(a.d.data(), s, sizeof(float) * w * h); orc::resizeNearest(a, nw, nh, b); memcpy(o, b.d.data(), sizeof(float) * nw * nh);
}
// pm_math.h on the host: kind 0 exp, 1 acos, 2 atan2(a,b), 3 sin, 4 cos, 5 sqrt, 6 a/b
void orc_math_eval(int kind, const float* a, const float* b, float* o, size_t n) {
	for (size_t i = 0; i < n; ++i) {
		float s, c;
		switch (kind) {
		case 0: o[i] = pm_expf(a[i]); break;
		case 1: o[i] = pm_acosf(a[i]); break;
		case 2: o[i] = pm_atan2f(a[i], b[i]); break;
		case 3: pm_sincosf(a[i], &s, &c); o[i] = s; break;
		case 4: pm_sincosf(a[i], &s, &c); o[i] = c; break;
		case 5: o[i] = pm_sqrtf(a[i]); break;
		case 7: o[i] = pm_hypot_d(a[i], b[i]); break;
		case 8: o[i] = a[i] / b[i]; break;                 // reference value of pm_div2's first quotient
		case 9: o[i] = a[i] / (b[i] + a[i]); break;        // ... and of its second quotient, other operand roles
		default: o[i] = a[i] / b[i]; break;
		}
	}
}
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4) {
	PmPhilox4 p = pm_philox4x32_10(c0, c1, c2, c3, k0, k1); for (int i = 0; i < 4; ++i) out4[i] = p.v[i];
}
void orc_view_init(const OrcView* ref, const OrcView* src, double* Hl, double* Hm, double* Hr, float* Tl, float* Tm, float* Tr, float* Tn) {
	orc::ViewData v; orc::Cam r;
	memcpy(r.K, ref->K, 72); memcpy(r.R, ref->R, 72); memcpy(r.C, ref->C, 24);
	memcpy(v.camera.K, src->K, 72); memcpy(v.camera.R, src->R, 72); memcpy(v.camera.C, src->C, 24);
	if (src->depth) { v.depthMap.create(1, 1); memcpy(v.cameraDepthMap.K, src->Kd, 72); memcpy(v.cameraDepthMap.R, src->Rd, 72); memcpy(v.cameraDepthMap.C, src->Cd, 24); }
	v.Init(r);
	memcpy(Hl, v.Hl, 72); memcpy(Hm, v.Hm, 24); memcpy(Hr, v.Hr, 72);
	if (src->depth) { memcpy(Tl, v.Tl, 36); memcpy(Tm, v.Tm, 12); memcpy(Tr, v.Tr, 36); memcpy(Tn, v.Tn, 12); }
}

} // extern "C"
