// filter_oracle.cpp -- CPU restatement of DepthMapsData::FilterDepthMap (libs/MVS/SceneDensify.cpp:1050-1299 in
// /root/reference): forward-splat the neighbours' depth maps into the reference view with a z-test on the 4 surrounding
// pixels (:1084-1128), then a per-pixel confidence-weighted vote (:1141-1212, bFilterAdjust) or a strict agreement test
// (:1213-1290).  *** TEST INFRASTRUCTURE ONLY *** (see pm_oracle.cpp).  PARITY UNPINNED by the reference.
// Camera maths in double with cv::Matx's accumulation order (libs/MVS/Camera.h:338-399).
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace flt {
struct Cam { double K[9], R[9], C[3]; };
static inline void mulMV(const double* M, const double* v, double* o) { for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += M[i*3+k] * v[k]; o[i] = s; } }
static inline void mulMtV(const double* M, const double* v, double* o) { for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += M[k*3+i] * v[k]; o[i] = s; } }
// Camera::TransformPointI2W(Point3(x,y,depth)) = R^T * I2C(X) + C, Camera.h:338-356
static inline void I2W(const Cam& c, double x, double y, double z, double* X) {
	const double ci[3] = {(x - c.K[2]) * z / c.K[0], (y - c.K[5]) * z / c.K[4], z};
	double t[3]; mulMtV(c.R, ci, t);
	for (int i = 0; i < 3; ++i) X[i] = t[i] + c.C[i];
}
// Camera::TransformPointW2C, Camera.h:388-390
static inline void W2C(const Cam& c, const double* X, double* o) { const double d[3] = {X[0] - c.C[0], X[1] - c.C[1], X[2] - c.C[2]}; mulMV(c.R, d, o); }
// Camera::TransformPointC2I(Point3), Camera.h:370-386
static inline void C2I(const Cam& c, const double* X, double* o) { o[0] = c.K[2] + c.K[0] * (X[0] / X[2]); o[1] = c.K[5] + c.K[4] * (X[1] / X[2]); }
static inline bool similar(float d0, float d1, float th) { return fabsf(d0 - d1) / d0 < th; } // IsDepthSimilar, Util.inl:798-809
}

extern "C" {

struct FltView { const float* depth; const float* conf; double K[9], R[9], C[3]; int w, h; };   // w, h: size of this view's maps; 0 = the reference view's (w, h of the call)

// ref: reference view; nb[0..N): its (valid) neighbour views in neighbour-list order (at most 8, SceneDensify.cpp:2152-2167).
// Returns 0 and fills newDepth/newConf, or 1 if the view cannot be filtered (N < nMinViews, :1060-1063).
int orc_filter_depth_map(const FltView* ref, const FltView* nb, int N, int w, int h, float dMin, float dMax, int bAdjust,
		unsigned nMinViewsFilter, unsigned nMinViewsFilterAdjust, unsigned nCalibratedImages, float fDepthDiffThreshold,
		float* newDepth, float* newConf) {
	using namespace flt;
	const unsigned nMinViews = std::min(nMinViewsFilter, nCalibratedImages - 1), nMinViewsAdjust = std::min(nMinViewsFilterAdjust, nCalibratedImages - 1);
	if ((unsigned)N < nMinViews || (unsigned)N < nMinViewsAdjust) return 1;
	Cam cref; memcpy(cref.K, ref->K, 72); memcpy(cref.R, ref->R, 72); memcpy(cref.C, ref->C, 24);
	const size_t P = (size_t)w * h;
	std::vector<std::vector<float>> depthMaps(N, std::vector<float>(P, 0.f)), confMaps(N, std::vector<float>(P, 0.f));
	std::vector<Cam> cams(N);
	for (int n = 0; n < N; ++n) {
		Cam& cam = cams[n]; memcpy(cam.K, nb[n].K, 72); memcpy(cam.R, nb[n].R, 72); memcpy(cam.C, nb[n].C, 24);
		const int nw = nb[n].w ? nb[n].w : w, nh = nb[n].h ? nb[n].h : h;   // the neighbour's own depth-map size (:1085: depthData.depthMap.size())
		for (int i = 0; i < nh; ++i) for (int j = 0; j < nw; ++j) {
			const float depth = nb[n].depth[(size_t)i * nw + j];
			if (depth == 0) continue;
			double X[3], camX[3], imgX[2];
			I2W(cam, (double)j, (double)i, (double)depth, X);
			W2C(cref, X, camX);
			if (camX[2] <= 0) continue;
			C2I(cref, camX, imgX);
			const int xs[2] = {(int)floor(imgX[0]), (int)ceil(imgX[0])}, ys[2] = {(int)floor(imgX[1]), (int)ceil(imgX[1])};
			const int px[4] = {xs[0], xs[0], xs[1], xs[1]}, py[4] = {ys[0], ys[1], ys[0], ys[1]};
			for (int p = 0; p < 4; ++p) {
				if (!(px[p] >= 0 && py[p] >= 0 && px[p] < w && py[p] < h)) continue;
				float& depthRef = depthMaps[n][(size_t)py[p] * w + px[p]];
				if (depthRef != 0 && depthRef < (float)camX[2]) continue;
				depthRef = (float)camX[2];
				if (bAdjust) confMaps[n][(size_t)py[p] * w + px[p]] = nb[n].conf[(size_t)i * nw + j];
			}
		}
	}
	const float thDepthDiff = fDepthDiffThreshold * 1.2f;
	if (bAdjust) {
		for (int i = 0; i < h; ++i) for (int j = 0; j < w; ++j) {
			const size_t xr = (size_t)i * w + j;
			const float depth = ref->depth[xr];
			if (depth == 0) { newDepth[xr] = 0; newConf[xr] = 0; continue; }
			float posConf = ref->conf[xr], negConf = 0;
			float avgDepth = depth * posConf;
			unsigned nPosViews = 0, nNegViews = 0;
			unsigned n = (unsigned)N;
			bool discard = false;
			do {
				const float d = depthMaps[--n][xr];
				if (d == 0) {
					if (nPosViews + nNegViews + n < nMinViews) { discard = true; break; }
					continue;
				}
				if (similar(depth, d, thDepthDiff)) {
					const float c = confMaps[n][xr];
					avgDepth += d * c; posConf += c; ++nPosViews;
				} else {
					if (depth > d) negConf += confMaps[n][xr];
					else {
						double X[3], cx[3], ix[2];
						I2W(cref, (double)j, (double)i, (double)depth, X);
						W2C(cams[n], X, cx); C2I(cams[n], cx, ix);
						const int x = (int)floor(ix[0] + .5), y = (int)floor(ix[1] + .5); // ROUND2INT(double)
						const int nw = nb[n].w ? nb[n].w : w, nh = nb[n].h ? nb[n].h : h;   // depthData.confMap.isInside(x), :1181
						if (x >= 0 && y >= 0 && x < nw && y < nh) { const float c = nb[n].conf[(size_t)y * nw + x]; negConf += (c > 0 ? c : confMaps[n][xr]); }
						else negConf += confMaps[n][xr];
					}
					++nNegViews;
				}
			} while (n);
			if (!discard && nPosViews >= nMinViewsAdjust && posConf > negConf) {
				avgDepth /= posConf;
				if (dMin <= avgDepth && avgDepth < dMax) { newDepth[xr] = avgDepth; newConf[xr] = posConf - negConf; continue; }
			}
			newDepth[xr] = 0; newConf[xr] = 0;
		}
	} else {
		const float thDepthDiffStrict = fDepthDiffThreshold * 0.8f;
		const unsigned nMinGoodViewsProc = 75, nMinGoodViewsDeltaProc = 65, nDeltas = 4;
		const unsigned nMinViewsDelta = nMinViews * (nDeltas - 2);
		const int dxs[4] = {-1, 1, 0, 0}, dys[4] = {0, 0, -1, 1};
		for (int i = 0; i < h; ++i) for (int j = 0; j < w; ++j) {
			const size_t xr = (size_t)i * w + j;
			const float depth = ref->depth[xr];
			newDepth[xr] = 0; newConf[xr] = 0;
			if (depth == 0) continue;
			unsigned nGood = 0, nViews = 0;
			for (int n = N; n-- > 0; ) { const float d = depthMaps[n][xr]; if (d > 0) { ++nViews; if (similar(depth, d, thDepthDiffStrict)) ++nGood; } }
			if (nGood < nMinViews || nGood < nViews * nMinGoodViewsProc / 100) continue;
			nGood = 0; nViews = 0;
			for (unsigned dd = 0; dd < nDeltas; ++dd) {
				const int x = j + dxs[dd], y = i + dys[dd];
				if (!(x >= 0 && y >= 0 && x < w && y < h)) continue; // the reference reads out of bounds here; unreachable for estimated maps (4-px empty border)
				for (int n = N; n-- > 0; ) { const float d = depthMaps[n][(size_t)y * w + x]; if (d > 0) { ++nViews; if (similar(depth, d, thDepthDiff)) ++nGood; } }
			}
			if (nGood < nMinViewsDelta || nGood < nViews * nMinGoodViewsDeltaProc / 100) continue;
			newDepth[xr] = depth; newConf[xr] = ref->conf[xr];
		}
	}
	return 0;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// DepthMapsData::GapInterpolation, libs/MVS/SceneDensify.cpp:904-1045: fill row gaps then column gaps of at most
// nIpolGapSize invalid pixels between two similar depths (threshold fDepthDiffThreshold*2.5) by linear interpolation of
// depth and of the normal's direction angles; confidence = min of the two ends.  In place, literal transcription.
#include "../openmvs_amd/csrc/pm_math.h"
namespace flt {
static inline void Normal2Dir(const float* d, float* p) { p[0] = pm_atan2f(d[1], d[0]); p[1] = pm_acosf(pm_clampf(d[2], -1.f, 1.f)); } // Util.inl:754-759
static inline void Dir2Normal(const float* p, float* d) { float sx, cx, sy, cy; pm_sincosf(p[0], &sx, &cx); pm_sincosf(p[1], &sy, &cy); d[0] = cx * sy; d[1] = sx * sy; d[2] = cy; }
static void gapPass(float* depth, float* normal, float* conf, int w, int h, bool rows, unsigned nIpolGapSize, float th) {
	const int outer = rows ? h : w, inner = rows ? w : h;
	for (int o = 0; o < outer; ++o) {
		unsigned count = 0;
		for (int u = 0; u < inner; ++u) {
			const size_t at = rows ? (size_t)o * w + u : (size_t)u * w + o;
			const float d1 = depth[at];
			if (d1 <= 0) { ++count; continue; }
			if (count == 0) continue;
			if (count <= nIpolGapSize && (unsigned)u > count) {
				int uc = u - (int)count; const int uf = uc - 1;
				const size_t af = rows ? (size_t)o * w + uf : (size_t)uf * w + o;
				const float d0 = depth[af];
				if (similar(d0, d1, th)) {
					const float diff = (d1 - d0) / (float)(count + 1);
					float d = d0;
					const float c = pm_minf(conf[af], conf[at]);
					float dir1[2], dir2[2];
					Normal2Dir(normal + af * 3, dir1); Normal2Dir(normal + at * 3, dir2);
					const float dd[2] = {(dir2[0] - dir1[0]) / (float)(count + 1), (dir2[1] - dir1[1]) / (float)(count + 1)};
					do {
						const size_t ac = rows ? (size_t)o * w + uc : (size_t)uc * w + o;
						depth[ac] = (d += diff);
						dir1[0] += dd[0]; dir1[1] += dd[1];
						Dir2Normal(dir1, normal + ac * 3);
						conf[ac] = c;
					} while (++uc < u);
				}
			}
			count = 0;
		}
	}
}
}
extern "C" void orc_gap_interpolation(float* depth, float* normal, float* conf, int w, int h, unsigned nIpolGapSize, float fDepthDiffThreshold) {
	const float th = fDepthDiffThreshold * 2.5f;
	flt::gapPass(depth, normal, conf, w, h, true, nIpolGapSize, th);
	flt::gapPass(depth, normal, conf, w, h, false, nIpolGapSize, th);
}

// ---------------------------------------------------------------------------------------------
// DepthMapsData::RemoveSmallSegments, libs/MVS/SceneDensify.cpp:809-900, literal: region growing from seeds in
// column-major order along *directed* similarity edges (|d_cur - d_nb| / d_cur < 0.7 * fDepthDiffThreshold);
// segments smaller than nSpeckleSize are invalidated.
extern "C" void orc_remove_small_segments(float* depth, float* normal, float* conf, int w, int h, unsigned speckle_size, float fDepthDiffThreshold) {
	const float th = fDepthDiffThreshold * 0.7f;
	std::vector<unsigned char> done((size_t)w * h, 0);
	std::vector<int> seg; seg.reserve((size_t)w * h);
	for (int u = 0; u < w; ++u) for (int v = 0; v < h; ++v) {
		if (done[(size_t)v * w + u]) continue;
		seg.clear(); seg.push_back(v * w + u);
		size_t cur = 0;
		while (cur < seg.size()) {
			const int a = seg[cur]; const int ax = a % w, ay = a / w;
			const float dc = depth[a];
			if (dc > 0) {
				const int nx[4] = {ax - 1, ax + 1, ax, ax}, ny[4] = {ay, ay, ay - 1, ay + 1};
				for (int i = 0; i < 4; ++i) {
					if (!(nx[i] >= 0 && ny[i] >= 0 && nx[i] < w && ny[i] < h)) continue;
					const int b = ny[i] * w + nx[i];
					if (done[b]) continue;
					const float dn = depth[b];
					if (dn > 0 && flt::similar(dc, dn, th)) { seg.push_back(b); done[b] = 1; }
				}
			}
			++cur;
			done[a] = 1;
		}
		if (seg.size() < speckle_size)
			for (int a : seg) { depth[a] = 0; normal[a * 3] = normal[a * 3 + 1] = normal[a * 3 + 2] = 0; conf[a] = 0; }
	}
}
