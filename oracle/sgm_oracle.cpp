// sgm_oracle.cpp -- CPU restatement of SemiGlobalMatcher::Match(ViewData, ViewData, ...)
// (libs/MVS/SemiGlobalMatcher.cpp:863-1302 in /root/reference): WZNCC cost volume, 8-path cost
// aggregation (the threaded variant's path set, :1083-1200) and winner-take-all.
//
// *** TEST INFRASTRUCTURE ONLY *** (see pm_oracle.cpp).  PINNED by the reference's own code since round 3: oracle/_ref/libref_sgm.so is
// SemiGlobalMatcher::Match cut verbatim from /root/reference (oracle/ref/ref_sgm_harness.cpp) and tests/test_ref_pinning.py compares disparity,
// cost, cost volume and accumulated sums bit for bit (uniform, ragged and holed ranges); known-answer tests in tests/test_sgm_oracle.py.  The aggregation
// recurrence is transcribed literally, O(D^2) inner loop included (:1030-1044); the GPU uses the
// O(D) form, so the tests also check the two forms agree.  exp() is pm_expf (shared with the GPU).
#include "../openmvs_amd/csrc/pm_math.h"
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <limits>
#include <vector>

namespace sgmo {

typedef int16_t Disparity; typedef uint8_t Cost; typedef uint16_t AccumCost; typedef uint64_t Index;
struct Range { Disparity minDisp, maxDisp; int numDisp() const { return maxDisp - minDisp; } bool isValid() const { return minDisp < maxDisp; } };
struct PixelData { Index idx; Range range; };   // SemiGlobalMatcher.h:79-82 (16 bytes)
enum { HWX = 3, HWY = 3, NT = 49 };

static inline int Round2Int(float x) { return (int)floorf(x + .5f); } // ROUND2INT without _USE_FAST_FLOAT2INT, Types.h:949-955

struct LineData { std::vector<AccumCost> L; Range R; };

// pixelAccum lambda, SemiGlobalMatcher.cpp:1003-1046 (literal)
static void pixelAccum(const Cost* costs, const LineData& Lp, LineData& Ls, AccumCost* accums, float DI, AccumCost P1, const AccumCost* P2s) {
	const AccumCost P2 = P2s[std::abs(Round2Int(255.f * DI))];
	const Disparity minDisp = std::max(Lp.R.minDisp, Ls.R.minDisp);
	const Disparity maxDisp = std::min(Lp.R.maxDisp, Ls.R.maxDisp);
	if (minDisp >= maxDisp) {
		const int numDisp = Ls.R.numDisp();
		for (int i = 0; i < numDisp; ++i)
			accums[i] += (Ls.L[i] = (AccumCost)(costs[i] + P2));
	} else {
		AccumCost minLp = std::numeric_limits<AccumCost>::max();
		for (int dp = minDisp; dp < maxDisp; ++dp) minLp = std::min(minLp, Lp.L[dp - Lp.R.minDisp]);
		for (int d = Ls.R.minDisp; d < Ls.R.maxDisp; ++d) {
			const int idxDisp = d - Ls.R.minDisp;
			AccumCost L = std::numeric_limits<AccumCost>::max();
			for (int dp = minDisp; dp < maxDisp; ++dp) {
				const int idxDispp = dp - Lp.R.minDisp;
				AccumCost v;
				if (dp == d) v = Lp.L[idxDispp];
				else if (dp == d - 1 || dp == d + 1) v = (AccumCost)(Lp.L[idxDispp] + P1);
				else v = (AccumCost)(Lp.L[idxDispp] + P2);
				if (L > v) L = v;
			}
			L = (AccumCost)(costs[idxDisp] + L - minLp);
			Ls.L[idxDisp] = L;
			accums[idxDisp] += L;
		}
	}
}

struct Ctx {
	int w, h, vw, vh; // image size and valid size (w-6, h-6)
	const uint8_t* colorL; const float* grayL; const float* grayR;
	const PixelData* pixels; Cost* costs; AccumCost* accums; int maxNumDisp;
	AccumCost P1; const AccumCost* P2s;
};

// cost volume, SemiGlobalMatcher.cpp:874-985
static void computeCosts(const Ctx& c) {
	const float eps = 1e-3f;
	const float sigmaColor = -1.f / (2.f * ((0.3f * 255) * (0.3f * 255)));
	const float sigmaSpatial = -1.f / (2.f * ((0.4f * 7) * (0.4f * 7)));
	for (int r = 0; r < c.vh; ++r) for (int col = 0; col < c.vw; ++col) {
		const PixelData& px = c.pixels[(size_t)r * c.vw + col];
		if (!px.range.isValid()) continue;
		const int ux = col + HWX, uy = r + HWY;
		float weight[NT], tempWeight[NT];
		float normSq0 = 0, sumWeights = 0;
		int n = 0;
		const uint8_t* cc = c.colorL + ((size_t)uy * c.w + ux) * 3;
		for (int i = -HWY; i <= HWY; ++i) for (int j = -HWX; j <= HWX; ++j) {
			const int x = ux + j, y = uy + i;
			const uint8_t* a = c.colorL + ((size_t)y * c.w + x) * 3;
			unsigned s = 0;
			for (int k = 0; k < 3; ++k) { const unsigned d = a[k] < cc[k] ? cc[k] - a[k] : a[k] - cc[k]; s += d * d; }
			const float wColor = (float)s * sigmaColor;
			const float wSpatial = (float)(j * j + i * i) * sigmaSpatial;
			tempWeight[n] = c.grayL[(size_t)y * c.w + x];
			weight[n] = pm_expf(wColor + wSpatial);
			normSq0 += tempWeight[n] * weight[n];
			sumWeights += weight[n];
			++n;
		}
		const float tm = normSq0 / sumWeights;
		normSq0 = 0;
		for (n = 0; n < NT; ++n) { const float t = tempWeight[n] - tm; tempWeight[n] = weight[n] * t; normSq0 += tempWeight[n] * t; }
		Cost* costs = c.costs + px.idx;
		for (int d = px.range.minDisp; d < px.range.maxDisp; ++d) {
			float sum = 0, sumSq = 0, nom = 0;
			bool outside = false;
			n = 0;
			for (int i = -HWY; i <= HWY && !outside; ++i) for (int j = -HWX; j <= HWX; ++j) {
				const int x = ux + j + d, y = uy + i;
				if (!(x >= 0 && y >= 0 && x < c.w && y < c.h)) { outside = true; break; }
				const float f = c.grayR[(size_t)y * c.w + x];
				const float fw = f * weight[n];
				sum += fw; sumSq += f * fw; nom += f * tempWeight[n];
				++n;
			}
			if (outside) { *costs++ = 255; continue; }
			const float normSq1 = sumSq - (sum * sum) / sumWeights;
			const float ncc = nom / pm_sqrtf(normSq0 * normSq1 + eps);
			*costs++ = (ncc <= 0 ? (Cost)255 : (Cost)Round2Int((1.f - pm_minf(ncc, 1.f)) * 255.f));
		}
	}
}

// one directional line sweep: ACCUM_PIXELS, SemiGlobalMatcher.cpp:1065-1082
static void sweepLine(const Ctx& c, int x, int y, int dx, int dy) {
	LineData lines[2];
	for (auto& l : lines) { l.L.assign(c.maxNumDisp, 0); l.R.minDisp = l.R.maxDisp = 0; }
	int cur = 0;
	float Ip = 0.5f;
	for (; x >= 0 && y >= 0 && x < c.vw && y < c.vh; x += dx, y += dy) {
		const PixelData& px = c.pixels[(size_t)y * c.vw + x];
		if (!px.range.isValid()) continue;      // invalid pixels do not reset Lp / Ip (:1071-1072)
		const LineData& Lp = lines[cur]; LineData& Ls = lines[cur ^ 1];
		Ls.R = px.range;
		const float I = c.grayL[(size_t)y * c.w + x]; // imageGray(u) with the *valid-grid* coordinate (:1078) -- replicated quirk
		pixelAccum(c.costs + px.idx, Lp, Ls, c.accums + px.idx, I - Ip, c.P1, c.P2s);
		Ip = I;
		cur ^= 1;
	}
}

// 8 path directions with the threaded variant's line start sets, SemiGlobalMatcher.cpp:1083-1200
static void aggregate(const Ctx& c) {
	const int W = c.vw, H = c.vh;
	for (int x = 0; x < W; ++x) sweepLine(c, x, 0, 0, 1);           // width-down
	for (int y = 0; y < H; ++y) sweepLine(c, 0, y, 1, 0);           // height-right
	for (int x = 0; x < W; ++x) sweepLine(c, x, H - 1, 0, -1);      // width-up
	for (int y = 0; y < H; ++y) sweepLine(c, W - 1, y, -1, 0);      // height-left
	for (int x = 0; x < W; ++x) sweepLine(c, x, 0, 1, 1);           // right-down: starts on the top row ...
	for (int y = 1; y < H; ++y) sweepLine(c, 0, y, 1, 1);           // ... and on the left column (y >= 1)
	for (int x = 0; x < W - 1; ++x) sweepLine(c, x, 0, -1, 1);      // left-down: top row x < W-1 ...
	for (int y = 0; y < H; ++y) sweepLine(c, W - 1, y, -1, 1);      // ... and right column
	for (int x = 1; x < W; ++x) sweepLine(c, x, H - 1, 1, -1);      // right-up: bottom row x >= 1 ...
	for (int y = H - 1; y >= 0; --y) sweepLine(c, 0, y, 1, -1);     // ... and left column
	for (int x = W - 1; x >= 0; --x) sweepLine(c, x, H - 1, -1, -1);// left-up: bottom row ...
	for (int y = H - 2; y >= 0; --y) sweepLine(c, W - 1, y, -1, -1);// ... and right column (y <= H-2)
}

// winner-take-all, SemiGlobalMatcher.cpp:1272-1301
static void wta(const Ctx& c, Disparity* disp, AccumCost* cost) {
	for (size_t i = 0; i < (size_t)c.vw * c.vh; ++i) {
		const PixelData& px = c.pixels[i];
		if (px.range.isValid()) {
			const AccumCost* a = c.accums + px.idx;
			int best = 0;
			for (int k = 1; k < px.range.numDisp(); ++k) if (a[best] > a[k]) best = k;
			disp[i] = (Disparity)(px.range.minDisp + best); cost[i] = a[best];
		} else { disp[i] = px.range.minDisp; cost[i] = 0xFFFF; }
	}
}

} // namespace sgmo

extern "C" {

// GenerateP2s, SemiGlobalMatcher.cpp:518-524 (defaults P2=4, alpha=14, beta=38; exp = pm_expf)
void orc_sgm_generate_p2s(uint16_t P2, float alpha, float beta, uint16_t* out256) {
	for (int i = 0; i < 256; ++i) {
		const float fi = (float)i;
		out256[i] = (uint16_t)sgmo::Round2Int((float)P2 * (1.f + alpha * pm_expf(-(fi * fi) / (2.f * (beta * beta)))));
	}
}

// pixels: (w-6)*(h-6) entries {u64 idx; i16 min,max; pad}; costs/accums: numCosts entries (outputs)
int orc_sgm_match(const uint8_t* colorL, const float* grayL, const float* grayR, int w, int h,
		const void* pixels, uint64_t numCosts, int maxNumDisp, uint16_t P1, const uint16_t* P2s,
		int16_t* disparity, uint16_t* cost, uint8_t* costsOut, uint16_t* accumsOut) {
	sgmo::Ctx c;
	c.w = w; c.h = h; c.vw = w - 6; c.vh = h - 6;
	c.colorL = colorL; c.grayL = grayL; c.grayR = grayR;
	c.pixels = (const sgmo::PixelData*)pixels; c.maxNumDisp = maxNumDisp; c.P1 = P1; c.P2s = P2s;
	std::vector<uint8_t> costs(numCosts, 0); std::vector<uint16_t> accums(numCosts, 0);
	c.costs = costs.data(); c.accums = accums.data();
	sgmo::computeCosts(c);
	sgmo::aggregate(c);
	sgmo::wta(c, disparity, cost);
	if (costsOut) memcpy(costsOut, costs.data(), numCosts);
	if (accumsOut) memcpy(accumsOut, accums.data(), numCosts * 2);
	return 0;
}

// brute-force check hook: literal O(D^2) recurrence vs the O(D) form used on the GPU, one step
int orc_sgm_step_forms_agree(const uint16_t* Lp, int pmin, int pmax, const uint8_t* costs, int smin, int smax, uint16_t P1, uint16_t P2) {
	sgmo::LineData lp, ls; lp.R.minDisp = (int16_t)pmin; lp.R.maxDisp = (int16_t)pmax; lp.L.assign(Lp, Lp + std::max(0, pmax - pmin));
	ls.R.minDisp = (int16_t)smin; ls.R.maxDisp = (int16_t)smax; ls.L.assign(smax - smin, 0);
	std::vector<uint16_t> acc(smax - smin, 0);
	uint16_t P2s[256]; for (auto& v : P2s) v = P2;
	sgmo::pixelAccum(costs, lp, ls, acc.data(), 0.f, P1, P2s);
	const int lo = std::max(pmin, smin), hi = std::min(pmax, smax);
	for (int d = smin; d < smax; ++d) {
		int L;
		if (lo >= hi) L = costs[d - smin] + P2;
		else {
			int m = 0xFFFF; for (int dp = lo; dp < hi; ++dp) m = std::min<int>(m, Lp[dp - pmin]);
			int best = m + P2;
			if (d >= lo && d < hi) best = std::min<int>(best, Lp[d - pmin]);
			if (d - 1 >= lo && d - 1 < hi) best = std::min<int>(best, Lp[d - 1 - pmin] + P1);
			if (d + 1 >= lo && d + 1 < hi) best = std::min<int>(best, Lp[d + 1 - pmin] + P1);
			L = costs[d - smin] + best - m;
		}
		if ((uint16_t)L != ls.L[d - smin]) return 0;
	}
	return 1;
}

} // extern "C"
