// sgm_kernels.hip -- CDNA4 (gfx950) kernels for SemiGlobalMatcher::Match(ViewData,ViewData,...)
// (libs/MVS/SemiGlobalMatcher.cpp:863-1302): WZNCC cost volume, 8-path aggregation, winner-take-all.
//
// Layout is the reference's: a ragged cost volume, pixel p owns numDisp(p) consecutive entries at
// PixelData::idx (u8 costs, u16 path sums).  All three kernels map the 64 lanes of a wave onto the
// disparities of ONE pixel, so a wave's accesses to costs / sums are one contiguous 64..128-byte
// segment (coalesced), the cross-disparity minimum is a wave reduction, and the path recurrence
// L(d) <- Lp(d-1), Lp(d), Lp(d+1) goes through a 2-slot LDS line buffer.  Integer work is exact.
//
// The aggregation is a chain of dependent pixels along each path, so one wave owns one line and
// walks it in chunks of SGM_T pixels: the PixelData, cost bytes and running sums of a whole chunk
// are requested up front (independent loads, latency overlapped), then the chunk is consumed
// serially.  The O(D^2) inner loop of the reference (:1030-1044) is evaluated in its O(D) form
// (valid because P1 <= P2; see oracle/sgm_oracle.cpp: orc_sgm_step_forms_agree).
#pragma once
#include <hip/hip_runtime.h>
#include "pm_math.h"

struct SGMPixel { unsigned long long idx; short minDisp, maxDisp; int pad; }; // == SGMHipPixelData
#define SGM_HW 3
#define SGM_NT 49
#define SGM_T 8          // pixels per prefetch chunk
#define SGM_INF 0x3fffffff

__device__ __forceinline__ int sgm_round2int(float x) { return (int)pm_floorf(x + .5f); } // ROUND2INT, Types.h:949-955

// ---- cost volume, SemiGlobalMatcher.cpp:874-985: one wave per valid-grid pixel ---------------
__global__ __launch_bounds__(256) void sgm_cost_kernel(const unsigned char* __restrict__ colorL, const float* __restrict__ grayL,
		const float* __restrict__ grayR, int w, int h, int vw, int vh, const SGMPixel* __restrict__ pixels, unsigned char* __restrict__ costs) {
	__shared__ float2 s_w[4][SGM_NT + 1];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const long pix = (long)blockIdx.x * 4 + wave;
	const bool active = pix < (long)vw * vh;
	SGMPixel px; px.idx = 0; px.minDisp = 0; px.maxDisp = 0;
	if (active) px = pixels[pix];
	const bool valid = active && px.minDisp < px.maxDisp;
	const int ux = (int)(pix % vw) + SGM_HW, uy = (int)(pix / vw) + SGM_HW;
	const float sigmaColor = -1.f / (2.f * ((0.3f * 255) * (0.3f * 255)));
	const float sigmaSpatial = -1.f / (2.f * ((0.4f * 7) * (0.4f * 7)));
	if (valid && lane < SGM_NT) {
		const int i = lane / 7 - SGM_HW, j = lane % 7 - SGM_HW;
		const unsigned char* a = colorL + ((size_t)(uy + i) * w + (ux + j)) * 3;
		const unsigned char* c = colorL + ((size_t)uy * w + ux) * 3;
		unsigned s = 0;
#pragma unroll
		for (int k = 0; k < 3; ++k) { const unsigned d = a[k] < c[k] ? c[k] - a[k] : a[k] - c[k]; s += d * d; }
		const float wColor = (float)s * sigmaColor;
		const float wSpatial = (float)(j * j + i * i) * sigmaSpatial;
		s_w[wave][lane] = make_float2(pm_expf(wColor + wSpatial), grayL[(size_t)(uy + i) * w + (ux + j)]);
	}
	__syncthreads();
	float sumW = 0.f, normSq0 = 0.f, tm = 0.f;
	if (valid) {
		float acc = 0.f;
		for (int k = 0; k < SGM_NT; ++k) { const float2 p = s_w[wave][k]; acc += p.y * p.x; sumW += p.x; }
		tm = acc / sumW;
		for (int k = 0; k < SGM_NT; ++k) { const float2 p = s_w[wave][k]; const float t = p.y - tm; const float tw = p.x * t; normSq0 += tw * t; }
	}
	__syncthreads();
	if (valid && lane < SGM_NT) { const float2 p = s_w[wave][lane]; s_w[wave][lane] = make_float2(p.x, p.x * (p.y - tm)); }
	__syncthreads();
	if (!valid) return;
	const float eps = 1e-3f;
	for (int d = px.minDisp + lane; d < px.maxDisp; d += 64) {
		unsigned char cost;
		if (ux - SGM_HW + d < 0 || ux + SGM_HW + d >= w) cost = 255; // some tap outside the right image (:954-957)
		else {
			float sum = 0.f, sumSq = 0.f, nom = 0.f;
			int n = 0;
			for (int i = -SGM_HW; i <= SGM_HW; ++i) {
				const float* row = grayR + (size_t)(uy + i) * w + (ux + d);
#pragma unroll
				for (int j = -SGM_HW; j <= SGM_HW; ++j) {
					const float f = row[j];
					const float2 pw = s_w[wave][n++];
					const float fw = f * pw.x;
					sum += fw; sumSq += f * fw; nom += f * pw.y;
				}
			}
			const float normSq1 = sumSq - (sum * sum) / sumW;
			const float ncc = nom / pm_sqrtf(normSq0 * normSq1 + eps);
			cost = ncc <= 0 ? (unsigned char)255 : (unsigned char)sgm_round2int((1.f - pm_minf(ncc, 1.f)) * 255.f);
		}
		costs[px.idx + (unsigned)(d - px.minDisp)] = cost;
	}
}

// line start sets of one path direction: nA lines from (ax,ay) stepping (adx,ady), then the rest from (bx,by)
struct SGMLines { int nA, ax, ay, adx, ady, nB, bx, by, bdx, bdy; };

// ---- wave helpers ---------------------------------------------------------------------------
// wave64 minimum on the VALU cross-lane network (DPP row shifts + row broadcasts, then one readlane) instead of six
// dependent ds_bpermute round trips through the LDS crossbar: this reduction sits on the critical path of every step
// of the path recurrence.
__device__ __forceinline__ int sgm_wave_min(int v) {
	const int big = SGM_INF;
	v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false)); // row_shr:1
	v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false)); // row_shr:2
	v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false)); // row_shr:4
	v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false)); // row_shr:8  -> lane 15 of each row of 16 holds the row minimum
	v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1 and 3
	v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
	return __builtin_amdgcn_readlane(v, 63);
}

// accums(d) += L(d) (:1020,1043).  The eight path kernels run concurrently on eight streams, so the sum is an atomic add;
// two u16 sums share a 32-bit word (no carry between halves: a sum never exceeds 8*(255+60) = 2520), and the lane owning
// the even entry adds its right neighbour's value in the same atomic.  Must be called by all 64 lanes.
__device__ __forceinline__ void sgm_accumulate(unsigned* words, unsigned long long idx, int k, int nD, int L, int lane) {
	const int Lnext = __shfl_down(L, 1, 64);                       // value of entry k+1 (lane+1), garbage for lane 63
	const unsigned long long pos = idx + (unsigned)k;
	if (k < nD) {
		if ((pos & 1ull) == 0) {
			unsigned v = (unsigned)L;
			if (lane < 63 && k + 1 < nD) v |= (unsigned)Lnext << 16;
			atomicAdd(words + (pos >> 1), v);
		} else if (lane == 0 || k == 0) {
			atomicAdd(words + (pos >> 1), (unsigned)L << 16);       // odd entry whose even partner belongs to nobody in this wave-instruction
		}
	}
	// an odd entry at lane 63+1 of the previous 64-chunk is handled above by (lane == 0); the even entry of lane 63 adds alone
}

// ---- one path direction, SemiGlobalMatcher.cpp:1003-1046 + ACCUM_PIXELS :1065-1082 -----------
// One 64-thread workgroup (one wave) per line.  NK = ceil(maxNumDisp / 64) disparities per lane.
// Line start sets are the threaded variant's (:1083-1200); see the host for the numbering.
template <int NK>
__global__ __launch_bounds__(64) void sgm_path_kernel(const float* __restrict__ grayL, int w, int vw, int vh,
		const SGMPixel* __restrict__ pixels, const unsigned char* __restrict__ costs, unsigned* __restrict__ accumWords,
		const unsigned short* __restrict__ P2s, int P1, int dx, int dy, SGMLines ln, int maxNumDisp) {
	extern __shared__ __attribute__((aligned(16))) int s_L[]; // 2 x (maxNumDisp + 2): previous / current line of L
	const int lane = threadIdx.x;
	const int line = blockIdx.x;
	int x, y;
	if (line < ln.nA) { x = ln.ax + line * ln.adx; y = ln.ay + line * ln.ady; }
	else { const int i = line - ln.nA; x = ln.bx + i * ln.bdx; y = ln.by + i * ln.bdy; }
	const int stride = maxNumDisp + 2;
	int cur = 0;
	int rpMin = 0, rpMax = 0;
	float Ip = 0.5f;
	for (int k = lane; k < 2 * stride; k += 64) s_L[k] = 0;
	// the P2 table is indexed by a value computed inside the recurrence: keep it in LDS so the chain never waits on HBM
	__shared__ unsigned short s_P2[256];
	for (int k = lane; k < 256; k += 64) s_P2[k] = P2s[k];
	__syncthreads();
	// Software pipeline over chunks of SGM_T pixels: while chunk k is consumed (a serial recurrence), the cost bytes
	// and running sums of chunk k+1 and the pixel table of chunk k+2 are already in flight, so no step of the
	// recurrence waits on HBM.  Chunk state lives in registers (fully unrolled arrays).
	SGMPixel pxA[SGM_T], pxB[SGM_T]; float gA[SGM_T], gB[SGM_T]; bool okA[SGM_T], okB[SGM_T];
	unsigned char cA[SGM_T][NK], cB[SGM_T][NK];
	auto loadPixels = [&](int cx, int cy, SGMPixel* px, float* g, bool* ok) {
#pragma unroll
		for (int t = 0; t < SGM_T; ++t) {
			const int tx = cx + t * dx, ty = cy + t * dy;
			ok[t] = tx >= 0 && ty >= 0 && tx < vw && ty < vh;
			px[t].idx = 0; px[t].minDisp = 0; px[t].maxDisp = 0; g[t] = 0.f;
			if (ok[t]) {
				px[t] = pixels[(size_t)ty * vw + tx];
				g[t] = grayL[(size_t)ty * w + tx]; // imageGray(u) with the valid-grid coordinate: the reference's quirk (:1078)
			}
		}
	};
	auto loadCosts = [&](const SGMPixel* px, bool* ok, unsigned char (*c8)[NK]) {
#pragma unroll
		for (int t = 0; t < SGM_T; ++t) {
			const int nD = px[t].maxDisp - px[t].minDisp;
			ok[t] = ok[t] && nD > 0;
#pragma unroll
			for (int q = 0; q < NK; ++q) {
				const int k = lane + 64 * q;
				c8[t][q] = 0;
				if (ok[t] && k < nD) c8[t][q] = costs[px[t].idx + k];
			}
		}
	};
	loadPixels(x, y, pxA, gA, okA);
	loadCosts(pxA, okA, cA);
	loadPixels(x + SGM_T * dx, y + SGM_T * dy, pxB, gB, okB);
	while (x >= 0 && y >= 0 && x < vw && y < vh) {
		loadCosts(pxB, okB, cB);                                      // chunk k+1: cost bytes
		SGMPixel pxC[SGM_T]; float gC[SGM_T]; bool okC[SGM_T];
		loadPixels(x + 2 * SGM_T * dx, y + 2 * SGM_T * dy, pxC, gC, okC); // chunk k+2: pixel table
		// ---- consume chunk k serially ------------------------------------------------------------
#pragma unroll
		for (int t = 0; t < SGM_T; ++t) {
			if (!okA[t]) continue; // invalid pixels do not reset Lp / Ip (:1071-1072)
			const int rsMin = pxA[t].minDisp, rsMax = pxA[t].maxDisp, nD = rsMax - rsMin;
			const float DI = gA[t] - Ip;
			int ip = sgm_round2int(255.f * DI); ip = ip < 0 ? -ip : ip;
			const int P2 = s_P2[ip];
			const int lo = max(rpMin, rsMin), hi = min(rpMax, rsMax);
			const int* Lp = s_L + cur * stride + 1;       // Lp[d - rpMin]
			int* Ls = s_L + (cur ^ 1) * stride + 1;        // Ls[d - rsMin]
			if (lo >= hi) {
#pragma unroll
				for (int q = 0; q < NK; ++q) {
					const int k = lane + 64 * q;
					const int L = (int)cA[t][q] + P2;
					if (k < nD) Ls[k] = L;
					sgm_accumulate(accumWords, pxA[t].idx, k, nD, L, lane);
				}
			} else {
				int m = SGM_INF;
				for (int dp = lo + lane; dp < hi; dp += 64) m = min(m, Lp[dp - rpMin]);
				m = sgm_wave_min(m);
#pragma unroll
				for (int q = 0; q < NK; ++q) {
					const int k = lane + 64 * q;
					int L = 0;
					if (k < nD) {
						const int d = rsMin + k;
						int best = m + P2;
						if (d >= lo && d < hi) best = min(best, Lp[d - rpMin]);
						if (d - 1 >= lo && d - 1 < hi) best = min(best, Lp[d - 1 - rpMin] + P1);
						if (d + 1 >= lo && d + 1 < hi) best = min(best, Lp[d + 1 - rpMin] + P1);
						L = (int)cA[t][q] + best - m;
						Ls[k] = L;
					}
					sgm_accumulate(accumWords, pxA[t].idx, k, nD, L, lane);
				}
			}
			rpMin = rsMin; rpMax = rsMax; Ip = gA[t]; cur ^= 1;
			__syncthreads();
		}
		x += SGM_T * dx; y += SGM_T * dy;
		// rotate the pipeline registers
#pragma unroll
		for (int t = 0; t < SGM_T; ++t) {
			pxA[t] = pxB[t]; gA[t] = gB[t]; okA[t] = okB[t];
			pxB[t] = pxC[t]; gB[t] = gC[t]; okB[t] = okC[t];
#pragma unroll
			for (int q = 0; q < NK; ++q) cA[t][q] = cB[t][q];
		}
	}
}

// ---- winner-take-all, SemiGlobalMatcher.cpp:1272-1301: one wave per pixel --------------------
__global__ __launch_bounds__(256) void sgm_wta_kernel(const SGMPixel* __restrict__ pixels, const unsigned short* __restrict__ accums,
		long nPix, short* __restrict__ disp, unsigned short* __restrict__ cost) {
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const long pix = (long)blockIdx.x * 4 + wave;
	if (pix >= nPix) return;
	const SGMPixel px = pixels[pix];
	const int nD = px.maxDisp - px.minDisp;
	if (nD <= 0) { if (lane == 0) { disp[pix] = px.minDisp; cost[pix] = 0xFFFF; } return; }
	// first minimum == lexicographic minimum of (value, index)
	unsigned key = 0xFFFFFFFFu;
	for (int k = lane; k < nD; k += 64) key = min(key, ((unsigned)accums[px.idx + k] << 16) | (unsigned)k);
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, m, 64));
	if (lane == 0) { disp[pix] = (short)(px.minDisp + (int)(key & 0xFFFFu)); cost[pix] = (unsigned short)(key >> 16); }
}
