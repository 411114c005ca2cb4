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
// a pointer rebuilt from an integer is "generic" to the compiler (flat_load, two counters); these say it is HBM (global_load, scalar base + lane offset)
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
typedef const unsigned char __attribute__((address_space(1)))* sgm_gcb;
typedef const unsigned short __attribute__((address_space(1)))* sgm_gcs;
#else
typedef const unsigned char* sgm_gcb;
typedef const unsigned short* sgm_gcs;
#endif
#define SGM_HW 3
#define SGM_NT 49
#ifndef SGM_T
#define SGM_T 8          // pixels per prefetch chunk
#endif
#define SGM_INF 0x3fffffff

__device__ __forceinline__ int sgm_round2int(float x) { return (int)pm_floorf(x + .5f); } // ROUND2INT, Types.h:949-955

// ---- cost volume, SemiGlobalMatcher.cpp:874-985 --------------------------------------------------------------------------
// The kernel is fp32-VALU bound (49 taps x 3 running sums per cost, all in the reference's summation order), so the design goal
// is wave-instructions per cost:
//  * the per-pixel prologue (weighted mean and variance of the left window, :905-935) is two serial 49-term sums; done by the
//    wave that owns the pixel it costs 196 wave-instructions per pixel, done one pixel per LANE (sgm_setup_kernel) it costs 1/64
//    of that.  Its three results per pixel travel through a 16-byte record;
//  * a wave computes the costs of TWO horizontally adjacent pixels, one disparity of each per lane, as float2 lanes: the three
//    multiply-add pairs of a tap become v_pk_mul_f32 / v_pk_add_f32 (IEEE, unfused: -ffp-contract=off), and one ds_read_b128
//    delivers the weights of both pixels.
typedef float sgm_v2f __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define SGM_SCHED_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)   // no memory access is moved and nothing is scheduled across this point
#define SGM_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#else
#define SGM_SCHED_BARRIER() do {} while (0)
#define SGM_PIN3(a, b, c) do {} while (0)
#endif

__device__ __forceinline__ float sgm_weight(const unsigned char* __restrict__ colorL, int w, int ux, int uy, int i, int j) {
	const float sigmaColor = -1.f / (2.f * ((0.3f * 255) * (0.3f * 255)));
	const float sigmaSpatial = -1.f / (2.f * ((0.4f * 7) * (0.4f * 7)));
	const unsigned char* a = colorL + ((size_t)(uy + i) * w + (ux + j)) * 3;
	const unsigned char* c = colorL + ((size_t)uy * w + ux) * 3;
	unsigned s = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) { const unsigned d = a[k] < c[k] ? c[k] - a[k] : a[k] - c[k]; s += d * d; }
	const float wColor = (float)s * sigmaColor;
	const float wSpatial = (float)(j * j + i * i) * sigmaSpatial;
	return pm_expf(wColor + wSpatial);
}

// one valid-grid pixel per lane: {sumW, mean, normSq0, 0}
__global__ __launch_bounds__(256) void sgm_setup_kernel(const unsigned char* __restrict__ colorL, const float* __restrict__ grayL,
		int w, int vw, int vh, const SGMPixel* __restrict__ pixels, float4* __restrict__ setup) {
	const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (pix >= (long)vw * vh) return;
	const SGMPixel px = pixels[pix];
	if (!(px.minDisp < px.maxDisp)) { setup[pix] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
	const int ux = (int)(pix % vw) + SGM_HW, uy = (int)(pix / vw) + SGM_HW;
	float wk[SGM_NT], vk[SGM_NT];
	float acc = 0.f, sumW = 0.f;
#pragma unroll
	for (int k = 0; k < SGM_NT; ++k) {
		const int i = k / 7 - SGM_HW, j = k % 7 - SGM_HW;
		wk[k] = sgm_weight(colorL, w, ux, uy, i, j);
		vk[k] = grayL[(size_t)(uy + i) * w + (ux + j)];
		acc += vk[k] * wk[k]; sumW += wk[k];
	}
	const float tm = acc / sumW;
	float normSq0 = 0.f;
#pragma unroll
	for (int k = 0; k < SGM_NT; ++k) { const float t = vk[k] - tm; const float tw = wk[k] * t; normSq0 += tw * t; }
	setup[pix] = make_float4(sumW, tm, normSq0, 0.f);
}

__device__ __forceinline__ unsigned char sgm_cost_of(float sum, float sumSq, float nom, float sumW, float normSq0) {
	const float eps = 1e-3f;
	const float normSq1 = sumSq - (sum * sum) / sumW;
	const float ncc = nom / pm_sqrtf(normSq0 * normSq1 + eps);
	return ncc <= 0 ? (unsigned char)255 : (unsigned char)sgm_round2int((1.f - pm_minf(ncc, 1.f)) * 255.f);
}

// one wave per pair of horizontally adjacent valid-grid pixels (A, B = A + 1)
__global__ __launch_bounds__(256, 4) void sgm_cost_kernel(const unsigned char* __restrict__ colorL, const float* __restrict__ grayL,
		const float* __restrict__ grayR, int w, int h, int vw, int vh, const SGMPixel* __restrict__ pixels,
		const float4* __restrict__ setup, unsigned char* __restrict__ costs) {
	__shared__ float4 s_w[4][SGM_NT + 1];                              // (wA, wB, wA*(vA-meanA), wB*(vB-meanB)) per tap
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int ppr = (vw + 1) >> 1;                                     // pairs per row
	const long pair = (long)blockIdx.x * 4 + wave;
	if (pair >= (long)ppr * vh) return;                                // (no workgroup barrier below: each wave owns its LDS rows)
	const int row = (int)(pair / ppr), colA = (int)(pair % ppr) * 2;
	const long pixA = (long)row * vw + colA;
	const bool hasB = colA + 1 < vw;
	const SGMPixel pxA = pixels[pixA];
	SGMPixel pxB; pxB.idx = 0; pxB.minDisp = 0; pxB.maxDisp = 0; pxB.pad = 0;
	if (hasB) pxB = pixels[pixA + 1];
	const int nDA = pxA.maxDisp > pxA.minDisp ? pxA.maxDisp - pxA.minDisp : 0;
	const int nDB = pxB.maxDisp > pxB.minDisp ? pxB.maxDisp - pxB.minDisp : 0;
	if (nDA == 0 && nDB == 0) return;
	const int ux = colA + SGM_HW, uy = row + SGM_HW;                   // pixel A in image coordinates; B is at ux + 1
	const float4 sA = setup[pixA];
	const float4 sB = hasB ? setup[pixA + 1] : make_float4(1.f, 0.f, 0.f, 0.f);
	if (lane < SGM_NT) {
		const int i = lane / 7 - SGM_HW, j = lane % 7 - SGM_HW;
		float wA = 0.f, wB = 0.f, tA = 0.f, tB = 0.f;
		if (nDA) { wA = sgm_weight(colorL, w, ux, uy, i, j); tA = wA * (grayL[(size_t)(uy + i) * w + (ux + j)] - sA.y); }
		if (nDB) { wB = sgm_weight(colorL, w, ux + 1, uy, i, j); tB = wB * (grayL[(size_t)(uy + i) * w + (ux + 1 + j)] - sB.y); }
		s_w[wave][lane] = make_float4(wA, wB, tA, tB);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const int nDmax = nDA > nDB ? nDA : nDB;
#pragma unroll 1
	for (int k = lane; k < nDmax; k += 64) {
		asm volatile("" ::: "memory");   // keep the 49 weight reads inside the iteration (hoisted, they occupy 196 VGPRs and spill)
		const int dA = pxA.minDisp + k, dB = pxB.minDisp + k;
		const bool actA = k < nDA, actB = k < nDB;
		const bool inA = actA && !(ux - SGM_HW + dA < 0 || ux + SGM_HW + dA >= w);       // all taps inside the right image (:954-957)
		const bool inB = actB && !(ux + 1 - SGM_HW + dB < 0 || ux + 1 + SGM_HW + dB >= w);
		const int cA = inA ? ux + dA : SGM_HW, cB = inB ? ux + 1 + dB : SGM_HW;           // a safe column for lanes that will not use the result
		sgm_v2f sum = {0.f, 0.f}, sumSq = {0.f, 0.f}, nom = {0.f, 0.f};
		int n = 0;
		for (int i = -SGM_HW; i <= SGM_HW; ++i) {
			const float* rowA = grayR + (size_t)(uy + i) * w + cA;
			const float* rowB = grayR + (size_t)(uy + i) * w + cB;
#pragma unroll
			for (int j = -SGM_HW; j <= SGM_HW; ++j) {
#ifdef SGM_PROBE_NO_LOADS      /* timing probes (never in the product build; results are not the reference's): what bounds the cost kernel? */
				const sgm_v2f f = {0.3f + 0.001f * (float)(j + k), 0.4f + 0.001f * (float)(i + k)};
#else
				const sgm_v2f f = {rowA[j], rowB[j]};
#endif
#ifdef SGM_PROBE_NO_WEIGHT_READS
				const float4 pw = make_float4(0.02f, 0.021f, 0.003f * (float)n, 0.002f * (float)n); ++n;
#else
				const float4 pw = s_w[wave][n++];
#endif
				const sgm_v2f pww = {pw.x, pw.y}, pwt = {pw.z, pw.w};
				const sgm_v2f fw = f * pww;
				sum += fw; sumSq += f * fw; nom += f * pwt;
			}
		}
		if (actA) costs[pxA.idx + (unsigned)k] = inA ? sgm_cost_of(sum.x, sumSq.x, nom.x, sA.x, sA.z) : (unsigned char)255;
		if (actB) costs[pxB.idx + (unsigned)k] = inB ? sgm_cost_of(sum.y, sumSq.y, nom.y, sB.x, sB.z) : (unsigned char)255;
	}
}

// ---- the same cost volume with one valid-grid pixel per LANE ---------------------------------------------------------------------------------------
// sgm_cost_kernel spends, per 64 costs, 49 broadcast ds_read_b128 (the pixel's weights are wave-uniform there), 28 unaligned multi-dword loads
// and 294 packed VALU instructions, and the three units overlap badly (probes: r02_sgm_cost_probes.log; packed fp32 multiplies and adds move no more
// flops per cycle than plain ones on this part: r03_valu_rate.log).  Here a lane owns a pixel and walks its disparities: the 49 weights w and the
// 49 products w*(v-mean) live in the lane's registers for the whole walk (or the products in a lane-private LDS column, SGM_PX_T_IN_LDS),
// the 7x7 window of the right image slides with the disparity (7 new texels per cost; the loads of neighbouring lanes are neighbouring addresses),
// and what remains per cost is the arithmetic the reference's summation order demands: 49 x (2 mul + 1 add, 1 mul + 1 add, 1 mul + 1 add), all
// full-rate VOP2.  The prologue of sgm_setup_kernel (weighted mean / variance of the left window) is the same 49 weights and is done here too --
// no setup pass, no 16-byte record.  A lane collects four costs in a register and stores them as one aligned dword of its pixel's run.
#ifndef SGM_PX_T_IN_LDS
#define SGM_PX_T_IN_LDS 0   // 1: the 49 products w*(v-mean) in a lane-private LDS column (167 VGPRs, 3 waves per SIMD) instead of registers (221 VGPRs, 2 waves).
#endif                      // Measured at 2048x1536 (profiles/r03_sgm_call12.log, r03_sgm_call11_px_cost_u64.log): D = 64: 2.24 vs 2.22 ms, D = 128: 4.32 vs 3.71 ms -- the
                            // just-in-time LDS reads put their latency on the chain of each running sum; registers win.
#if SGM_PX_T_IN_LDS
#define SGM_PX_T(n) s_t[wave][n][lane]
#define SGM_PX_WAVES 3
#else
#define SGM_PX_T(n) tk[n]
#define SGM_PX_WAVES 2
#endif
#define SGM_PX_STEP(R_)                                                                                                                          \
	{                                                                                                                                             \
		const int k = kb + (R_);                                                                                                                  \
		if (k >= nDmax) break;                                                                                                                    \
		if (SGM_PX_T_IN_LDS) asm volatile("" ::: "memory");   /* keep the 49 LDS reads inside the step: hoisted out of the loop they are 49 registers again */ \
		const int dn = d0 + k + 1 + SGM_HW;                      /* column of the texels entering the window for disparity k+1 */              \
		const unsigned cn = (unsigned)(ux + dn < 0 ? 0 : (ux + dn >= w ? w - 1 : ux + dn));                                                       \
		float sum = 0.f, sumSq = 0.f, nom = 0.f;                                                                                                  \
		_Pragma("unroll") for (int n = 0; n < SGM_NT; ++n) {                                                                                      \
			const float f = win[n / 7][((R_) + n % 7) % 7];                                                                                       \
			/* the leftmost column leaves the window with this step: its slot takes the texel of the entering column as soon as it has been read */ \
			if (n % 7 == 0) win[n / 7][(R_)] = *(const float*)(rowsRb + (((unsigned)(n / 7) * (unsigned)w + cn) << 2));                           \
			const float fw = f * wk[n];                                                                                                           \
			sum += fw; sumSq += f * fw; nom += f * SGM_PX_T(n);                                                                                   \
		}                                                                                                                                         \
		const int d = d0 + k;                                                                                                                     \
		const bool in = !(ux - SGM_HW + d < 0 || ux + SGM_HW + d >= w);                                                                           \
		const unsigned c = in ? (unsigned)sgm_cost_of(sum, sumSq, nom, sumW, normSq0) : 255u;                                                    \
		if (k < nD) {                                                                                                                             \
			const unsigned pos = (idxLow + (unsigned)k) & 3u;   /* byte of its dword in the volume */                                             \
			packed |= c << (8u * pos);                                                                                                            \
			if (pos == 3u || k == nD - 1) {                                                                                                       \
				unsigned char* at = costs + px.idx + (unsigned)k;   /* address of this (the last collected) byte */                               \
				const unsigned first = (unsigned)k < pos ? pos - (unsigned)k : 0u;   /* first byte of the dword that belongs to this pixel */      \
				if (pos == 3u && first == 0u) *reinterpret_cast<unsigned*>(at - 3) = packed;                                                      \
				else for (unsigned bb = first; bb <= pos; ++bb) at[(int)bb - (int)pos] = (unsigned char)(packed >> (8u * bb));                    \
				packed = 0u;                                                                                                                      \
			}                                                                                                                                     \
		}                                                                                                                                         \
	}

__global__ __launch_bounds__(256, SGM_PX_WAVES) void sgm_cost_px_kernel(const unsigned char* __restrict__ colorL, const float* __restrict__ grayL,
		const float* __restrict__ grayR, int w, int h, int vw, int vh, const SGMPixel* __restrict__ pixels, unsigned char* __restrict__ costs) {
#if SGM_PX_T_IN_LDS
	__shared__ float s_t[4][SGM_NT][64];                               // [wave][tap][lane]: w * (v - mean) of the lane's pixel (first: v itself)
#else
	float tk[SGM_NT];
#endif
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tpr = (vw + 63) >> 6;                                     // 64-pixel tiles per row
	const long tile = (long)blockIdx.x * 4 + wave;
	if (tile >= (long)tpr * vh) return;                                 // (no workgroup barrier below: each wave owns its LDS slice)
	const int row = (int)(tile / tpr), col = (int)(tile % tpr) * 64 + lane;
	const bool have = col < vw;
	const long pix = (long)row * vw + (have ? col : vw - 1);
	const SGMPixel px = pixels[pix];
	const int nD = have && px.maxDisp > px.minDisp ? px.maxDisp - px.minDisp : 0;
	int nDmax = nD;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) nDmax = max(nDmax, __shfl_xor(nDmax, o, 64));
	if (nDmax == 0) return;
	const unsigned idxLow = (unsigned)(px.idx & 3ull);
	const int ux = (have ? col : vw - 1) + SGM_HW, uy = row + SGM_HW;
	// left window: weights, weighted mean, t = w * (v - mean), normSq0 (:905-935, the sums in tap order)
	float wk[SGM_NT];
	float sumW = 0.f, normSq0 = 0.f;
	{
		float acc = 0.f;
#pragma unroll
		for (int n = 0; n < SGM_NT; ++n) {
			const int i = n / 7 - SGM_HW, j = n % 7 - SGM_HW;
			wk[n] = sgm_weight(colorL, w, ux, uy, i, j);
			const float v = grayL[(size_t)(uy + i) * w + (ux + j)];
			SGM_PX_T(n) = v;
			acc += v * wk[n]; sumW += wk[n];
		}
		const float tm = acc / sumW;
#pragma unroll
		for (int n = 0; n < SGM_NT; ++n) { const float t = SGM_PX_T(n) - tm; const float tw = wk[n] * t; normSq0 += tw * t; SGM_PX_T(n) = tw; }
	}
	// right window for the first disparity: columns ux+d0-3 .. ux+d0+3 (clamped into the image; a clamped column only feeds costs that are 255 anyway)
	const int d0 = px.minDisp;
	const char* rowsRb = (const char*)(grayR + (size_t)(uy - SGM_HW) * w);   // (wave-uniform)
	float win[7][7];                                                    // win[i][s]: row uy-3+i, slot s = (column offset + 3 + k) mod 7
#pragma unroll
	for (int s7 = 0; s7 < 7; ++s7) {
		const int c0 = ux + d0 - SGM_HW + s7;
		const int cc = c0 < 0 ? 0 : (c0 >= w ? w - 1 : c0);
#pragma unroll
		for (int i = 0; i < 7; ++i) win[i][s7] = *(const float*)(rowsRb + (((unsigned)i * (unsigned)w + (unsigned)cc) << 2));
	}
	unsigned packed = 0u;
#pragma unroll 1
	for (int kb = 0; kb < nDmax; kb += 7) {
		SGM_PX_STEP(0) SGM_PX_STEP(1) SGM_PX_STEP(2) SGM_PX_STEP(3) SGM_PX_STEP(4) SGM_PX_STEP(5) SGM_PX_STEP(6)
	}
}
#undef SGM_PX_STEP
#undef SGM_PX_T

// ---- the lane-per-pixel cost kernel when every pixel has the same range (a plain Match, the first tSGM level) -------------------------------------------
// Lane l of a wave owns pixel x0 + l of a row and all lanes are at the same disparity, so tap (i, j) of lane l at disparity index k is column
// l + k + j + 3 of a strip of 64 + nD + 6 right-image columns that the whole wave shares: the strip (7 rows) is loaded into LDS once, and the sliding window
// of sgm_cost_px_kernel -- 49 registers per lane and 7 global loads per cost -- becomes 49 conflict-free ds_read_b32 per cost (consecutive lanes, consecutive
// addresses; ~20 % of the LDS rate next to 300 VALU instructions).  ~125 VGPRs: 3 waves per SIMD instead of 2, and no vector memory traffic in the walk but
// the packed cost stores.  Same arithmetic, same order.
#ifndef SGM_UNI_WAVES
#define SGM_UNI_WAVES 3
#endif
#ifndef SGM_UNI_PAIR
#define SGM_UNI_PAIR 1     // two disparities per trip of the cost loop (0: one)
#endif
template <int MD>   // nD <= MD: sizes the strip
__global__ __launch_bounds__(256, SGM_UNI_WAVES) void sgm_cost_uni_kernel(const unsigned char* __restrict__ colorL, const float* __restrict__ grayL,
		const float* __restrict__ grayR, int w, int h, int vw, int vh, int minDisp, int nDall, unsigned char* __restrict__ costs) {
	// column-major with 9 floats per column: every tap of a lane is within 60 dwords of one address register (ds_read2_b32 immediates; row-major needs a
	// register per row), and consecutive lanes are 9 dwords apart -- coprime with the 64 banks, so a read is conflict-free
	constexpr int NCOL = 64 + MD + 8, CS = 9;
	__shared__ float s_r[4][NCOL * CS];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tpr = (vw + 63) >> 6;                                     // 64-pixel tiles per row
	const long tile = (long)blockIdx.x * 4 + wave;
	if (tile >= (long)tpr * vh) return;                                 // (no workgroup barrier below: each wave owns its LDS slice)
	const int row = (int)(tile / tpr), col0 = (int)(tile % tpr) * 64, col = col0 + lane;
	const bool have = col < vw;
	const int nD = have ? nDall : 0;
	const unsigned long long idx = (unsigned long long)((long)row * vw + (have ? col : vw - 1)) * (unsigned long long)nDall;   // PixelData::idx of a uniform table
	const unsigned idxLow = (unsigned)(idx & 3ull);
	const int ux = (have ? col : vw - 1) + SGM_HW, uy = row + SGM_HW;
	// the strip: columns cb .. cb + 64 + nD + 5 of rows uy-3 .. uy+3 (clamped into the image; a clamped column only feeds costs that are 255 anyway)
	const int cb = col0 + SGM_HW + minDisp - SGM_HW;
	for (int c = lane; c < 64 + nDall + 7; c += 64) {                   // (+1: the pair loop reads one column past an odd range; that cost is never stored)
		const int cc = cb + c < 0 ? 0 : (cb + c >= w ? w - 1 : cb + c);
#pragma unroll
		for (int i = 0; i < 7; ++i) s_r[wave][c * CS + i] = grayR[(size_t)(uy - SGM_HW + i) * w + cc];
	}
	// left window: weights, weighted mean, t = w * (v - mean), normSq0 (:905-935, the sums in tap order)
	float wk[SGM_NT], tk[SGM_NT];
	float sumW = 0.f, normSq0 = 0.f;
	{
		float acc = 0.f;
#pragma unroll
		for (int n = 0; n < SGM_NT; ++n) {
			const int i = n / 7 - SGM_HW, j = n % 7 - SGM_HW;
			wk[n] = sgm_weight(colorL, w, ux, uy, i, j);
			tk[n] = grayL[(size_t)(uy + i) * w + (ux + j)];
			acc += tk[n] * wk[n]; sumW += wk[n];
		}
		const float tm = acc / sumW;
#pragma unroll
		for (int n = 0; n < SGM_NT; ++n) { const float t = tk[n] - tm; const float tw = wk[n] * t; normSq0 += tw * t; tk[n] = tw; }
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	unsigned packed = 0u;
	const float* strip = &s_r[wave][(have ? lane : vw - 1 - col0) * CS];   // a lane without a pixel repeats the row's last one
	// one cost byte into the pixel's run of the volume: collected four to a dword
	auto emit = [&](int k, unsigned c) {
		if (k < nD) {
			const unsigned pos = (idxLow + (unsigned)k) & 3u;               // byte of its dword in the volume
			packed |= c << (8u * pos);
			if (pos == 3u || k == nD - 1) {
				unsigned char* at = costs + idx + (unsigned)k;              // address of this (the last collected) byte
				const unsigned first = (unsigned)k < pos ? pos - (unsigned)k : 0u;   // first byte of the dword that belongs to this pixel
				if (pos == 3u && first == 0u) *reinterpret_cast<unsigned*>(at - 3) = packed;
				else for (unsigned bb = first; bb <= pos; ++bb) at[(int)bb - (int)pos] = (unsigned char)(packed >> (8u * bb));
				packed = 0u;
			}
		}
	};
#if SGM_UNI_PAIR
	// TWO disparities per trip: tap (i, j) of disparity k + 1 is tap (i, j + 1) of disparity k, so a row of the pair is eight strip values instead of fourteen -- 56 LDS
	// reads for two costs instead of 98 (the reads' latency, with three waves per SIMD, is what kept the kernel at two thirds of its issue floor).  Each cost's own taps
	// in its own order: n = row * 7 + column.
#pragma unroll 1
	for (int k = 0; k < nDall; k += 2) {
		float sumA = 0.f, sumSqA = 0.f, nomA = 0.f, sumB = 0.f, sumSqB = 0.f, nomB = 0.f;
		const float* sk = strip + k * CS;
		float fa[8], fb[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) fa[j] = sk[j * CS];
#pragma unroll
		for (int r = 0; r < 7; ++r) {
			float (&cur)[8] = (r & 1) ? fb : fa;
			float (&nxt)[8] = (r & 1) ? fa : fb;
			if (r < 6) {
#pragma unroll
				for (int j = 0; j < 8; ++j) nxt[j] = sk[j * CS + r + 1];
			}
			SGM_SCHED_BARRIER();
#pragma unroll
			for (int j = 0; j < 7; ++j) {
				const float f = cur[j];
				const float fw = f * wk[r * 7 + j];
				sumA += fw; sumSqA += f * fw; nomA += f * tk[r * 7 + j];
			}
			SGM_PIN3(sumA, sumSqA, nomA);
#pragma unroll
			for (int j = 0; j < 7; ++j) {
				const float f = cur[j + 1];
				const float fw = f * wk[r * 7 + j];
				sumB += fw; sumSqB += f * fw; nomB += f * tk[r * 7 + j];
			}
			SGM_PIN3(sumB, sumSqB, nomB);
			SGM_SCHED_BARRIER();
		}
		const int d = minDisp + k;
		const bool inA = !(ux - SGM_HW + d < 0 || ux + SGM_HW + d >= w), inB = !(ux - SGM_HW + d + 1 < 0 || ux + SGM_HW + d + 1 >= w);
		emit(k, inA ? (unsigned)sgm_cost_of(sumA, sumSqA, nomA, sumW, normSq0) : 255u);
		if (k + 1 < nDall) emit(k + 1, inB ? (unsigned)sgm_cost_of(sumB, sumSqB, nomB, sumW, normSq0) : 255u);
	}
#else
#pragma unroll 1
	for (int k = 0; k < nDall; ++k) {
		float sum = 0.f, sumSq = 0.f, nom = 0.f;
		// the 49 taps row by row, the next row's seven strip values requested while this row's are consumed (two register sets of seven; with all 49 reads hoisted to the
		// head of the iteration -- what the scheduler does with the flat loop -- the kernel needs 147 registers for weights and samples alone and spilled 41 of them into
		// scratch inside this loop).  Same taps, same order: n = row * 7 + column.
		const float* sk = strip + k * CS;
		float fa[7], fb[7];
#pragma unroll
		for (int j = 0; j < 7; ++j) fa[j] = sk[j * CS];
#pragma unroll
		for (int r = 0; r < 7; ++r) {
			float (&cur)[7] = (r & 1) ? fb : fa;
			float (&nxt)[7] = (r & 1) ? fa : fb;
			if (r < 6) {
#pragma unroll
				for (int j = 0; j < 7; ++j) nxt[j] = sk[j * CS + r + 1];
			}
			SGM_SCHED_BARRIER();
#pragma unroll
			for (int j = 0; j < 7; ++j) {
				const float f = cur[j];
				const float fw = f * wk[r * 7 + j];
				sum += fw; sumSq += f * fw; nom += f * tk[r * 7 + j];
			}
			SGM_PIN3(sum, sumSq, nom);     // the row's sums exist HERE (the compiler otherwise sinks all 49 taps' arithmetic below the reads, into the block that uses the cost)
			SGM_SCHED_BARRIER();
		}
		const int d = minDisp + k;
		const bool in = !(ux - SGM_HW + d < 0 || ux + SGM_HW + d >= w);
		emit(k, in ? (unsigned)sgm_cost_of(sum, sumSq, nom, sumW, normSq0) : 255u);
	}
#endif
}

// line start sets of one path direction: nA lines from (ax,ay) stepping (adx,ady), then the rest from (bx,by)
struct SGMLines { int nA, ax, ay, adx, ady, nB, bx, by, bdx, bdy; };

// ---- wave helpers ---------------------------------------------------------------------------
// wave64 minimum on the VALU cross-lane network (DPP row shifts + row broadcasts, then one readlane) instead of six
// dependent ds_bpermute round trips through the LDS crossbar: this reduction sits on the critical path of every step
// of the path recurrence.
// One step is a single v_min_i32_dpp (lanes without a source are write-disabled and keep their value); the compiler emits v_mov_b32_dpp + v_min_i32 and
// re-materialises the fill value for each, three instructions per step, when this is written with __builtin_amdgcn_update_dpp.  The s_nop covers the
// VALU-write -> DPP-read hazard (2 wait states), which the compiler cannot see inside an asm statement.
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define SGM_MIN_DPP(v, CTRLSTR, ctrl, rm) asm("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 " CTRLSTR : "+v"(v))
#else   // host pass and the CPU emulator of the test-suite
#define SGM_MIN_DPP(v, CTRLSTR, ctrl, rm) v = min(v, __builtin_amdgcn_update_dpp(v, v, ctrl, rm, 0xf, false))
#endif
__device__ __forceinline__ int sgm_wave_min(int v) {
	SGM_MIN_DPP(v, "row_shr:1 row_mask:0xf bank_mask:0xf", 0x111, 0xf);
	SGM_MIN_DPP(v, "row_shr:2 row_mask:0xf bank_mask:0xf", 0x112, 0xf);
	SGM_MIN_DPP(v, "row_shr:4 row_mask:0xf bank_mask:0xf", 0x114, 0xf);
	SGM_MIN_DPP(v, "row_shr:8 row_mask:0xf bank_mask:0xf", 0x118, 0xf);   // lane 15 of each row of 16 holds the row minimum
	SGM_MIN_DPP(v, "row_bcast:15 row_mask:0xa bank_mask:0xf", 0x142, 0xa); // into rows 1 and 3
	SGM_MIN_DPP(v, "row_bcast:31 row_mask:0xc bank_mask:0xf", 0x143, 0xc); // into rows 2 and 3: lane 63 holds the wave minimum
	return __builtin_amdgcn_readlane(v, 63);
}

// minimum over an aligned group of 16 lanes, returned in every lane of the group (quad butterflies, then mirrors inside 8 and 16 lanes)
__device__ __forceinline__ int sgm_sub_min16(int v) {
	v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));          // quad_perm [1,0,3,2]
	v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));          // quad_perm [2,3,0,1]
	v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));         // row_half_mirror
	v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));         // row_mirror
	return v;
}

// accums(d) += L(d) (:1020,1043).  The eight path kernels run concurrently on eight streams, so the sum is an atomic add;
// two u16 sums share a 32-bit word (no carry between halves: a sum never exceeds 8*(255+60) = 2520).  The lane whose entry sits
// in the low half adds its right neighbour's value in the same atomic; an entry in a high half whose low-half partner is not in
// this wave-instruction (lane 0) adds alone.  One predicated atomic per lane, no divergent control flow.  Called by all 64 lanes.
__device__ __forceinline__ void sgm_accumulate(unsigned* wordsBase, unsigned par, int k, int nD, int L, int lane) {
	const int Lnext = __builtin_amdgcn_update_dpp(L, L, 0x130, 0xf, 0xf, false);   // wave_shl:1 = value of entry k+1 (lane+1) on the VALU's DPP network, no LDS crossbar trip; not used by lane 63
	const unsigned e = ((unsigned)k + par) & 1u;                   // 0: this entry is the low half of its word
	const bool act = k < nD;
	const bool pair = lane < 63 && k + 1 < nD;
	const unsigned val = e == 0u ? ((unsigned)L | (pair ? (unsigned)Lnext << 16 : 0u)) : ((unsigned)L << 16);
	if (act && (e == 0u || lane == 0)) atomicAdd(wordsBase + (((unsigned)k + par) >> 1), val);
}

// ---- one path direction, SemiGlobalMatcher.cpp:1003-1046 + ACCUM_PIXELS :1065-1082 -----------
// One 64-thread workgroup (one wave) per line.  NK = ceil(maxNumDisp / 64) disparities per lane.
// Line start sets are the threaded variant's (:1083-1200); see the host for the numbering.
//
// The step of the recurrence is a chain of dependent instructions executed by one wave, and the whole aggregation is
// 8 x (pixels) such steps: its cost is the instruction count of a step.  Hence:
//  * the previous line of L sits in LDS with SGM_INF on both sides (a slot holds logical indices [-MD, 2*MD), MD = 64*NK) and
//    every step rewrites all of [0, MD) (L or SGM_INF), so Lp(d-1), Lp(d), Lp(d+1) are three unconditional reads at
//    k + (rsMin - rpMin) and the range tests of the reference reduce to "k > 0" and "k < nD-1";
//  * the pixel table of the line is staged through LDS 64 pixels at a time by one coalesced load per lane, so no per-pixel
//    state lives in scalar registers across the chunk (the first version spilled SGPRs into VGPR lanes in every step);
//  * cost bytes are prefetched one sub-chunk (SGM_T pixels) ahead in registers, A/B double-buffered without register moves;
//    the single s_waitcnt vmcnt(0) per sub-chunk also drains the accumulate atomics (loads and atomics share the counter, so
//    the compiler cannot wait for less);
//  * the workgroup is one wave and the LDS executes a wave's instructions in order: no barrier between steps.
#define SGM_TT 64        // pixels per pixel-table chunk
// A place where the code relies on a wavefront issuing its LDS instructions in program order: a store by one lane must not overtake an earlier
// load of the same location by another lane.  True by construction on the hardware (one wave, one instruction stream), so nothing is emitted; the
// CPU emulator the tests run these kernels under (tests/cpp/hipemu) executes lanes one after the other and defines this as a lane rendezvous.
#ifndef WAVE_LOCKSTEP_POINT
#define WAVE_LOCKSTEP_POINT() ((void)0)
#endif
struct SGMStep { int rpMin, rpMax, cur; float Ip; };

// DELTA: instead of adding L into the shared u16 sums with atomics, the direction writes L - C -- at most P2, a byte -- into its own byte volume (dvol); sgm_sum_wta_kernel
// forms 8 C + the eight deltas afterwards.  Every entry of every valid pixel lies on exactly one line of each direction, so each byte is written exactly once.
template <int NK, bool DELTA>
__device__ __forceinline__ void sgm_step(int* sL, const unsigned short* sP2, const SGMPixel& px, float g, const unsigned char* c8,
		unsigned* accumWords, unsigned char* dvol, int P1, int lane, SGMStep& st) {
	constexpr int MD = 64 * NK, SL = 3 * MD;
	const int rsMin = px.minDisp, rsMax = px.maxDisp, nD = rsMax - rsMin;
	if (nD <= 0) return;                                            // invalid pixels do not reset Lp / Ip (:1071-1072)
	const float DI = g - st.Ip;
	int ip = sgm_round2int(255.f * DI); ip = ip < 0 ? -ip : ip;
	const int P2 = sP2[ip];
	const int lo = max(st.rpMin, rsMin), hi = min(st.rpMax, rsMax);
	const int* Lp = sL + st.cur * SL + MD + (rsMin - st.rpMin);     // Lp[k] = previous L at the disparity of entry k
	int* Ls = sL + (st.cur ^ 1) * SL + MD;
	unsigned* wordsBase = accumWords + (px.idx >> 1);
	const unsigned par = (unsigned)(px.idx & 1ull);
	if (lo >= hi) {                                                 // no common disparity (also the first pixel of a line)
#pragma unroll
		for (int q = 0; q < NK; ++q) {
			const int k = lane + 64 * q;
			const int L = (int)c8[q] + P2;
			Ls[k] = k < nD ? L : SGM_INF;
			if (DELTA) { if (k < nD) dvol[px.idx + (unsigned)k] = (unsigned char)P2; }
			else sgm_accumulate(wordsBase, par, k, nD, L, lane);
		}
	} else {
		int a0[NK], am[NK], ap[NK];
		int m = SGM_INF;
#pragma unroll
		for (int q = 0; q < NK; ++q) {
			const int k = lane + 64 * q;
			a0[q] = k < nD ? Lp[k] : SGM_INF; am[q] = Lp[k - 1]; ap[q] = Lp[k + 1];
			m = min(m, a0[q]);
		}
		m = sgm_wave_min(m);
#pragma unroll
		for (int q = 0; q < NK; ++q) {
			const int k = lane + 64 * q;
			const int side = min(k > 0 ? am[q] : SGM_INF, k < nD - 1 ? ap[q] : SGM_INF) + P1;
			const int best = min(min(m + P2, a0[q]), side);
			const int L = (int)c8[q] + best - m;
			Ls[k] = k < nD ? L : SGM_INF;
			if (DELTA) { if (k < nD) dvol[px.idx + (unsigned)k] = (unsigned char)(best - m); }
			else sgm_accumulate(wordsBase, par, k, nD, L, lane);
		}
	}
	st.rpMin = rsMin; st.rpMax = rsMax; st.Ip = g; st.cur ^= 1;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// All eight directions in one grid: the lines of direction i are the workgroups first[i] .. first[i+1]-1.  (Eight launches on eight
// streams shared four hardware queues and overlapped only ~2.4x.)  The host lists the directions longest lines first.
struct SGMDirs { int dx[8], dy[8]; SGMLines ln[8]; int first[9]; };

template <int NK, bool DELTA = false>
__global__ __launch_bounds__(64) void sgm_path_kernel(const float* __restrict__ grayL, int w, int vw, int vh,
		const SGMPixel* __restrict__ pixels, const unsigned char* __restrict__ costs, unsigned* __restrict__ accumWords,
		const unsigned short* __restrict__ P2s, int P1, SGMDirs dirs, unsigned char* __restrict__ deltas = nullptr, unsigned long long numCosts = 0) {
	constexpr int MD = 64 * NK, SL = 3 * MD;
	__shared__ int s_L[2 * SL];
	__shared__ unsigned short s_P2[256];
	__shared__ SGMPixel s_px[2][SGM_TT];
	__shared__ float s_g[2][SGM_TT];
	const int lane = threadIdx.x;
	int dir = 0;
#pragma unroll
	for (int i = 1; i < 8; ++i) dir += (int)blockIdx.x >= dirs.first[i] ? 1 : 0;
	const int line = (int)blockIdx.x - dirs.first[dir];
	const int dx = dirs.dx[dir], dy = dirs.dy[dir];
	const SGMLines ln = dirs.ln[dir];
	unsigned char* dvol = DELTA ? deltas + (unsigned long long)dir * numCosts : nullptr;   // this direction's byte volume
	int x, y;
	if (line < ln.nA) { x = ln.ax + line * ln.adx; y = ln.ay + line * ln.ady; }
	else { const int i = line - ln.nA; x = ln.bx + i * ln.bdx; y = ln.by + i * ln.bdy; }
	for (int k = lane; k < 2 * SL; k += 64) s_L[k] = SGM_INF;
	for (int k = lane; k < 256; k += 64) s_P2[k] = P2s[k];
	// pixel-table chunk: lane t owns pixel t of the chunk
	auto tableLoad = [&](int cx, int cy, SGMPixel& px, float& g) {
		const int tx = cx + lane * dx, ty = cy + lane * dy;
		px.idx = 0; px.minDisp = 0; px.maxDisp = 0; px.pad = 0; g = 0.f;
		if (tx >= 0 && ty >= 0 && tx < vw && ty < vh) {
			px = pixels[(size_t)ty * vw + tx];
			g = grayL[(size_t)ty * w + tx]; // imageGray(u) with the valid-grid coordinate: the reference's quirk (:1078)
		}
	};
	auto costLoad = [&](int slot, int t0, unsigned char (*c8)[NK]) {   // cost bytes of pixels t0..t0+SGM_T-1 of table slot `slot`
#pragma unroll
		for (int t = 0; t < SGM_T; ++t) {
			const SGMPixel px = s_px[slot][t0 + t];
			const int nD = px.maxDisp - px.minDisp;
#pragma unroll
			for (int q = 0; q < NK; ++q) {
				const int k = lane + 64 * q;
				c8[t][q] = 0;
				if (k < nD) c8[t][q] = costs[px.idx + k];
			}
		}
	};
	SGMPixel tpx; float tg;
	tableLoad(x, y, tpx, tg);
	s_px[0][lane] = tpx; s_g[0][lane] = tg;                         // (waits for the loads)
	tableLoad(x + SGM_TT * dx, y + SGM_TT * dy, tpx, tg);           // chunk 1, in flight
	__syncthreads();
	unsigned char cA[SGM_T][NK], cB[SGM_T][NK];
	costLoad(0, 0, cA);
	SGMStep st; st.rpMin = 0; st.rpMax = 0; st.cur = 0; st.Ip = 0.5f;
	int slot = 0;
	while (x >= 0 && y >= 0 && x < vw && y < vh) {
		// table of the next chunk: complete by now (requested a whole chunk ago); park it in the other slot and request the one after
		__builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
		WAVE_LOCKSTEP_POINT();                                        // every lane is done reading slot^1 (the chunk before this one)
		s_px[slot ^ 1][lane] = tpx; s_g[slot ^ 1][lane] = tg;
		tableLoad(x + 2 * SGM_TT * dx, y + 2 * SGM_TT * dy, tpx, tg);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
#pragma unroll 1
		for (int t0 = 0; t0 < SGM_TT; t0 += 2 * SGM_T) {
			// sub-chunk A is in registers (or arriving); request B, consume A; then request the next A, consume B
			__builtin_amdgcn_s_waitcnt(0x0F70);
			costLoad(slot, t0 + SGM_T, cB);
#pragma unroll
			for (int t = 0; t < SGM_T; ++t) sgm_step<NK, DELTA>(s_L, s_P2, s_px[slot][t0 + t], s_g[slot][t0 + t], cA[t], accumWords, dvol, P1, lane, st);
			__builtin_amdgcn_s_waitcnt(0x0F70);
			if (t0 + 2 * SGM_T < SGM_TT) costLoad(slot, t0 + 2 * SGM_T, cA); else costLoad(slot ^ 1, 0, cA);
#pragma unroll
			for (int t = 0; t < SGM_T; ++t) sgm_step<NK, DELTA>(s_L, s_P2, s_px[slot][t0 + SGM_T + t], s_g[slot][t0 + SGM_T + t], cB[t], accumWords, dvol, P1, lane, st);
		}
		x += SGM_TT * dx; y += SGM_TT * dy; slot ^= 1;
	}
}

// ---- the same recurrence when every pixel of the valid grid has the same disparity range (the first level of tSGM and a plain `Match`:
// SemiGlobalMatcher.cpp:874-896 gives all pixels [minDisp, maxDisp)) ---------------------------------------------------------------------------------
// With one range, entry k of a pixel is disparity minDisp + k for every pixel, so the previous pixel's L needs no re-indexing: the line of L stays in
// registers (lane l owns entries l*NK .. l*NK+NK-1), Lp(d-1) / Lp(d+1) are the neighbouring lanes' registers (DPP wave shifts, SGM_INF shifted in at
// both ends), and PixelData::idx is pixel * nD -- no pixel table, no LDS line buffer, no fences.  The penalty P2 of a step depends only on the grey
// values of this pixel and the one before it (:1073-1079), so a chunk of 64 pixels computes its 64 penalties at once, one per lane, and a step
// fetches its own with v_readlane.  A step is then ~25 VALU instructions instead of ~90 (+ ~50 scalar ones); results are the same integers.
// The host checks the premise (sgm_uniform_check_kernel) and falls back to sgm_path_kernel when it does not hold.
struct SGMUniform { int ok, minDisp, maxDisp, pad; };
__global__ __launch_bounds__(256) void sgm_uniform_check_kernel(const SGMPixel* __restrict__ pixels, long nPix, SGMUniform* __restrict__ out) {
	const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (pix >= nPix) return;
	const SGMPixel p0 = pixels[0], px = pixels[pix];
	const int nD = p0.maxDisp - p0.minDisp;
	const bool same = nD > 0 && px.minDisp == p0.minDisp && px.maxDisp == p0.maxDisp && px.idx == (unsigned long long)pix * (unsigned long long)nD;
	if (!same) out->ok = 0;                                           // (every writer stores the same value)
	if (pix == 0) { out->minDisp = p0.minDisp; out->maxDisp = p0.maxDisp; }
}

#ifndef SGM_UT
#define SGM_UT 8         // pixels per cost prefetch sub-chunk of the uniform-range kernel
#endif
//
// DELTA: no atomics at all.  L(d) = C(d) + (best - min Lp) and 0 <= best - min Lp <= P2, so a direction only has to record that one byte per entry, in its own
// volume (deltas + dir * numCosts), with plain coalesced stores; sgm_sum_wta_kernel then forms sum_r L_r(d) = 8 C(d) + sum_r delta_r(d), writes the u16 sums the
// reference keeps (imageAccumCosts) and takes the winner in the same pass.  The sums are the same integers in any order.  Needs max P2 <= 255 (host check) and
// 8 bytes per entry of scratch; replaces 3.2 GB of 16-bit atomic payload (the limiter of the atomic version) by 1.6 GB of byte stores.
// STAGE (DELTA, nD a multiple of 16): the delta bytes of 32 consecutive pixels of the line are collected in LDS and written out as 16-byte pieces -- two store
// instructions per lane and 32 steps instead of one byte store per step.  With a store in flight in every step each use of a prefetched cost byte had to wait for
// *everything* outstanding (loads and stores share vmcnt and complete out of order with respect to each other, so the compiler waits for zero): the step was as long
// as a store round trip.
#define SGM_OC 32
template <int NK, int ALIGN, bool DELTA, bool STAGE = false>   // ALIGN: 2 if nD is even (every pixel's sums start on a 32-bit word: idx = pixel * nD), else 1
__global__ __launch_bounds__(64) void sgm_path_uniform_kernel(const float* __restrict__ grayL, int w, int vw, int vh, int nD,
		const unsigned char* __restrict__ costs, unsigned* __restrict__ accumWords, const unsigned short* __restrict__ P2s, int P1, SGMDirs dirs,
		unsigned char* __restrict__ deltas, unsigned long long numCosts) {
	__shared__ unsigned short s_P2[256];
	__shared__ __attribute__((aligned(16))) unsigned char s_out[STAGE ? SGM_OC * 64 * NK : 16];
	const int lane = threadIdx.x;
	int dir = 0;
#pragma unroll
	for (int i = 1; i < 8; ++i) dir += (int)blockIdx.x >= dirs.first[i] ? 1 : 0;
	const int line = (int)blockIdx.x - dirs.first[dir];
	const int dx = dirs.dx[dir], dy = dirs.dy[dir];
	const SGMLines ln = dirs.ln[dir];
	int x, y;
	if (line < ln.nA) { x = ln.ax + line * ln.adx; y = ln.ay + line * ln.ady; }
	else { const int i = line - ln.nA; x = ln.bx + i * ln.bdx; y = ln.by + i * ln.bdy; }
	for (int k = lane; k < 256; k += 64) s_P2[k] = P2s[k];
	__syncthreads();
	if (x < 0 || y < 0 || x >= vw || y >= vh) return;
	// pixels on the line (UNIFORM_KERNEL_BODY)
	int n = 0x7fffffff;
	if (dx > 0) n = min(n, vw - x); else if (dx < 0) n = min(n, x + 1);
	if (dy > 0) n = min(n, vh - y); else if (dy < 0) n = min(n, y + 1);
	const long long idx0 = ((long long)y * vw + x) * nD;              // PixelData::idx of the first pixel, and its step along the line
	const long long dIdx = ((long long)dy * vw + dx) * nD;
	static_assert(NK == 1 || NK == 2, "lane-local word pairs are written out for one or two entries per lane");
	const int k0 = lane * NK;
	auto grayLoad = [&](int i0) -> float {                           // grey value of pixel i0 + lane of the line (valid-grid coordinate on the full image: the reference's quirk, :1078)
		const int i = i0 + lane;
		return i < n ? grayL[(size_t)(y + i * dy) * w + (x + i * dx)] : 0.f;
	};
	// cost bytes of pixels i0 .. i0+SGM_UT-1 of the line.  Lanes past the range read the bytes that follow (the next pixel's costs; the volume is
	// allocated with 256 spare bytes) and never use them, which keeps the loads free of per-lane predicates.
	// The address of a pixel's costs is wave-uniform: it is kept in a scalar register pair that steps along the line (pinned there: left to itself the compiler
	// folds the per-pixel offsets into 64-bit VGPR pairs, one pair per prefetched pixel -- 154 VGPRs, 3 waves per SIMD instead of 8).
	auto costLoad = [&](int i0, unsigned char (*c8)[NK]) {
		unsigned long long base = (unsigned long long)costs + (unsigned long long)(idx0 + (long long)i0 * dIdx);
		const bool all = i0 + SGM_UT <= n;                              // (uniform) all of them on the line
#pragma unroll
		for (int t = 0; t < SGM_UT; ++t) {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
			asm volatile("" : "+s"(base));
#endif
			const sgm_gcb bp = (sgm_gcb)base;
			if (all || i0 + t < n) {
				if (NK == 2 && ALIGN == 2) { const unsigned short v = ((sgm_gcs)base)[(unsigned)lane]; c8[t][0] = (unsigned char)(v & 255); c8[t][NK - 1] = (unsigned char)(v >> 8); }
				else {
#pragma unroll
					for (int q = 0; q < NK; ++q) c8[t][q] = bp[(unsigned)(k0 + q)];
				}
			} else {
#pragma unroll
				for (int q = 0; q < NK; ++q) c8[t][q] = 0;
			}
			base += (unsigned long long)dIdx;
		}
	};
	int L[NK];
#pragma unroll
	for (int q = 0; q < NK; ++q) L[q] = SGM_INF;
	auto step = [&](int i, int P2, const unsigned char* c8) {
		if (i >= n) return;
		const long long idx = idx0 + (long long)i * dIdx;
		int Ln[NK], dl[NK];
		if (i == 0) {                                                   // no previous pixel: L = C + P2 (:1012-1021)
#pragma unroll
			for (int q = 0; q < NK; ++q) { Ln[q] = (int)c8[q] + P2; dl[q] = P2; }
		} else {
			int m = L[0];
#pragma unroll
			for (int q = 1; q < NK; ++q) m = min(m, L[q]);
			m = sgm_wave_min(m);
			const int mP2 = m + P2;
			const int fromLeft = __builtin_amdgcn_update_dpp(SGM_INF, L[NK - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1: entry k0-1 (lane 0: none)
			const int fromRight = __builtin_amdgcn_update_dpp(SGM_INF, L[0], 0x130, 0xf, 0xf, false);        // wave_shl:1: entry k0+NK (lane 63: none)
#pragma unroll
			for (int q = 0; q < NK; ++q) {
				const int am = q == 0 ? fromLeft : L[q - 1], ap = q == NK - 1 ? fromRight : L[q + 1];
				const int best = min(min(mP2, L[q]), min(am, ap) + P1);
				dl[q] = best - m;
				Ln[q] = (int)c8[q] + dl[q];
			}
		}
#pragma unroll
		for (int q = 0; q < NK; ++q) L[q] = k0 + q < nD ? Ln[q] : SGM_INF;
		if (DELTA && STAGE) {
			unsigned char* row = s_out + (i & (SGM_OC - 1)) * (64 * NK) + k0;   // every lane writes; bytes past nD are never flushed
			if (NK == 2) *reinterpret_cast<unsigned short*>(row) = (unsigned short)((dl[0] & 255) | (dl[NK - 1] << 8));
			else row[0] = (unsigned char)dl[0];
			return;
		}
		if (DELTA) {
			unsigned char* out = deltas + (unsigned long long)dir * numCosts + idx;
			if (NK == 2 && ALIGN == 2) { if (k0 + 1 < nD) *reinterpret_cast<unsigned short*>(out + k0) = (unsigned short)(dl[0] | (dl[NK - 1] << 8)); else if (k0 < nD) out[k0] = (unsigned char)dl[0]; }
			else {
#pragma unroll
				for (int q = 0; q < NK; ++q) if (k0 + q < nD) out[k0 + q] = (unsigned char)dl[q];
			}
			return;
		}
		// accums(d) += L(d): as sgm_accumulate, with the pair of a word taken from this lane's registers where it can be
		unsigned* words = accumWords + (idx >> 1);
		const unsigned par = ALIGN >= 2 ? 0u : (unsigned)(idx & 1ll);
		unsigned v[NK + 1];
#pragma unroll
		for (int q = 0; q < NK; ++q) v[q] = k0 + q < nD ? (unsigned)L[q] : 0u;
		v[NK] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[0], 0x130, 0xf, 0xf, false);   // the next lane's first entry (lane 63: none)
		// (Four sums per 64-bit atomic was tried for ranges that are multiples of 4: 2.46 instead of 2.24 ms at D = 64, profiles/r03_sgm_call11.log.)
		if (NK == 1) {
			// par == 0: even lanes hold the low halves and add the pair; par == 1: odd lanes do, and entry 0 (the high half of the first word) adds alone
			if (par == 0u) { if (!(lane & 1) && k0 < nD) atomicAdd(words + (lane >> 1), v[0] | (v[1] << 16)); }
			else if (k0 < nD) { if (lane & 1) atomicAdd(words + ((lane + 1) >> 1), v[0] | (v[1] << 16)); else if (lane == 0) atomicAdd(words, v[0] << 16); }
		} else {
			// lane l owns entries 2l, 2l+1: with par == 0 they are the halves of word l; with par == 1 the pair is (2l+1, 2l+2), completed by the next
			// lane's first entry, and entry 0 of the pixel adds alone
			if (par == 0u) { if (k0 < nD) atomicAdd(words + lane, v[0] | (v[1] << 16)); }
			else {
				if (k0 + 1 < nD) atomicAdd(words + lane + 1, v[1] | (v[2] << 16));
				if (lane == 0) atomicAdd(words, v[0] << 16);
			}
		}
	};
	// staged delta bytes of pixels first .. first+count-1 of the line (count <= SGM_OC) -> their volume, 16 bytes per lane and trip
	auto flush = [&](int first, int count) {
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const int pp = nD >> 4;                                         // 16-byte pieces per pixel
		unsigned char* vol = deltas + (unsigned long long)dir * numCosts;
		for (int p = lane; p < count * pp; p += 64) {
			const int r = p / pp, part = p - r * pp;
			const float4 v = *reinterpret_cast<const float4*>(s_out + ((first + r) & (SGM_OC - 1)) * (64 * NK) + part * 16);
			*reinterpret_cast<float4*>(vol + (idx0 + (long long)(first + r) * dIdx) + part * 16) = v;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	};
	float g = grayLoad(0), gNext = grayLoad(64);
	float carry = 0.5f;                                               // Ip before the first pixel (:1066)
	unsigned char cA[SGM_UT][NK], cB[SGM_UT][NK];
	costLoad(0, cA);
#pragma unroll 1
	for (int i0 = 0; i0 < n; i0 += 64) {
		// the 64 penalties of this chunk: lane t has pixel i0 + t, its predecessor's grey value comes from lane t-1 (lane 0: the chunk before)
		const float Ip = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, carry), __builtin_bit_cast(int, g), 0x138, 0xf, 0xf, false));
		int ip = sgm_round2int(255.f * (g - Ip)); ip = ip < 0 ? -ip : ip;
		const int P2v = (int)s_P2[ip > 255 ? 255 : ip];
		carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), 63));
		g = gNext; gNext = grayLoad(i0 + 128);
#pragma unroll 1
		for (int s = 0; s < 64 && i0 + s < n; s += 2 * SGM_UT) {
			costLoad(i0 + s + SGM_UT, cB);
#pragma unroll
			for (int t = 0; t < SGM_UT; ++t) step(i0 + s + t, __builtin_amdgcn_readlane(P2v, s + t), cA[t]);
			costLoad(i0 + s + 2 * SGM_UT, cA);
#pragma unroll
			for (int t = 0; t < SGM_UT; ++t) step(i0 + s + SGM_UT + t, __builtin_amdgcn_readlane(P2v, s + SGM_UT + t), cB[t]);
			if (DELTA && STAGE) {
				const int done = min(n, i0 + s + 2 * SGM_UT);                 // pixels of the line finished so far
				if ((done & (SGM_OC - 1)) == 0 || done == n) { const int first = (done - 1) & ~(SGM_OC - 1); flush(first, done - first); }
			}
		}
	}
}

// ---- sums and winner of the DELTA aggregation: accums(d) = 8 C(d) + sum over the 8 directions of delta_r(d); first minimum (:1272-1301) -----------------
// 16 lanes per pixel, four pixels per wave.  A lane takes whole dwords of the byte volumes: the pixel's run [idx, idx + nD) is covered by the dwords from idx & ~3 on, so
// every load is aligned whatever idx is (ragged ranges start anywhere); bytes of the first and last dword that belong to the neighbouring pixels are masked out.
__global__ __launch_bounds__(256) void sgm_sum_wta_kernel(const SGMPixel* __restrict__ pixels, const unsigned char* __restrict__ costs, const unsigned char* __restrict__ deltas,
		unsigned long long numCosts, unsigned short* __restrict__ accums, long nPix, short* __restrict__ disp, unsigned short* __restrict__ cost) {
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane >> 4, kk = lane & 15;
	const long pix = ((long)blockIdx.x * 4 + wave) * 4 + sub;
	SGMPixel px; px.idx = 0; px.minDisp = 0; px.maxDisp = 0; px.pad = 0;
	const bool have = pix < nPix;
	if (have) px = pixels[pix];
	const int nD = px.maxDisp - px.minDisp;
	unsigned key = 0xFFFFFFFFu;
	if ((numCosts & 3ull) == 0ull) {             // the per-direction volumes start numCosts apart: dword loads keep their alignment in every one of them
		const int head = (int)(px.idx & 3ull);       // bytes of the first dword that precede the run
		const unsigned long long base = px.idx - (unsigned long long)head;
		for (int j = kk * 4; j < head + nD; j += 64) {   // j = byte offset of this lane's dword from `base`
			const unsigned c = *reinterpret_cast<const unsigned*>(costs + base + j);
			unsigned s[4];
#pragma unroll
			for (int b = 0; b < 4; ++b) s[b] = ((c >> (8 * b)) & 255u) * 8u;
#pragma unroll
			for (int r = 0; r < 8; ++r) {
				const unsigned dv = *reinterpret_cast<const unsigned*>(deltas + (unsigned long long)r * numCosts + base + j);
#pragma unroll
				for (int b = 0; b < 4; ++b) s[b] += (dv >> (8 * b)) & 255u;
			}
			const int k0 = j - head;                     // entry of byte 0 of the dword
			if (k0 >= 0 && k0 + 4 <= nD) {
				uint2 o; o.x = s[0] | (s[1] << 16); o.y = s[2] | (s[3] << 16);
				*reinterpret_cast<uint2*>(accums + base + j) = o;          // (base + j) % 4 == 0: 8-byte aligned
#pragma unroll
				for (int b = 0; b < 4; ++b) key = min(key, (s[b] << 16) | (unsigned)(k0 + b));
			} else {
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const int k = k0 + b;
					if (k >= 0 && k < nD) { accums[px.idx + (unsigned)k] = (unsigned short)s[b]; key = min(key, (s[b] << 16) | (unsigned)k); }
				}
			}
		}
	} else {
		for (int k = kk; k < nD; k += 16) {
			unsigned v = (unsigned)costs[px.idx + k] * 8u;
			for (int r = 0; r < 8; ++r) v += deltas[(unsigned long long)r * numCosts + px.idx + k];
			accums[px.idx + k] = (unsigned short)v;
			key = min(key, (v << 16) | (unsigned)k);
		}
	}
	key = (unsigned)sgm_sub_min16((int)(key ^ 0x80000000u)) ^ 0x80000000u;
	if (have && kk == 0) {
		if (nD <= 0) { disp[pix] = px.minDisp; cost[pix] = 0xFFFF; }
		else { disp[pix] = (short)(px.minDisp + (int)(key & 0xFFFFu)); cost[pix] = (unsigned short)(key >> 16); }
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
