// pm_kernels.hip -- CDNA4 (gfx950) kernels of the PatchMatch depth-map estimator.
//
// What is computed: the reference's CPU estimator (libs/MVS/DepthMap.cpp:415-971, passes of
// libs/MVS/SceneDensify.cpp:490-576), written from scratch for a 64-wide wavefront machine.
//
// Work decomposition ("pixel group" mapping): G = next_pow2(#source views) adjacent lanes own one
// pixel; lane v of the group scores the hypothesis against source view v (25 bilinear taps walked
// serially, exactly the reference's accumulation order), the two smallest view scores are found
// with a log2(G)-step __shfl_xor butterfly inside the group (wave64 cross-lane, no LDS traffic),
// and everything that is per pixel (RNG, plane perturbation, accept/reject) is evaluated
// redundantly by the G lanes so no broadcast is needed.  A wave64 therefore advances 64/G pixels
// and a 256-thread workgroup 256/G.  The 25 bilateral patch weights of each pixel are computed
// cooperatively by its G lanes and staged in LDS (float2 {w, w*(I-mean)} per tap, read back as
// broadcast ds_read_b64).
//
// Schedule: the CPU sweep visits pixels in anti-diagonal order (MapMatrix2ZigzagIdx,
// DepthMap.cpp:329-356): a pixel sees *new* values at its left/top neighbours and *old* values at
// right/bottom (DepthMap.cpp:641-766).  All pixels of one anti-diagonal are independent, so one
// launch processes one anti-diagonal of every reference view in the batch (grid.y = views), in place;
// launches on one stream order the diagonals.  This reproduces the sequential result bit for bit.
//
// No MFMA: the path is a gather stencil.  No FMA contraction (-ffp-contract=off) -- see pm_math.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "pm_math.h"

#define PM_MAX_SRC 16
#ifndef PM_BLOCK
#define PM_BLOCK 64    // one wave per workgroup: LDS windows are per wave, so small workgroups pack the CU's 160 KB best (tools/tune.py)
#endif
#ifndef PM_USE_TILES
#define PM_USE_TILES 1   // stage source-image windows in LDS in the sweep kernel
#endif
#ifndef PM_TAPCHUNK
#define PM_TAPCHUNK 5   // taps of a row whose loads are issued together (5 = whole row)
#endif
#ifndef PM_MINWAVES
#define PM_MINWAVES 3   // waves per SIMD the register allocator must leave room for (measured: 3 beats 4 (spills) and 5)
#endif
// Pointers read out of PMTask live in the generic address space as far as the compiler knows, which
// turns every access into a flat_load; they are all HBM buffers, so say so (global_load, own vmcnt).
typedef const float __attribute__((address_space(1)))* pm_gcf;
typedef float __attribute__((address_space(1)))* pm_gf;
// 16-byte entries of the quad images (PMSrcView::imgQ): one global_load_dwordx4 per bilinear sample
#if defined(__HIP_DEVICE_COMPILE__)
typedef float pm_f4v __attribute__((ext_vector_type(4)));
typedef const pm_f4v __attribute__((address_space(1)))* pm_gcf4;
__device__ __forceinline__ pm_gcf4 pm_glob4(const float4* p) { return (pm_gcf4)p; }
__device__ __forceinline__ void pm_load4(pm_gcf4 base, unsigned off, float& a, float& b, float& c, float& d) { const pm_f4v t = base[off]; a = t.x; b = t.y; c = t.z; d = t.w; }
#else
typedef const float4* pm_gcf4;
__device__ __forceinline__ pm_gcf4 pm_glob4(const float4* p) { return p; }
__device__ __forceinline__ void pm_load4(pm_gcf4 base, unsigned off, float& a, float& b, float& c, float& d) { const float4 t = base[off]; a = t.x; b = t.y; c = t.z; d = t.w; }
#endif
__device__ __forceinline__ pm_gcf pm_glob(const float* p) { return (pm_gcf)p; }
__device__ __forceinline__ pm_gf pm_globw(float* p) { return (pm_gf)p; }
#define PM_HW 4      // nSizeHalfWindow, DepthMap.h:277
#define PM_NT 25     // nTexels, DepthMap.h:281

struct PMSrcView {
	// "hot" block, 13 doubles: what every hypothesis evaluation reads of its source view.  The sweep kernel copies it (and the geometric block)
	// into LDS once per visit; the layout is the copy's contract (see PM_SRC_HOT / PM_SRC_GEO below).
	double Hl[9];         // K_j R_j R_0^T          (ViewData::Init, DepthMap.h:175-185)
	double Hm[3];         // K_j R_j (C_0 - C_j)
	int w, h;             // size of the source image at this level
	// geometric block, 14 doubles: transforms of the consistency term and the source view's depth-map (nullable; geometric pass) with its own
	// size: the map is addressed through cameraDepthMap (Tl..Tn), not through the image's camera (DepthMap.h:170-171, DepthMap.cpp:535-551)
	float Tl[9], Tm[3], Tr[9], Tn[3];
	const float* depth;
	int dw, dh;
	const float* img;     // source image at this pyramid level, row-major
	const float* imgS;    // same image, anti-diagonal-major ("skewed"): texel (u,v) at (u+v)*h + v
	const float4* imgQ;   // anti-diagonal-major "quad" image: entry (u,v) = {I(u,v), I(u+1,v), I(u,v+1), I(u+1,v+1)} -- the four texels of a bilinear
	                      // sample in ONE 16-byte load (the window-less tap rows are bound by the number of vector-memory instructions, not by bytes:
	                      // the stride-2 taps of a patch never share texels, so the quad image is read at the same byte rate as the plain one)
};
#define PM_SRC_HOT 13     // doubles
#define PM_SRC_GEO 14
static_assert(offsetof(PMSrcView, Hm) == 72 && offsetof(PMSrcView, w) == 96 && offsetof(PMSrcView, Tl) == 8 * PM_SRC_HOT
	&& offsetof(PMSrcView, depth) == 8 * PM_SRC_HOT + 96 && offsetof(PMSrcView, dw) == 8 * PM_SRC_HOT + 104 && offsetof(PMSrcView, img) == 8 * (PM_SRC_HOT + PM_SRC_GEO), "PMSrcView layout");
struct PMTask {           // one reference view at one pyramid level
	float* depth; float* normal; float* conf;
	const float* prior;   // nullable: low-resolution depth prior at this level
	const float* ref;     // reference image at this level, row-major
	const float* refS;    // reference image, anti-diagonal-major
	const unsigned char* mask; // nullable: ignore mask at this level, 0 = pixel is not estimated (DepthData::ApplyIgnoreMask + masked MapMatrix2ZigzagIdx)
	int w, h, nSrc, pad0;
	double Hr[9];         // K_0^-1
	int hrUpper;          // 1 if Hr[1] == Hr[3] == Hr[6] == Hr[7] == 0 exactly (zero-skew K): products with those vanish exactly
	int pad1;
	double fx, fy, cx, cy;
	float dMin, dMax, dMinSqr, dMaxSqr;
	uint32_t k0, k1base;  // Philox key: (seed, viewID*0x9E3779B1 + pass)
	PMSrcView src[PM_MAX_SRC];
};
struct PMKParams {        // DepthEstimator ctor constants, DepthMap.cpp:397-406
	float smoothBonusDepth, smoothBonusNormal, smoothSigmaDepth, smoothSigmaNormal;
	float thMagnitudeSq, angle1Range, angle2Range, thConfSmall, thConfBig, thConfRand, thRobust;
	float thKeep, geoWeight, depthRatio;
	uint32_t nRandomIters;
};

enum { PM_STREAM_INIT = 0, PM_STREAM_RAND = 1, PM_STREAM_REFINE = 2 };

#define PM_INF __builtin_huge_valf()
// the value must exist in a register at this point (device only; nothing for the host build of the emulator)
#if defined(__HIP_DEVICE_COMPILE__)
#define PM_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define PM_OPAQUE(x) do {} while (0)
#endif

// Optional in-kernel phase timing (build with -DPM_PROFILE): lane 0 of every wave accumulates s_memtime deltas
// per phase into pm_prof[]; read back with pmhip_prof_get.  Phases: 0 setup (weights, neighbour gather, tiles),
// 1 hypothesis generation, 2 smoothness factors, 3 homography, 4 taps, 5 score epilogue, 6 aggregation+accept,
// 7 number of outer trips, 8 trips x active pixel-lanes, 9 waves, 10 tap rows served from LDS, 11 tap rows total; the sweep kernel splits
// phase 0 further: 12 = head of the visit (own + neighbour estimates, patch texels, weights), 13 = neighbour set-up + window placement, 0 = window staging.
#ifdef PM_PROFILE
__device__ unsigned long long pm_prof[16];
struct PmProfAcc { unsigned long long a[16]; unsigned long long t; };
#define PM_PROF_ARG , PmProfAcc& _pa
#define PM_PROF_PASS , _pa
#define PM_PROF_DECL PmProfAcc _pa; for (int _i = 0; _i < 16; ++_i) _pa.a[_i] = 0; _pa.t = __builtin_readcyclecounter()
#define PM_TICK(i) do { const unsigned long long _n = __builtin_readcyclecounter(); _pa.a[i] += _n - _pa.t; _pa.t = _n; } while (0)
#define PM_COUNT(i, n) do { _pa.a[i] += (unsigned long long)(n); } while (0)
#define PM_PROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int _i = 0; _i < 16; ++_i) atomicAdd(&pm_prof[_i], _pa.a[_i]); } while (0)
#else
#define PM_PROF_ARG
#define PM_PROF_PASS
#define PM_PROF_DECL do {} while (0)
#define PM_TICK(i) do {} while (0)
#define PM_COUNT(i, n) do {} while (0)
#define PM_PROF_FLUSH() do {} while (0)
#endif
#define PM_FD2R(d) ((d) * (PM_PI_F / 180.f))

// ---- small device helpers -----------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int pm_mul24(int a, int b) { return __mul24(a, b); }
#else
__device__ __forceinline__ int pm_mul24(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
#endif
__device__ __forceinline__ bool pm_inside1(float px, float py, int w, int h) {
	// isInsideWithBorder<float,1>, libs/Common/Types.h:1649-1651
	return px >= 1.f && py >= 1.f && px <= (float)(w - 2) && py <= (float)(h - 2);
}
__device__ __forceinline__ bool pm_in_range(float d, float lo, float hi) { return lo <= d && d < hi; } // ISINSIDE, Types.h:1193

// Dir2Normal / Normal2Dir, libs/Common/Util.inl:754-766
__device__ __forceinline__ void pm_dir2normal(float p0, float p1, float& nx, float& ny, float& nz) {
	float sx, cx, sy, cy;
#ifdef PM_PROBE_CHEAP_TRIG
	sx = p0 * 0.3f; cx = 1.f - sx * sx; sy = -0.2f + p1 * 0.01f; cy = -0.97f;
#else
	pm_sincosf(p0, &sx, &cx); pm_sincosf(p1, &sy, &cy);
#endif
	nx = cx * sy; ny = sx * sy; nz = cy;
}
// RandomNormal, DepthMap.h:439-444
__device__ __forceinline__ void pm_random_normal(float u1, float u2, float vx, float vy, float vz, float& nx, float& ny, float& nz) {
	const float a0 = PM_FD2R(0.f), a1 = PM_FD2R(180.f), b0 = PM_FD2R(90.f), b1 = PM_FD2R(180.f);
	const float p0 = a0 + (a1 - a0) * u1;
	const float p1 = b0 + (b1 - b0) * u2;
	pm_dir2normal(p0, p1, nx, ny, nz);
	if (nx * vx + ny * vy + nz * vz > 0) { nx = -nx; ny = -ny; nz = -nz; }
}
// CorrectNormal, DepthMap.h:447-453 (+ axis-angle matrix, libs/Common/Rotation.inl:701-728), float
__device__ __forceinline__ void pm_correct_normal(float vx, float vy, float vz, float& nx, float& ny, float& nz) {
	const float cosAngLen = nx * vx + ny * vy + nz * vz;
	if (cosAngLen >= 0) {
		const float nv = pm_sqrtf(vx * vx + vy * vy + vz * vz);
		const float phi = pm_minf((pm_acosf(pm_clampf(cosAngLen / nv, -1.f, 1.f)) - PM_FD2R(90.f)) * 1.01f, -0.001f);
		float a0 = ny * vz - nz * vy, a1 = nz * vx - nx * vz, a2 = nx * vy - ny * vx;
		const float an = pm_sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
		const float ia = 1.f / an;
		a0 *= ia; a1 *= ia; a2 *= ia;
		float s, c; pm_sincosf(phi, &s, &c);
		const float O[9] = {0.f, -a2, a1, a2, 0.f, -a0, -a1, a0, 0.f};
		float Rm[9];
#pragma unroll
		for (int i = 0; i < 3; ++i)
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				float t = 0.f;
#pragma unroll
				for (int k = 0; k < 3; ++k) t += O[i * 3 + k] * O[k * 3 + j];
				Rm[i * 3 + j] = ((i == j ? 1.f : 0.f) + O[i * 3 + j] * s) + t * (1.f - c);
			}
		const float r0 = Rm[0] * nx + Rm[1] * ny + Rm[2] * nz;
		const float r1 = Rm[3] * nx + Rm[4] * ny + Rm[5] * nz;
		const float r2 = Rm[6] * nx + Rm[7] * ny + Rm[8] * nz;
		nx = r0; ny = r1; nz = r2;
	}
}

// Cross-lane moves inside a pixel's group on the VALU's DPP network (quad permutes, mirrors inside 8 and 16 lanes) instead of ds_bpermute
// through the LDS crossbar (__shfl*): full-rate instructions, no LDS round trip in the per-evaluation chain.
#define PM_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), (ctrl), 0xf, 0xf, false))
// two smallest of the group's values (as a multiset); exact (comparisons only), so any pairing of the lanes gives the same pair.
// Pairings: lane ^ 1, lane ^ 2 (quad permutes), i <-> 7 - i, i <-> 15 - i (mirrors) -- after step k every lane holds the result of its 2^k lanes.
template <int G>
__device__ __forceinline__ void pm_group_min2(float s, float s2, float& m1, float& m2) {
	float a = s, b = s2;   // this lane's two smallest scores (s <= s2; s2 = +inf when the lane scores a single view)
#define PM_MIN2_STEP(ctrl) { const float oa = PM_DPP_F(a, ctrl), ob = PM_DPP_F(b, ctrl); \
		const float na = pm_minf(a, oa), nb = pm_minf(pm_maxf(a, oa), pm_minf(b, ob)); a = na; b = nb; }
	if (G >= 2) PM_MIN2_STEP(0xB1)    // quad_perm [1,0,3,2]
	if (G >= 4) PM_MIN2_STEP(0x4E)    // quad_perm [2,3,0,1]
	if (G >= 8) PM_MIN2_STEP(0x141)   // row_half_mirror
	if (G >= 16) PM_MIN2_STEP(0x140)  // row_mirror
#undef PM_MIN2_STEP
	m1 = a; m2 = b;
}
// value of lane k (k = 0..3) of the caller's quad
template <int K> __device__ __forceinline__ float pm_quad_bcast(float v) { return PM_DPP_F(v, K * 0x55); }

// ScorePixelImage for this lane's source view, DepthMap.cpp:465-564.
// sf[]: the (view-independent) smoothness factors of the up-to-4 close neighbours, in insertion
// order; exactly 1.f for a neighbour that does not exist or does not take part (DepthMap.cpp:524-533).
// SKEW selects the image layout the 100 bilinear taps read: the sweep kernel walks an anti-diagonal, so
// the footprints of the pixels of one wave lie along an anti-diagonal of the source image too; in the
// anti-diagonal-major copy those texels are contiguous (one or two 128-B lines per view instead of one
// line per pixel), which is what the vector L1 / texture-address unit is bound by here.  Same values.
// ComputeHomographyMatrix, DepthMap.h:414-423: (Hl + Hm * (n^T / (n.X0 * depth))) * Hr in double, cast to float
// hlm: the view's Hl (9) and Hm (3), contiguous as in PMSrcView's hot block -- in HBM (init kernel) or in the wave's LDS copy (sweep kernel)
__device__ __forceinline__ void pm_homography(const double* hlm, const PMTask& t, double X0x, double X0y,
		float depth, float nx, float ny, float nz, float* H) {
	// the twelve matrix entries are requested first, together: one round trip (overlapping the division below) instead of one per row
#ifdef PM_PROBE_F32_HOMOGRAPHY
	{
		float hl[9], hm[3];
		for (int i = 0; i < 9; ++i) hl[i] = (float)hlm[i];
		for (int i = 0; i < 3; ++i) hm[i] = (float)hlm[9 + i];
		const float ndxf = (nx * (float)X0x + ny * (float)X0y) + nz, invf = __builtin_amdgcn_rcpf(ndxf * depth);
		const float r0 = nx * invf, r1 = ny * invf, r2 = nz * invf;
		for (int i = 0; i < 3; ++i) {
			const float m0 = hl[i * 3] + hm[i] * r0, m1 = hl[i * 3 + 1] + hm[i] * r1, m2 = hl[i * 3 + 2] + hm[i] * r2;
			H[i * 3] = m0 * (float)t.Hr[0]; H[i * 3 + 1] = m1 * (float)t.Hr[4]; H[i * 3 + 2] = (m0 * (float)t.Hr[2] + m1 * (float)t.Hr[5]) + m2 * (float)t.Hr[8];
		}
		return;
	}
#endif
	double Hl[9], Hm[3];
#pragma unroll
	for (int i = 0; i < 9; ++i) Hl[i] = hlm[i];
#pragma unroll
	for (int i = 0; i < 3; ++i) Hm[i] = hlm[9 + i];
	const double n0 = (double)nx, n1 = (double)ny, n2 = (double)nz;
	const double ndx = (n0 * X0x + n1 * X0y) + n2;
	const double den = ndx * (double)depth;
	const double inv = (den == 0.0) ? 1e+14 : 1.0 / den; // INVERT, Types.h:1234
	const double r0 = n0 * inv, r1 = n1 * inv, r2 = n2 * inv;
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const double hm = Hm[i];
		const double m0 = Hl[i * 3 + 0] + hm * r0;
		const double m1 = Hl[i * 3 + 1] + hm * r1;
		const double m2 = Hl[i * 3 + 2] + hm * r2;
		if (t.hrUpper) {
			// (m0*Hr0j + m1*Hr1j) + m2*Hr2j with the structurally-zero entries of K^-1 dropped: x*0 == 0 and 0+y == y exactly
			H[i * 3 + 0] = (float)(m0 * t.Hr[0]);
			H[i * 3 + 1] = (float)(m1 * t.Hr[4]);
			H[i * 3 + 2] = (float)((m0 * t.Hr[2] + m1 * t.Hr[5]) + m2 * t.Hr[8]);
		} else {
#pragma unroll
			for (int j = 0; j < 3; ++j)
				H[i * 3 + j] = (float)((m0 * t.Hr[j] + m1 * t.Hr[3 + j]) + m2 * t.Hr[6 + j]);
		}
	}
}

// LDS source tiles of the sweep kernel: per (wave, source view) a PM_TR x TC window of the anti-diagonal-major
// image, placed around the footprints of the wave's pixels under their current planes.  TC = pixels per wave + 16.
#ifndef PM_TR
#define PM_TR 20        // window rows (anti-diagonals): 19 are needed (patch 17 + bilinear 2), the rest is slack for perturbed planes
#endif
#ifndef PM_TCX
#define PM_TCX 10       // window columns = pixels per wave + PM_TCX: PPW + 9 are needed (10: 13.3 KB of windows + weights per wave, measured +5 % over 12)
#endif
#define PM_TILE_PAD 4   // per-view stride = PM_TR*TC + 4 floats: staggers the views over the LDS banks
#ifndef PM_XCD_REMAP
#define PM_XCD_REMAP 1  // sweep kernel: contiguous (view, chunk) ranges per XCD (see the kernel)
#endif
// Three switches of an experiment that did not pay (profiles/r02_variants_call7_ilp_block.log: all on 30.8, all off 32.1 Mpix/s): writing the
// per-hypothesis chains that do not depend on each other as one branch-free block so that the scheduler can interleave them.  Off by default.
// Prepared for round 3, not yet timed (default off; bit-exact under the emulator): the smoothness-factor chain (13 % of a step by the timing probes) is
// evaluated inside pm_score_view, in the same basic block as the first tap row, whose ~220 independent instructions can cover its latency; every lane
// of a pixel that evaluates a hypothesis takes part (lanes without a source view run the row on zeros and are flagged), so the broadcast of the
// factors stays inside fully active quads.  One source view per lane and >= 4 lanes per pixel only.
#ifndef PM_SMOOTH_IN_ROW0
#define PM_SMOOTH_IN_ROW0 0
#endif
// Also prepared for round 3 (default off, bit-exact under the emulator, not yet timed): a tap row that fails the LDS attempt is first retried with the same
// optimistic code reading its 20 texels from the image in HBM (~240 instructions), and only a row whose positions are not exact goes through
// pm_tap_row_global (~500).  The redone rows are 11.7 % of a step by the timing probe; a version that kept the first attempt's positions alive
// instead of recomputing them lost 3.3 % to register pressure.
#ifndef PM_GLOBAL_FAST_ROW
#define PM_GLOBAL_FAST_ROW 0
#endif
#ifndef PM_ILP_HOMOGRAPHY
#define PM_ILP_HOMOGRAPHY 0   // the lane's homography computed next to the smoothness factors instead of inside pm_score_view
#endif
#ifndef PM_ILP_PHILOX
#define PM_ILP_PHILOX 0       // the random draw of the next refinement iteration computed one evaluation ahead
#endif
#ifndef PM_ILP_SMOOTH
#define PM_ILP_SMOOTH 0       // smoothness factors computed by every lane without a branch (0: only by the lanes that own a close neighbour)
#endif
// TIMING PROBES (never defined in the product build): each removes or cheapens one part of an evaluation so that its cost inside the real, fully
// loaded kernel can be read off the benchmark -- the counter passes of rocprofv3 do not work on this pool.  The maps such a build produces are NOT the
// reference's (the scores change), only the time is meaningful; tools/build_variants.py + tools/tune.py run them (profiles/r02_probe_variants.log).
//   PM_PROBE_NO_SMOOTH      smoothness factors = 1 (no exp / acos / sqrt / division chain)
//   PM_PROBE_F32_HOMOGRAPHY homography in float (no f64 arithmetic, no f64 division)
//   PM_PROBE_CHEAP_DRAW     a three-multiply hash instead of the ten Philox rounds
//   PM_PROBE_FAST_EPILOGUE  v_rcp / v_rsq approximations instead of the correctly rounded division and square root of the score epilogue
//   PM_PROBE_NO_GEO_SAMPLES geometric term = its constant 4 (no dependent depth-map loads, no divisions)
//   PM_PROBE_NO_TAPS        the 25 taps contribute constants (no divisions, LDS reads, bilinear weights)
//   PM_PROBE_NO_STAGING     the source windows are not loaded (the taps read whatever the LDS holds)
//   PM_PROBE_NO_FALLBACK    a tap row that fails its exactness test is not redone through global memory
//   PM_PROBE_NO_WEIGHTS     the 25 bilateral patch weights are constants (no exp, no patch texel loads)
//   PM_PROBE_CHEAP_TRIG     sin / cos / atan2 / acos of the hypothesis construction replaced by two-instruction stand-ins
#ifndef PM_WIDE_MINWAVES
#define PM_WIDE_MINWAVES 3   // waves per SIMD the one-wave-per-pixel kernel is compiled for
#endif
#ifndef PM_WIDE_TILES
#define PM_WIDE_TILES 0   // one-wave-per-pixel kernel: LDS windows (1) or window-less quad-image tap rows (0; one depth map 1.08 -> 0.80 s, profiles/r03_small_batches_call7.log)
#endif
#ifndef PM_WINBATCH
#define PM_WINBATCH 4   // source windows whose global loads are in flight together when a visit stages its windows
#endif

// One tap row (5 taps) of ScorePixelImage through global loads, with the reference's per-tap tests; any layout.  X = position of the row's first tap.
// The reference returns thRobust at the first tap that leaves the image (DepthMap.cpp:484-485).  Here a tap outside only raises a flag and its
// address is clamped, so there is no branch between taps: the 20 loads of the row are issued back to back and the sums of a flagged hypothesis are
// simply discarded -- identical result, no load ever depends on a previous load.
template <bool SKEW>
__device__ __forceinline__ void pm_tap_row_global(const pm_gcf img, int sw, int sh, float h0, float h3, float h6, float X0, float X1, float X2,
		const float2* wrow, float& sum, float& sumSq, float& num, bool& oob)
{
	const int lxMax = sw - 2, lyMax = sh - 2;
	float fxs[5], fys[5];
	unsigned offs[5];   // texel offsets fit 32 bits (an image or its skewed copy is < 2^32 floats)
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		float ptx, pty;
		pm_div2(X0, X1, X2, &ptx, &pty); // == X0 / X2, X1 / X2 (TPoint2(Point3), Types.h:1291)
		oob = oob || !pm_inside1(ptx, pty, sw, sh);
		// TImage::sample, libs/Common/Types.inl:2273-2281
		int lx = (int)ptx, ly = (int)pty;
		fxs[j] = ptx - (float)lx; fys[j] = pty - (float)ly;
		lx = min(max(lx, 0), lxMax); ly = min(max(ly, 0), lyMax);
		offs[j] = SKEW ? (unsigned)(lx + ly) * (unsigned)sh + (unsigned)ly : (unsigned)ly * (unsigned)sw + (unsigned)lx;
		X0 += h0; X1 += h3; X2 += h6;
	}
	float v00[5], v01[5], v10[5], v11[5];
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		const pm_gcf p = img + offs[j];
		// texels (lx,ly), (lx+1,ly), (lx,ly+1), (lx+1,ly+1) sit at skew (s,t), (s+1,t), (s+1,t+1), (s+2,t+1)
		if (SKEW) { v00[j] = p[0]; v01[j] = p[sh]; v10[j] = p[sh + 1]; v11[j] = p[2 * sh + 1]; }
		else { v00[j] = p[0]; v01[j] = p[1]; v10[j] = p[sw]; v11[j] = p[sw + 1]; }
	}
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		const float fx = fxs[j], fx1 = 1.f - fx, fy = fys[j], fy1 = 1.f - fy;
		const float v = (v00[j] * fx1 + v01[j] * fx) * fy1 + (v10[j] * fx1 + v11[j] * fx) * fy;
		const float2 pw = wrow[j];
		const float vw = v * pw.x;
		sum += vw;
		sumSq += v * vw;
		num += v * pw.y;
	}
}

// The same row served from the wave's LDS window of the anti-diagonal-major source image, optimistically: no test between the taps.  The divisions
// use the unguarded reciprocal refinement, the texel indices are only clamped into the window; while it goes the row tracks the extremes of z, of
// the projected positions and of the window row index, and one comparison set at the end says whether every tap (a) had 2^-40 <= z <= 2^40 (with
// `sane`: |x|, |y| < 1e18 -- then the quotients are the correctly rounded ones and nothing is NaN), (b) was inside the image (isInsideWithBorder<1>)
// and (c) inside the window.  If so the three running sums are exactly what pm_tap_row_global computes and the call returns true; with (a) but a tap
// outside the image the hypothesis is flagged (`oob`) and the row is done as well; otherwise the sums are left untouched and the caller redoes the
// row through global loads.  ~40 VALU instructions per tap instead of ~100.
template <int TC, bool FROM_IMAGE = false>
__device__ __forceinline__ bool pm_tap_row_lds(const float* tile, int ts0, int tt0, int sw, int sh, bool sane, float h0, float h3, float h6,
		float X0, float X1, float X2, const float2* wrow, float& sum, float& sumSq, float& num, bool& oob, const pm_gcf4 imgQ = (pm_gcf4)nullptr)
{
	constexpr int MAXI = PM_TR * TC - 2 * TC - 2;   // idx + 2*TC + 1 stays inside the window
	const int cidx = -(ts0 * TC + tt0);
	float fxs[5], fys[5];
	const float* q[5];
	unsigned goff[5];   // FROM_IMAGE: texel offsets in the anti-diagonal-major image, clamped into it (a row with a tap outside is not committed)
	float zlo = X2, zhi = X2, pxlo = PM_INF, pxhi = -PM_INF, pylo = PM_INF, pyhi = -PM_INF;
	int slo = 0x7fffffff, shi = (int)0x80000000;
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		float ptx, pty;
		pm_div2_inrange(X0, X1, X2, &ptx, &pty);
		zlo = pm_fminf(zlo, X2); zhi = pm_fmaxf(zhi, X2);
		pxlo = pm_fminf(pxlo, ptx); pxhi = pm_fmaxf(pxhi, ptx); pylo = pm_fminf(pylo, pty); pyhi = pm_fmaxf(pyhi, pty);
		const int lx = (int)ptx, ly = (int)pty;
		fxs[j] = pm_fract_pos(ptx); fys[j] = pm_fract_pos(pty);   // == ptx - (float)lx for the positions the row is accepted with (>= 1)
		const int sk = lx + ly;
		slo = min(slo, sk); shi = max(shi, sk);
		if (FROM_IMAGE) {
			const int lxc = min(max(lx, 0), sw - 2), lyc = min(max(ly, 0), sh - 2);
			goff[j] = (unsigned)(lxc + lyc) * (unsigned)sh + (unsigned)lyc;
		} else {
			const int idx = pm_mul24(sk, TC) + (ly + cidx);   // 24-bit multiply: full rate (a 32-bit one is a quarter-rate v_mad_u64_u32); garbage only where the row fails anyway
			q[j] = tile + min((unsigned)idx, (unsigned)MAXI);
		}
		X0 += h0; X1 += h3; X2 += h6;
	}
	float s0 = sum, s1 = sumSq, s2 = num;
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		float v00, v01, v10, v11;
		if (FROM_IMAGE) pm_load4(imgQ, goff[j], v00, v01, v10, v11);
		else { v00 = q[j][0]; v01 = q[j][TC]; v10 = q[j][TC + 1]; v11 = q[j][2 * TC + 1]; }
		const float fx = fxs[j], fx1 = 1.f - fx, fy = fys[j], fy1 = 1.f - fy;
		const float v = (v00 * fx1 + v01 * fx) * fy1 + (v10 * fx1 + v11 * fx) * fy;
		const float2 pw = wrow[j];
		const float vw = v * pw.x;
		s0 += vw;
		s1 += v * vw;
		s2 += v * pw.y;
	}
	// With 2^-40 <= z <= 2^40 and `sane` every quotient above is the correctly rounded one, so the image test is exact: a tap outside the image
	// flags the hypothesis just as pm_tap_row_global would (its sums are then never used) and the row is done -- source views that do not see the
	// pixel at all are the common case of a failed row and must not cost a second pass through global memory.
	const bool exact = sane && zlo >= 9.094947e-13f && zhi <= 1.0995116e12f;
	const bool inImage = pxlo >= 1.f && pylo >= 1.f && pxhi <= (float)(sw - 2) && pyhi <= (float)(sh - 2);
	// (int)pty in [tt0, tt0 + TC - 2]  <=>  tt0 <= pty < tt0 + TC - 1 once pty >= 1
	const bool inWindow = pylo >= (float)tt0 && pyhi < (float)(tt0 + TC - 1) && slo >= ts0 && shi <= ts0 + PM_TR - 3;
	if (exact && !inImage) { oob = true; return true; }
	const bool ok = exact && (FROM_IMAGE || inWindow);
	if (ok) { sum = s0; sumSq = s1; num = s2; }
	return ok;
}

// The window-less optimistic rows (pm_tap_row_lds<.., FROM_IMAGE>) as a software pipeline over the five rows of a patch.  A wave walks ~110 tap rows
// per visit and every row began with five scattered 16-byte loads whose latency (L2 / HBM: the quad images of a batch are far larger than the caches)
// nothing else in the wave could cover -- the counters showed waves ~40 % VALU-active and a diagonal's launch lasting as long as one wave's chain of
// such round trips.  Here the texel addresses and loads of row i+1 are issued before row i is consumed, and a row has no control flow: its sums are
// committed unconditionally, `oob` / `redo` only record what the range checks found.  A patch with a row that was not exact (z outside
// [2^-40, 2^40] or !sane: rare) is redone as a whole by the caller through pm_tap_row_global -- the fast rows produce exactly its sums, so redoing them
// changes nothing.  PM_ROW_PIPELINE 0 restores the row-at-a-time loop.
#ifndef PM_ROW_PIPELINE
#define PM_ROW_PIPELINE 0
#endif
struct PMRowTaps { float ptx[5], pty[5], zlo, zhi; unsigned goff[5]; };
struct PMRowQuads { float v00[5], v01[5], v10[5], v11[5]; };
__device__ __forceinline__ void pm_row_prep(PMRowTaps& r, int sw, int sh, float h0, float h3, float h6, float X0, float X1, float X2) {
	r.zlo = X2; r.zhi = X2;
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		pm_div2_inrange(X0, X1, X2, &r.ptx[j], &r.pty[j]);
		r.zlo = pm_fminf(r.zlo, X2); r.zhi = pm_fmaxf(r.zhi, X2);
		const int lx = (int)r.ptx[j], ly = (int)r.pty[j];
		const int lxc = min(max(lx, 0), sw - 2), lyc = min(max(ly, 0), sh - 2);
		r.goff[j] = (unsigned)(lxc + lyc) * (unsigned)sh + (unsigned)lyc;
		X0 += h0; X1 += h3; X2 += h6;
	}
}
__device__ __forceinline__ void pm_row_load(PMRowQuads& q, const PMRowTaps& r, const pm_gcf4 imgQ) {
#pragma unroll
	for (int j = 0; j < 5; ++j) pm_load4(imgQ, r.goff[j], q.v00[j], q.v01[j], q.v10[j], q.v11[j]);
}
__device__ __forceinline__ void pm_row_consume(const PMRowTaps& r, const PMRowQuads& q, int sw, int sh, bool sane, const float2* wrow,
		float& sum, float& sumSq, float& num, bool& oob, bool& redo) {
	float pxlo = PM_INF, pxhi = -PM_INF, pylo = PM_INF, pyhi = -PM_INF;
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		pxlo = pm_fminf(pxlo, r.ptx[j]); pxhi = pm_fmaxf(pxhi, r.ptx[j]); pylo = pm_fminf(pylo, r.pty[j]); pyhi = pm_fmaxf(pyhi, r.pty[j]);
		const float fx = pm_fract_pos(r.ptx[j]), fx1 = 1.f - fx, fy = pm_fract_pos(r.pty[j]), fy1 = 1.f - fy;
		const float v = (q.v00[j] * fx1 + q.v01[j] * fx) * fy1 + (q.v10[j] * fx1 + q.v11[j] * fx) * fy;
		const float2 pw = wrow[j];
		const float vw = v * pw.x;
		sum += vw;
		sumSq += v * vw;
		num += v * pw.y;
	}
	const bool exact = sane && r.zlo >= 9.094947e-13f && r.zhi <= 1.0995116e12f;
	const bool inImage = pxlo >= 1.f && pylo >= 1.f && pxhi <= (float)(sw - 2) && pyhi <= (float)(sh - 2);
	oob = oob || (exact && !inImage);
	redo = redo || !exact;
}

// PF: the pixel's low-resolution prior and its blend factor exp(normSq0 * sigma) sit in the spare 26th entry of the pixel's weight row in LDS
// (written once per visit by the sweep kernel) instead of in two registers that are live across the whole hypothesis loop
// what pm_score_view needs to evaluate the lane's smoothness factor itself (PM_SMOOTH_IN_ROW0)
struct PMSmoothIn { float qX0, qX1, qX2, qn0, qn1, qn2, vx, vy, vz; bool on; };

template <bool GEO, bool SKEW, int TC, bool PF = false, bool SIN = false>
__device__ __forceinline__ float pm_score_view(const PMSrcView& s, const PMTask& t, const PMKParams& kp,
		int x, int y, double X0x, double X0y, float normSq0, float sumW, const float2* wts,
		float depth, float nx, float ny, float nz,
		float sf0, float sf1, float sf2, float sf3, float prior,
		const float* tile, int ts0, int tt0, const double* hot, const double* geoTab, const float* Hpre, const PMSmoothIn* sin = nullptr, bool viewOk = true PM_PROF_ARG)
{
	// hot / geoTab: the hot and geometric blocks of `s` (PMSrcView), in HBM (init kernel) or in the wave's LDS copy (sweep kernel); the image
	// size travels with the homography entries
	const int sw = ((const int*)(hot + 12))[0], sh = ((const int*)(hot + 12))[1];
	float H[9];
	if (Hpre) {   // computed by the caller next to the other per-hypothesis chains (instruction-level parallelism), see pm_sweep_kernel
#pragma unroll
		for (int i = 0; i < 9; ++i) H[i] = Hpre[i];
	} else pm_homography(hot, t, X0x, X0y, depth, nx, ny, nz, H);
	PM_TICK(3);
	const float px = (float)(x - PM_HW), py = (float)(y - PM_HW);
	const float X0 = H[0] * px + H[1] * py + H[2];
	const float X1 = H[3] * px + H[4] * py + H[5];
	const float X2 = H[6] * px + H[7] * py + H[8];
	float bX0 = X0, bX1 = X1, bX2 = X2;
#pragma unroll
	for (int i = 0; i < 9; ++i) H[i] *= 2.f; // nSizeStep
	float sum = 0.f, sumSq = 0.f, num = 0.f;
	bool oob = !viewOk;   // a lane without a source view (SIN only) is flagged from the start: its rows are never redone, its score is discarded
	// |x|, |y| < 1e18 and |z| < 2^40 for every tap of the patch, from the first tap and the step sizes (8 steps along either axis at most)
	// FASTG: no LDS windows (PM_USE_TILES = 0); the optimistic row reads its texels straight from the anti-diagonal-major image (vector L1)
	constexpr bool FASTG = SKEW && TC == 0;
	bool sane = false;
	if (TC > 0 || FASTG)
		sane = pm_fabsf(X0) + 8.f * (pm_fabsf(H[0]) + pm_fabsf(H[1])) < 5e17f && pm_fabsf(X1) + 8.f * (pm_fabsf(H[3]) + pm_fabsf(H[4])) < 5e17f
			&& pm_fabsf(X2) + 8.f * (pm_fabsf(H[6]) + pm_fabsf(H[7])) < 5e11f;
	int iFirst = 0;
	if (SIN && TC > 0) {
		// the lane's smoothness factor (DepthMap.cpp:524-533), branch-free, then the first tap row: one basic block for the scheduler to interleave
		const float planeD = -depth * (nx * sin->vx + ny * sin->vy + nz * sin->vz); // InitPlane, DepthMap.cpp:963-971
		const float dist = (nx * sin->qX0 + (ny * sin->qX1 + nz * sin->qX2)) + planeD; // Planef::Distance, Eigen 3-dot order
		const float r = dist / depth;
		const float factorDepth = pm_expf((r * r) * kp.smoothSigmaDepth);
		const float ca = pm_clampf((nx * sin->qn0 + ny * sin->qn1 + nz * sin->qn2) /
			pm_sqrtf((nx * nx + ny * ny + nz * nz) * (sin->qn0 * sin->qn0 + sin->qn1 * sin->qn1 + sin->qn2 * sin->qn2)), -1.f, 1.f);
		const float ac = pm_acosf(ca);
		const float factorNormal = pm_expf((ac * ac) * kp.smoothSigmaNormal);
		const float f = (1.f - kp.smoothBonusDepth * factorDepth) * (1.f - kp.smoothBonusNormal * factorNormal);
		const float myF = sin->on ? f : 1.f;
		const bool done0 = pm_tap_row_lds<TC>(tile, ts0, tt0, sw, sh, sane, H[0], H[3], H[6], bX0, bX1, bX2, wts, sum, sumSq, num, oob) || oob;
		sf0 = pm_quad_bcast<0>(myF); sf1 = pm_quad_bcast<1>(myF); sf2 = pm_quad_bcast<2>(myF); sf3 = pm_quad_bcast<3>(myF);
		if (!done0) pm_tap_row_global<SKEW>(pm_glob(SKEW ? s.imgS : s.img), sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, wts, sum, sumSq, num, oob);
		bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
		iFirst = 1;
	}
#ifdef PM_PROBE_NO_TAPS
	if (TC > 0) { sum = 0.31f * sumW; sumSq = 0.11f * sumW + 0.01f * H[2]; num = 0.004f * H[5]; }
	else
#endif
	if (FASTG && PM_ROW_PIPELINE) {
		const pm_gcf4 imgQ = pm_glob4(s.imgQ);
		const float rX0 = bX0, rX1 = bX1, rX2 = bX2;
		const bool oobIn = oob;
		bool redo = false;
		PMRowTaps ta, tb; PMRowQuads qa, qb;
		pm_row_prep(ta, sw, sh, H[0], H[3], H[6], bX0, bX1, bX2);
		pm_row_load(qa, ta, imgQ);
		// rows (0,1), (2,3) as one loop body with the A / B register sets swapping roles (no copies), then row 4
#pragma unroll 1
		for (int i = 0; i < 4; i += 2) {
			bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
			pm_row_prep(tb, sw, sh, H[0], H[3], H[6], bX0, bX1, bX2); pm_row_load(qb, tb, imgQ);
			pm_row_consume(ta, qa, sw, sh, sane, wts + i * 5, sum, sumSq, num, oob, redo);
			bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
			pm_row_prep(ta, sw, sh, H[0], H[3], H[6], bX0, bX1, bX2); pm_row_load(qa, ta, imgQ);
			pm_row_consume(tb, qb, sw, sh, sane, wts + (i + 1) * 5, sum, sumSq, num, oob, redo);
		}
		pm_row_consume(ta, qa, sw, sh, sane, wts + 20, sum, sumSq, num, oob, redo);
#if !defined(__HIP_DEVICE_COMPILE__) && defined(PM_DEBUG_REDO)
		{ static unsigned long long c[3]; static bool reg = false; if (!reg) { reg = true; atexit([] { fprintf(stderr, "FASTG evaluations %llu, redo %llu, redo && !oob %llu\n", c[0], c[1], c[2]); }); } c[0]++; if (redo) c[1]++; if (redo && !oob) c[2]++; }
#endif
		if (redo && !oob) {   // a row outside the fast divisions' range: the whole patch through the guarded path (same sums where the fast rows were valid)
			sum = 0.f; sumSq = 0.f; num = 0.f; oob = oobIn;
			bX0 = rX0; bX1 = rX1; bX2 = rX2;
#pragma unroll 1
			for (int i = 0; i < 5; ++i) {
				pm_tap_row_global<SKEW>(pm_glob(SKEW ? s.imgS : s.img), sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, wts + i * 5, sum, sumSq, num, oob);
				bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
			}
		}
	} else
#pragma unroll 1
	for (int i = iFirst; i < 5; ++i) {
		bool done = false;
		if (TC > 0) done = pm_tap_row_lds<TC>(tile, ts0, tt0, sw, sh, sane, H[0], H[3], H[6], bX0, bX1, bX2, wts + i * 5, sum, sumSq, num, oob) || oob;
#ifdef PM_PROFILE
		if (TC > 0) { PM_COUNT(10, __popcll(__ballot(done))); PM_COUNT(11, __popcll(__ballot(true))); PM_COUNT(7, __all(done) ? 1 : 0); }
#endif
#ifdef PM_PROBE_NO_FALLBACK
		if (TC > 0) done = true;
#endif
		if (((PM_GLOBAL_FAST_ROW && TC > 0 && SKEW) || FASTG) && !done)
			done = pm_tap_row_lds<(TC > 0 ? TC : 1), true>(tile, ts0, tt0, sw, sh, sane, H[0], H[3], H[6], bX0, bX1, bX2, wts + i * 5, sum, sumSq, num, oob, pm_glob4(s.imgQ)) || oob;
		if (!done) pm_tap_row_global<SKEW>(pm_glob(SKEW ? s.imgS : s.img), sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, wts + i * 5, sum, sumSq, num, oob);
		bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
	}
	PM_TICK(4);
	if (oob) return kp.thRobust;
#ifdef PM_PROBE_FAST_EPILOGUE
	const float normSq1 = sumSq - (sum * sum) * __builtin_amdgcn_rcpf(sumW);
#else
	const float normSq1 = sumSq - (sum * sum) / sumW;
#endif
	const float nrmSq = normSq0 * normSq1;
	if (nrmSq <= 1e-16f) return kp.thRobust;
#ifdef PM_PROBE_FAST_EPILOGUE
	const float ncc = pm_clampf(num * __builtin_amdgcn_rsqf(nrmSq), -1.f, 1.f);
#else
	const float ncc = pm_clampf(num / pm_sqrtf(nrmSq), -1.f, 1.f);
#endif
	float score = 1.f - ncc;
	// (a factor of a neighbour that does not take part is exactly 1.f, and x * 1.f == x: no test needed, DepthMap.cpp:524-533)
	score *= sf0; score *= sf1; score *= sf2; score *= sf3;
	if (GEO) {
		// geometric consistency, DepthMap.cpp:535-551
		// the source view's depth-map pointer and its four transforms are requested together (they were five dependent round trips per evaluation)
		const float* gt = (const float*)geoTab;
		const float* sdepth = *(const float* const*)(geoTab + 12);
		const int dw = ((const int*)(geoTab + 13))[0], dh = ((const int*)(geoTab + 13))[1];
		float Tl[9], Tm[3], Tr[9], Tn[3];
#pragma unroll
		for (int i = 0; i < 9; ++i) { Tl[i] = gt[i]; Tr[i] = gt[12 + i]; }
#pragma unroll
		for (int i = 0; i < 3; ++i) { Tm[i] = gt[9 + i]; Tn[i] = gt[21 + i]; }
#pragma unroll
		for (int i = 0; i < 9; ++i) PM_OPAQUE(Tr[i]);
#pragma unroll
		for (int i = 0; i < 3; ++i) PM_OPAQUE(Tn[i]);
#ifdef PM_PROBE_NO_GEO_SAMPLES
		if (sdepth != nullptr) score += kp.geoWeight * 4.f;
		if (false) {
#else
		if (sdepth != nullptr) {
#endif
			float consistency = 4.f;
			const float Xc0 = (float)X0x * depth, Xc1 = (float)X0y * depth, Xc2 = depth;
			const float Y0 = (Tl[0] * Xc0 + Tl[1] * Xc1 + Tl[2] * Xc2) + Tm[0];
			const float Y1 = (Tl[3] * Xc0 + Tl[4] * Xc1 + Tl[5] * Xc2) + Tm[1];
			const float Y2 = (Tl[6] * Xc0 + Tl[7] * Xc1 + Tl[8] * Xc2) + Tm[2];
			if (Y2 > 0) {
				const float x1x = Y0 / Y2, x1y = Y1 / Y2;
				if (pm_inside1(x1x, x1y, dw, dh)) {
					// TImage::sample with validity functor, Types.inl:2299-2314; IsDepthSimilar(z,d,0.03), Util.inl:798-809
					const int lx = (int)x1x, ly = (int)x1y;
					const float fx = x1x - (float)lx, fx1 = 1.f - fx;
					const float fy = x1y - (float)ly, fy1 = 1.f - fy;
					const pm_gcf p = pm_glob(sdepth) + (size_t)ly * dw + lx;
					const float x0y0 = p[0], x1y0 = p[1], x0y1 = p[dw], x1y1 = p[dw + 1];
					const bool b00 = pm_fabsf(Y2 - x0y0) / Y2 < 0.03f, b10 = pm_fabsf(Y2 - x1y0) / Y2 < 0.03f;
					const bool b01 = pm_fabsf(Y2 - x0y1) / Y2 < 0.03f, b11 = pm_fabsf(Y2 - x1y1) / Y2 < 0.03f;
					if (b00 || b10 || b01 || b11) {
						const float depth1 =
							fy1 * (fx1 * (b00 ? x0y0 : (b10 ? x1y0 : (b01 ? x0y1 : x1y1))) + fx * (b10 ? x1y0 : (b00 ? x0y0 : (b11 ? x1y1 : x0y1)))) +
							fy  * (fx1 * (b01 ? x0y1 : (b11 ? x1y1 : (b00 ? x0y0 : x1y0))) + fx * (b11 ? x1y1 : (b01 ? x0y1 : (b10 ? x1y0 : x0y0))));
						const float Xd0 = x1x * depth1, Xd1 = x1y * depth1, Xd2 = depth1;
						const float B0 = (Tr[0] * Xd0 + Tr[1] * Xd1 + Tr[2] * Xd2) + Tn[0];
						const float B1 = (Tr[3] * Xd0 + Tr[4] * Xd1 + Tr[5] * Xd2) + Tn[1];
						const float B2 = (Tr[6] * Xd0 + Tr[7] * Xd1 + Tr[8] * Xd2) + Tn[2];
						const float xbx = B0 / B2, xby = B1 / B2;
						const float dx = (float)x - xbx, dy = (float)y - xby;
						const float dist = pm_sqrtf(dx * dx + dy * dy); // norm(Point2f) = SEACAVE::norm(TPoint2<float>): float (Types.inl:1021-1024); pinned by oracle/_ref
						consistency = pm_minf(pm_sqrtf(dist * (dist + 2.f)), consistency);
					}
				}
			}
			score += kp.geoWeight * consistency;
		}
	}
	// low-resolution prior, DepthMap.cpp:553-561
	if (PF) {
		const float2 pf = wts[PM_NT];
		float pfy = pf.y;
		PM_OPAQUE(pfy);   // both halves are read at once (the compiler otherwise splits the read and sinks one half under the test of the other)
		if (pf.x > 0) {
			const float deltaDepth = pm_minf(pm_fabsf(pf.x - depth) / pf.x, 0.5f);
			score = (1.f - pfy) * score + pfy * deltaDepth;
		}
	} else if (prior > 0) {
		const float deltaDepth = pm_minf(pm_fabsf(prior - depth) / prior, 0.5f);
		const float sigma = -1.f / (1.f * 0.02f);
		const float f = pm_expf(normSq0 * sigma);
		score = (1.f - f) * score + f * deltaDepth;
	}
	PM_TICK(5);
	return pm_minf(2.f, score);
}

// ScorePixel aggregation, DepthMap.cpp:594-611 (MINMEAN)
template <int G>
__device__ __forceinline__ float pm_aggregate(float viewScore, int nSrc, float thRobust, float viewScore2 = PM_INF) {
	float m1, m2;
	pm_group_min2<G>(viewScore, viewScore2, m1, m2);
	if (nSrc <= 1) return m1;
	if (m2 >= thRobust) return m1;
	return (m1 + m2) / 2.f;
}

// The same for view-major lanes (lane = v * (64 / G) + g: the lanes of a pixel are 64 / G apart): butterfly over the lane index bits above the pixel index
template <int G>
__device__ __forceinline__ float pm_aggregate_vm(float viewScore, int nSrc, float thRobust, float viewScore2) {
	float a = viewScore, b = viewScore2;
#pragma unroll
	for (int m = 64 / G; m < 64; m <<= 1) {
		const float oa = __shfl_xor(a, m, 64), ob = __shfl_xor(b, m, 64);
		const float na = pm_minf(a, oa), nb = pm_minf(pm_maxf(a, oa), pm_minf(b, ob)); a = na; b = nb;
	}
	if (nSrc <= 1) return a;
	if (b >= thRobust) return a;
	return (a + b) / 2.f;
}

// FillPixelPatch, DepthMap.cpp:422-462: cooperative weights into LDS; returns normSq0, sumW.
// Must be called by every thread of the workgroup (contains __syncthreads()).
template <int G, bool SKEW>
__device__ __forceinline__ void pm_fill_patch(const PMTask& t, bool inb, int x, int y, int v, float2* wts, float& normSq0, float& sumW) {
	const float sigmaColor = -1.f / (2.f * (0.1f * 0.1f));
	const float sigmaSpatial = -1.f / (2.f * 9.f);
	if (inb) {
		const pm_gcf refS = pm_glob(t.refS), ref = pm_glob(t.ref);
		const float colCenter = SKEW ? refS[(size_t)(x + y) * t.h + y] : ref[(size_t)y * t.w + x];
		// the texels of this lane's taps are requested together (one memory round trip at the head of the visit, not one per tap)
		constexpr int NK = (PM_NT + G - 1) / G;
		float Is[NK];
#pragma unroll
		for (int q = 0; q < NK; ++q) {
			const int k = v + q * G, kk = k < PM_NT ? k : v % PM_NT;   // (a lane without a tap re-reads one that exists: no address outside the patch)
			const int i = (kk / 5) * 2 - PM_HW, j = (kk % 5) * 2 - PM_HW;
			Is[q] = SKEW ? refS[(size_t)(x + j + y + i) * t.h + (y + i)] : ref[(size_t)(y + i) * t.w + (x + j)];
		}
#pragma unroll
		for (int q = 0; q < NK; ++q) {
			const int k = v + q * G;
			if (k >= PM_NT) break;
			const int i = (k / 5) * 2 - PM_HW, j = (k % 5) * 2 - PM_HW;
			const float I = Is[q];
			const float dc = I - colCenter;
			const float wColor = (dc * dc) * sigmaColor;
			const float wSpatial = (float)(j * j + i * i) * sigmaSpatial;
#ifdef PM_PROBE_NO_WEIGHTS
			wts[k] = make_float2(0.5f + 0.01f * (float)k, 0.3f + 0.02f * (float)(k % 7));
#else
			wts[k] = make_float2(pm_expf(wColor + wSpatial), I);
#endif
		}
	}
	__syncthreads();
	float tm = 0.f;
	normSq0 = 0.f; sumW = 0.f;
	if (inb) {
		float acc = 0.f;
		for (int k = 0; k < PM_NT; ++k) { const float2 p = wts[k]; acc += p.y * p.x; sumW += p.x; }
		tm = acc / sumW;
		for (int k = 0; k < PM_NT; ++k) { const float2 p = wts[k]; const float d = p.y - tm; const float tw = p.x * d; normSq0 += tw * d; }
	}
	__syncthreads();
	if (inb) {
		for (int k = v; k < PM_NT; k += G) { const float2 p = wts[k]; const float d = p.y - tm; wts[k] = make_float2(p.x, p.x * d); }
	}
	__syncthreads();
}

__device__ __forceinline__ float pm_pow2neg(unsigned i) { return pm_u2f((127u - i) << 23); } // scaleRanges[i] = 2^-i, DepthMap.cpp:359

// -------------------------------------------------------------------------------------------
// ScoreDepthMapTmp, SceneDensify.cpp:490-517: fully parallel, no neighbour dependency.
template <int G, bool GEO>
__global__ __launch_bounds__(PM_BLOCK) void pm_init_kernel(const PMTask* __restrict__ tasks, PMKParams kp, uint32_t pass) {
	constexpr int PPB = PM_BLOCK / G;
	__shared__ float2 s_w[PPB][PM_NT + 1];
	const PMTask& t = tasks[blockIdx.y];
	const int g = threadIdx.x / G, v = threadIdx.x % G;
	const int w = t.w, h = t.h;
	const long p = (long)blockIdx.x * PPB + g;
	const bool active = p < (long)w * h;
	const int x = active ? (int)(p % w) : 0, y = active ? (int)(p / w) : 0;
	const bool inb = active && x >= PM_HW && y >= PM_HW && x < w - PM_HW && y < h - PM_HW; // PreparePixelPatch
	float normSq0, sumW;
	pm_fill_patch<G, false>(t, inb, x, y, v, s_w[g], normSq0, sumW);
	if (!active) return;
	const size_t idx = (size_t)y * w + x;
	const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
	const float prior = t.prior ? pm_glob(t.prior)[idx] : 0.f;
	const bool masked = t.mask != nullptr && t.mask[idx] == 0;   // not in the reference's pixel list: depth/normal zeroed, never scored
	const bool valid = inb && !masked && !(normSq0 < kp.thMagnitudeSq && !(prior > 0));
	if (!valid) {
		if (v == 0) { gDepth[idx] = 0.f; gNormal[idx * 3] = 0.f; gNormal[idx * 3 + 1] = 0.f; gNormal[idx * 3 + 2] = 0.f; gConf[idx] = 2.f; }
		return;
	}
	const double X0x = ((double)x - t.cx) / t.fx, X0y = ((double)y - t.cy) / t.fy;
	const float vx = (float)X0x, vy = (float)X0y, vz = 1.f;
	float depth = gDepth[idx];
	float nx = gNormal[idx * 3], ny = gNormal[idx * 3 + 1], nz = gNormal[idx * 3 + 2];
	const PmPhilox4 r = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(PM_STREAM_INIT * 256), 0u, t.k0, t.k1base + pass);
	if (!pm_in_range(depth, t.dMin, t.dMax)) {
		const float rr = t.dMinSqr + (t.dMaxSqr - t.dMinSqr) * pm_u32_to_unit(r.v[0]);
		depth = rr * rr;
		pm_random_normal(pm_u32_to_unit(r.v[1]), pm_u32_to_unit(r.v[2]), vx, vy, vz, nx, ny, nz);
	} else if (nx * vx + ny * vy + nz * vz >= 0) {
		pm_random_normal(pm_u32_to_unit(r.v[1]), pm_u32_to_unit(r.v[2]), vx, vy, vz, nx, ny, nz);
	}
	float sc = PM_INF;
	PM_PROF_DECL;
	if (v < t.nSrc)
		sc = pm_score_view<GEO, false, 0>(t.src[v], t, kp, x, y, X0x, X0y, normSq0, sumW, s_w[g], depth, nx, ny, nz, 1.f, 1.f, 1.f, 1.f, prior, nullptr, 0, 0, t.src[v].Hl, (const double*)t.src[v].Tl, nullptr PM_PROF_PASS);
	const float conf = pm_aggregate<G>(sc, t.nSrc, kp.thRobust);
	if (v == 0) { gDepth[idx] = depth; gNormal[idx * 3] = nx; gNormal[idx * 3 + 1] = ny; gNormal[idx * 3 + 2] = nz; gConf[idx] = conf; }
}

// -------------------------------------------------------------------------------------------
// ProcessPixel, DepthMap.cpp:630-852, for all pixels of anti-diagonal x+y == d (x = xlo + i).
// dir 0 = LT2RB (left/top are new), 1 = RB2LT (right/bottom are new).
// VPL = source views per lane: lane v of a pixel's G lanes scores views v, v + G, ..., one after the other.  Everything that is per pixel
// (hypothesis generation, smoothness factors, the visit's set-up, accept / reject: about 70 % of a wave's time at one view per lane, measured
// with -DPM_PROFILE) is then shared by VPL times as many pixels per wave; MINMEAN does not care which lane scored which view.
template <int G, int VPL, bool GEO>
__global__ __launch_bounds__(PM_BLOCK, (!PM_USE_TILES ? PM_MINWAVES : VPL >= 4 ? 1 : VPL == 2 ? 2 : PM_MINWAVES)) void pm_sweep_kernel(const PMTask* __restrict__ tasks, PMKParams kp, int dir, int d, int xlo, int count, uint32_t pass) {
	constexpr int PPB = PM_BLOCK / G;
	constexpr int SL = (G >= 4) ? 1 : 4 / G; // smoothness slots owned per lane
	constexpr int PPW = 64 / G;               // pixels per wave
	constexpr int NV = G * VPL;               // source views the wave holds windows for
	constexpr int TC = PM_USE_TILES ? PPW + PM_TCX : 0;
	constexpr int TSTRIDE = PM_TR * TC + PM_TILE_PAD;
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	PM_PROF_DECL;
	__shared__ float2 s_w[PPB][PM_NT + 1];
	__shared__ float s_tile[PM_USE_TILES ? PM_BLOCK / 64 : 1][PM_USE_TILES ? NV * TSTRIDE : 1];
	__shared__ int2 s_org[PM_BLOCK / 64][NV];   // window origin (ts0, tt0) of every view
	// Per-view constants of the wave's source views: every hypothesis evaluation of every pixel needs the hot block of its lane's view (Hl, Hm, image
	// size; in the geometric pass also the four transforms and the depth-map pointer) -- a global round trip at the head of each evaluation when read
	// from the task.  One coalesced copy per visit puts them an LDS read away (832 B per wave at G = 8, twice that in the geometric pass).
	__shared__ double s_src[PM_BLOCK / 64][NV * NBD];
	// XCD-aware block mapping: workgroup b is observed to run on XCD b % 8 (dispatch order, x fastest), each XCD with its own 4 MB L2.  The remap
	// hands every XCD a contiguous range of (view, diagonal chunk) pairs -- the same few views launch after launch -- so the source windows of
	// neighbouring chunks and of the next diagonal are found in that XCD's L2 instead of being fetched into several of them.  Bijective for any
	// grid size; which workgroup handles which pixels does not matter for the result (the pixels of a diagonal are independent).
	unsigned vbx = blockIdx.x, vby = blockIdx.y;
	if (PM_XCD_REMAP) {
		const unsigned nbx = gridDim.x, nwg = nbx * gridDim.y, orig = blockIdx.y * nbx + blockIdx.x;
		const unsigned xcd = orig % 8u, q = nwg / 8u, r = nwg % 8u;
		const unsigned wgid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + orig / 8u;
		vby = wgid / nbx; vbx = wgid - vby * nbx;
	}
	const PMTask& t = tasks[vby];
	const int g = threadIdx.x / G, v = threadIdx.x % G;
	{
		double* dst = s_src[threadIdx.x >> 6];
		for (int i = threadIdx.x & 63; i < NV * NBD; i += 64) dst[i] = ((const double*)&t.src[i / NBD])[i % NBD];   // views >= nSrc: zeros (the task is memset), never used
	}
	const double* hotBase = s_src[threadIdx.x >> 6];   // view k's blocks at hotBase + k * NBD
	const int w = t.w, h = t.h;
	const int pi = vbx * PPB + g;
	const bool active = pi < count;
	const int x = xlo + (active ? pi : 0), y = d - x;
	const size_t idx = (size_t)y * w + x;
	const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
	// neighbour slots in insertion order (DepthMap.cpp:641-766): with sgn = -1 (LT2RB) / +1 (RB2LT):
	// slot0 (x+sgn,y), slot1 (x,y+sgn) are the already-updated propagation sources; slot2 (x-sgn,y), slot3 (x,y-sgn).
	const int sgn = dir == 0 ? -1 : 1;
	// Everything the visit reads of its own and its neighbours' estimates, the prior and the mask is requested here, before the patch weights:
	// the loads do not depend on each other, so they share one memory round trip with the patch texels (pm_fill_patch ends in a barrier, which
	// keeps the compiler from sinking them below the tests that decide whether the pixel is processed).  Neighbours outside the processable
	// area are redirected to the pixel itself and masked; a pixel that turns out not to be processed simply ignores what it fetched.
	size_t qis[4]; bool bok[4]; int qxs[4], qys[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int ox = (k == 0) ? sgn : (k == 2 ? -sgn : 0), oy = (k == 1) ? sgn : (k == 3 ? -sgn : 0);
		// bounds tests exactly as written: x > HW / y > HW / x < W-HW / y < H-HW
		bool ok;
		if (ox == -1) ok = x > PM_HW; else if (ox == 1) ok = x < w - PM_HW; else if (oy == -1) ok = y > PM_HW; else ok = y < h - PM_HW;
		bok[k] = ok; qxs[k] = x + ox; qys[k] = y + oy;
		qis[k] = ok ? (size_t)(y + oy) * w + (x + ox) : idx;
	}
	float nds[4] = {0.f, 0.f, 0.f, 0.f};
	float on0[SL], on1[SL], on2[SL];
	float oDepth = 0.f, oNx = 0.f, oNy = 0.f, oNz = 0.f, oConf = 2.f, prior = 0.f;
	unsigned char maskByte = 1;
#pragma unroll
	for (int q = 0; q < SL; ++q) { on0[q] = on1[q] = 0.f; on2[q] = 1.f; }
	if (active) {
		if (t.prior) prior = pm_glob(t.prior)[idx];
		if (t.mask != nullptr) maskByte = t.mask[idx];
#pragma unroll
		for (int k = 0; k < 4; ++k) nds[k] = gDepth[qis[k]];
#pragma unroll
		for (int q = 0; q < SL; ++q) {
			const int k = (G >= 4) ? (v & 3) : q * G + v;   // G >= 4: every quad of the group holds all four slots (lane v owns slot v % 4)
			const size_t qi = (k == 0) ? qis[0] : (k == 1) ? qis[1] : (k == 2) ? qis[2] : (k == 3) ? qis[3] : idx;
			on0[q] = gNormal[qi * 3]; on1[q] = gNormal[qi * 3 + 1]; on2[q] = gNormal[qi * 3 + 2];
		}
		oDepth = gDepth[idx]; oNx = gNormal[idx * 3]; oNy = gNormal[idx * 3 + 1]; oNz = gNormal[idx * 3 + 2]; oConf = gConf[idx];
	}
	float normSq0, sumW;
	pm_fill_patch<G, true>(t, active, x, y, v, s_w[g], normSq0, sumW);
	const bool masked = active && maskByte == 0;
	const bool valid = active && !masked && !(normSq0 < kp.thMagnitudeSq && !(prior > 0));
	// prior and its blend factor (DepthMap.cpp:558-559) go to the spare entry of the weight row: pm_score_view<.., PF = true> reads them there
	if (v == 0) s_w[g][PM_NT] = make_float2(prior, prior > 0 ? pm_expf(normSq0 * (-1.f / (1.f * 0.02f))) : 0.f);
	__syncthreads();
	PM_TICK(12);
	const double X0x = ((double)x - t.cx) / t.fx, X0y = ((double)y - t.cy) / t.fy;
	const float vx = (float)X0x, vy = (float)X0y, vz = 1.f;

	float depth = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, conf = 2.f;
	bool pok0 = false, pok1 = false; // propagation candidates (slots 0 and 1) exist; their estimates are re-read when used
	float qX0[SL], qX1[SL], qX2[SL], qn0[SL], qn1[SL], qn2[SL]; // my smoothness slot(s)
	unsigned closeMask = 0u;
#pragma unroll
	for (int q = 0; q < SL; ++q) { qX0[q] = qX1[q] = qX2[q] = 0.f; qn0[q] = qn1[q] = 0.f; qn2[q] = 1.f; }
	if (valid) {
		depth = oDepth; nx = oNx; ny = oNy; nz = oNz; conf = oConf;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool ok = bok[k] && nds[k] > 0;
			if (ok) closeMask |= 1u << k;
			if (k == 0 && ok) pok0 = true;
			if (k == 1 && ok) pok1 = true;
			if (ok && ((G >= 4) ? (v & 3) == k : (k % G) == v)) {
				const int q = (k / G < SL) ? k / G : 0;
				// TransformPointI2C(Point3(nx, ndepth)) in double then Cast<float>, Camera.h:338-344
				const double z = (double)nds[k];
				qX0[q] = (float)(((double)qxs[k] - t.cx) * z / t.fx);
				qX1[q] = (float)(((double)qys[k] - t.cy) * z / t.fy);
				qX2[q] = (float)z;
				qn0[q] = on0[q]; qn1[q] = on1[q]; qn2[q] = on2[q];
			}
		}
	}
	// ---- stage source tiles in LDS -------------------------------------------------------------------
	// Every hypothesis of this visit (propagated neighbours' planes, small perturbations of the current plane)
	// projects close to where the current plane does, so one window per (wave, view) around the current
	// footprints serves nearly all 100 x ~8 taps; the rest (random restarts, depth discontinuities) fall back
	// to global loads tap-row by tap-row.  Window origin = min over the wave's pixels of the footprint centre.
	const float* tileBase = nullptr;
	if (TC > 0) {
		const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
		for (int u = 0; u < VPL; ++u) {
			const int view = v + u * G;
			int cs = 0x7fffffff, ctt = 0x7fffffff;
			if (valid && view < t.nSrc) {
				float Hc[9];
				pm_homography(hotBase + view * NBD, t, X0x, X0y, depth, nx, ny, nz, Hc);
				const float fxp = (float)x, fyp = (float)y;
				const float c0 = Hc[0] * fxp + Hc[1] * fyp + Hc[2], c1 = Hc[3] * fxp + Hc[4] * fyp + Hc[5], c2 = Hc[6] * fxp + Hc[7] * fyp + Hc[8];
				const float cu = c0 / c2, cv = c1 / c2;
				if (cu > -1e6f && cu < 1e6f && cv > -1e6f && cv < 1e6f) { const int iu = (int)pm_floorf(cu), iv = (int)pm_floorf(cv); cs = iu + iv; ctt = iv; }
			}
#pragma unroll
			for (int m = G; m < 64; m <<= 1) { cs = min(cs, __shfl_xor(cs, m, 64)); ctt = min(ctt, __shfl_xor(ctt, m, 64)); }
			if (cs == 0x7fffffff) { cs = 0; ctt = 0; }
			if (lane < G) s_org[wave][view] = make_int2(cs - 8 - (PM_TR - 19) / 2, ctt - PM_HW - (PM_TCX - 9) / 2);
		}
		__syncthreads();
		PM_TICK(13);
		float* tw = s_tile[wave];
		const int nS = t.nSrc;
		constexpr int TCD = TC > 0 ? TC : 1;
		constexpr int NLD = (PM_TR * TCD + 63) / 64;
		constexpr int WB = NLD > 8 ? 2 : PM_WINBATCH;   // wide windows (many pixels per wave): fewer in flight, the registers are needed
		// WB windows are requested together: the loads of a window are one memory round trip, and the round trips of
		// the nSrc windows of a visit are a serial chain at the head of every wave's life
		for (int vb = 0; vb < nS; vb += WB) {
			float vals[WB][NLD];
#pragma unroll
			for (int b = 0; b < WB; ++b) {
				const int vv = vb + b;
				if (vv >= nS) break;
				const int2 org = s_org[wave][vv];
				const int fs0 = org.x, ft0 = org.y;
				const pm_gcf src = pm_glob(t.src[vv].imgS);
				const int sh = t.src[vv].h, sMax = t.src[vv].w + t.src[vv].h - 1;
#pragma unroll
				for (int k = 0; k < NLD; ++k) { // all loads of the batch first, then the LDS writes
					const int i = lane + 64 * k;
					const int r = i / TCD, c = i - r * TCD;
					const int ss = fs0 + r, tt = ft0 + c;
#ifdef PM_PROBE_NO_STAGING
					vals[b][k] = 0.25f;
#else
					vals[b][k] = (i < PM_TR * TC && ss >= 0 && ss < sMax && tt >= 0 && tt < sh) ? src[(size_t)ss * sh + tt] : 0.f;
#endif
				}
			}
#pragma unroll
			for (int b = 0; b < WB; ++b) {
				const int vv = vb + b;
				if (vv >= nS) break;
#ifndef PM_PROBE_NO_STAGING
#pragma unroll
				for (int k = 0; k < NLD; ++k) { const int i = lane + 64 * k; if (i < PM_TR * TC) tw[vv * TSTRIDE + i] = vals[b][k]; }
#endif
			}
		}
		tileBase = tw;
		__syncthreads();
	}
	// state machine: every outer trip scores at most one hypothesis per pixel, so the lanes of a wave
	// stay converged on the expensive part whatever branch each pixel is in.
	enum { ST_PROP0 = 0, ST_PROP1 = 1, ST_DECIDE = 2, ST_RAND = 3, ST_REFINE = 4, ST_DONE = 5 };
	int st = valid ? ST_PROP0 : ST_DONE;
	unsigned it = 0, idxScale = 0;
	float scaleRange = 1.f, depthRange = 0.f, p0 = 0.f, p1 = 0.f;
	bool smooth = true, changed = false;
	const uint32_t k1 = t.k1base + pass;
	PmPhilox4 rNext; unsigned rNextIt = 0xffffffffu;   // draw of refinement iteration rNextIt, computed one evaluation ahead (counter-based: same bits)
#pragma unroll
	for (int i = 0; i < 4; ++i) rNext.v[i] = 0u;
	PM_TICK(0); PM_COUNT(9, 1);
	for (;;) {
		bool need = false;
		float hd = 0.f, hnx = 0.f, hny = 0.f, hnz = 1.f, hp0 = 0.f, hp1 = 0.f;
		int hst = ST_DONE;
		while (!need && st != ST_DONE) {
			if (st <= ST_PROP1) {
				const bool vert = (st == ST_PROP1); ++st; // slot 0: same row, slot 1: same column
				const bool pok = vert ? pok1 : pok0;
				// the neighbour (already updated in this sweep, one diagonal earlier) is not touched again before we are done
				const size_t qi = vert ? (size_t)(y + sgn) * w + x : (size_t)y * w + (x + sgn);
				const size_t qj = pok ? qi : idx;
				const float pconf = gConf[qj], cnx = gNormal[qj * 3], cny = gNormal[qj * 3 + 1], cnz = gNormal[qj * 3 + 2], cd = gDepth[qj]; // one batch
				// (assigned whether or not the candidate is taken -- every state that sets `need` sets all four again -- so that the five loads stay
				// one batch: the compiler otherwise sinks four of them below the test of the fifth, two dependent round trips per propagation state)
				hd = cd; hnx = cnx; hny = cny; hnz = cnz;
				if (pok && pconf < kp.thKeep) {
					// InterpolatePixel, DepthMap.cpp:915-959
					float depthNew = cd; bool zero;
					if (vert) { // same column
						const float nx1 = (float)(((double)y - t.cy) / t.fy);
						const float denom = cnz + nx1 * cny;
						zero = pm_fabsf(denom) < 0.0001f;
						const float x1 = (float)(((double)(y + sgn) - t.cy) / t.fy);
						const float nom = cd * (cnz + x1 * cny);
						if (!zero) depthNew = nom / denom;
					} else {
						const float nx1 = (float)(((double)x - t.cx) / t.fx);
						const float denom = cnz + nx1 * cnx;
						zero = pm_fabsf(denom) < 0.0001f;
						const float x1 = (float)(((double)(x + sgn) - t.cx) / t.fx);
						const float nom = cd * (cnz + x1 * cnx);
						if (!zero) depthNew = nom / denom;
					}
					hd = (!zero && pm_in_range(depthNew, t.dMin, t.dMax)) ? depthNew : cd;
					hnx = cnx; hny = cny; hnz = cnz;
					pm_correct_normal(vx, vy, vz, hnx, hny, hnz);
					need = true; hst = ST_PROP0;
				}
			} else if (st == ST_DECIDE) {
				// RefineIters:, DepthMap.cpp:802-827
				if (conf <= kp.thConfSmall) idxScale = 2;
				else if (conf <= kp.thConfBig) idxScale = 1;
				else if (conf >= kp.thConfRand) { smooth = false; st = ST_RAND; it = 0; continue; }
				scaleRange = pm_pow2neg(idxScale);
				depthRange = depth * kp.depthRatio;
				p0 = pm_atan2f(ny, nx); p1 = pm_acosf(pm_clampf(nz, -1.f, 1.f)); // Normal2Dir
				st = ST_REFINE; it = 0;
			} else if (st == ST_RAND) {
				if (it >= kp.nRandomIters) { st = ST_DONE; break; }
				const PmPhilox4 r = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(PM_STREAM_RAND * 256) + it, 0u, t.k0, k1);
				++it;
				const float rr = t.dMinSqr + (t.dMaxSqr - t.dMinSqr) * pm_u32_to_unit(r.v[0]);
				hd = rr * rr;
				pm_random_normal(pm_u32_to_unit(r.v[1]), pm_u32_to_unit(r.v[2]), vx, vy, vz, hnx, hny, hnz);
				need = true; hst = ST_RAND;
			} else { // ST_REFINE, DepthMap.cpp:832-852
				if (it >= kp.nRandomIters) { st = ST_DONE; break; }
				PmPhilox4 r = rNext;
				if (!PM_ILP_PHILOX || rNextIt != it) r = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(PM_STREAM_REFINE * 256) + it, 0u, t.k0, k1);
				++it;
				const float ndepth = depth + (depthRange * scaleRange) * (2.f * pm_u32_to_unit(r.v[0]) - 1.f);
				if (!pm_in_range(ndepth, t.dMin, t.dMax)) continue;
				hp0 = p0 + (kp.angle1Range * scaleRange) * (2.f * pm_u32_to_unit(r.v[1]) - 1.f);
				hp1 = p1 + (kp.angle2Range * scaleRange) * (2.f * pm_u32_to_unit(r.v[2]) - 1.f);
				pm_dir2normal(hp0, hp1, hnx, hny, hnz);
				if (hnx * vx + hny * vy + hnz * vz >= 0) continue;
				hd = ndepth;
				need = true; hst = ST_REFINE;
			}
		}
		if (!__any(need)) break;
		PM_TICK(1); PM_COUNT(6 + 5, 0); PM_COUNT(8, __popcll(__ballot(need)));
		// smoothness factors of the hypothesis plane w.r.t. the close neighbours, DepthMap.cpp:524-533
		// smoothness factors of the hypothesis plane w.r.t. the close neighbours, DepthMap.cpp:524-533, one neighbour per lane.  (With the PM_ILP_*
		// switches the homography of the lane's view and the next refinement draw join this block branch-free, so that the scheduler could interleave
		// the three dependent chains; measured slower, see the switches.)
		float sf[4] = {1.f, 1.f, 1.f, 1.f};
		float Hpre[9];
		{
			const bool useS = need && smooth;
			const float planeD = -hd * (hnx * vx + hny * vy + hnz * vz); // InitPlane, DepthMap.cpp:963-971
			float myF[SL];
#pragma unroll
			for (int q = 0; q < SL; ++q) {
				const int k = (G >= 4) ? (v & 3) : q * G + v;
#ifdef PM_PROBE_NO_SMOOTH
				const bool on = false;
#else
				const bool on = !(PM_SMOOTH_IN_ROW0 && VPL == 1 && G >= 4 && TC > 0) && useS && k < 4 && ((closeMask >> k) & 1u);
#endif
				myF[q] = 1.f;
				if (PM_ILP_SMOOTH || on) {
					const float dist = (hnx * qX0[q] + (hny * qX1[q] + hnz * qX2[q])) + planeD; // Planef::Distance, Eigen 3-dot order
					const float r = dist / hd;
					const float factorDepth = pm_expf((r * r) * kp.smoothSigmaDepth);
					const float ca = pm_clampf((hnx * qn0[q] + hny * qn1[q] + hnz * qn2[q]) /
						pm_sqrtf((hnx * hnx + hny * hny + hnz * hnz) * (qn0[q] * qn0[q] + qn1[q] * qn1[q] + qn2[q] * qn2[q])), -1.f, 1.f);
					const float ac = pm_acosf(ca);
					const float factorNormal = pm_expf((ac * ac) * kp.smoothSigmaNormal);
					const float f = (1.f - kp.smoothBonusDepth * factorDepth) * (1.f - kp.smoothBonusNormal * factorNormal);
					myF[q] = on ? f : 1.f;
				}
			}
			if (VPL == 1 && PM_ILP_HOMOGRAPHY) pm_homography(hotBase + v * NBD, t, X0x, X0y, hd, hnx, hny, hnz, Hpre);
			if (PM_ILP_PHILOX) { rNext = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(PM_STREAM_REFINE * 256) + it, 0u, t.k0, k1); rNextIt = it; }
			if (G >= 4) { sf[0] = pm_quad_bcast<0>(myF[0]); sf[1] = pm_quad_bcast<1>(myF[0]); sf[2] = pm_quad_bcast<2>(myF[0]); sf[3] = pm_quad_bcast<3>(myF[0]); }
			else {
#pragma unroll
				for (int k = 0; k < 4; ++k)
					sf[k] = __shfl(myF[k / G < SL ? k / G : 0], (k % G), G);
			}
		}
		PM_TICK(2);
		float sc = PM_INF, sc2 = PM_INF;   // the lane's two smallest view scores
		if (PM_SMOOTH_IN_ROW0 && VPL == 1 && G >= 4 && TC > 0) {
			// every lane of a pixel with a hypothesis enters: the smoothness chain is evaluated next to the first tap row (see the switch)
			if (need) {
				const bool viewOk = v < t.nSrc;
				const int slot = v & 3;
				PMSmoothIn sin;
				sin.qX0 = qX0[0]; sin.qX1 = qX1[0]; sin.qX2 = qX2[0]; sin.qn0 = qn0[0]; sin.qn1 = qn1[0]; sin.qn2 = qn2[0];
				sin.vx = vx; sin.vy = vy; sin.vz = vz;
				sin.on = smooth && ((closeMask >> slot) & 1u);
				const int2 org = s_org[threadIdx.x >> 6][v];
				const float s1 = pm_score_view<GEO, true, TC, true, true>(t.src[v], t, kp, x, y, X0x, X0y, normSq0, sumW, s_w[g], hd, hnx, hny, hnz, 1.f, 1.f, 1.f, 1.f, 0.f,
					tileBase + v * TSTRIDE, org.x, org.y, hotBase + v * NBD, hotBase + v * NBD + PM_SRC_HOT, nullptr, &sin, viewOk PM_PROF_PASS);
				if (viewOk) sc = s1;
			}
		} else {
#pragma unroll 1
		for (int u = 0; u < VPL; ++u) {
			const int view = v + u * G;
			if (need && view < t.nSrc) {
				int2 org = make_int2(0, 0);
				if (TC > 0) org = s_org[threadIdx.x >> 6][view];
				const float s1 = pm_score_view<GEO, true, TC, true>(t.src[view], t, kp, x, y, X0x, X0y, normSq0, sumW, s_w[g], hd, hnx, hny, hnz, sf[0], sf[1], sf[2], sf[3], 0.f,
					tileBase + view * TSTRIDE, org.x, org.y, hotBase + view * NBD, hotBase + view * NBD + PM_SRC_HOT, (VPL == 1 && PM_ILP_HOMOGRAPHY) ? Hpre : nullptr PM_PROF_PASS);
				if (s1 < sc) { sc2 = sc; sc = s1; } else if (s1 < sc2) sc2 = s1;
			}
		}
		}
		const float nconf = pm_aggregate<G>(sc, t.nSrc, kp.thRobust, sc2);
		if (need && conf > nconf) {
			conf = nconf; depth = hd; nx = hnx; ny = hny; nz = hnz; changed = true;
			if (hst == ST_RAND) { if (conf < kp.thConfRand) st = ST_DECIDE; }
			else if (hst == ST_REFINE) { p0 = hp0; p1 = hp1; scaleRange = pm_pow2neg(++idxScale); }
		}
		PM_TICK(6);
	}
	PM_PROF_FLUSH();
	if (changed && v == 0) { gDepth[idx] = depth; gNormal[idx * 3] = nx; gNormal[idx * 3 + 1] = ny; gNormal[idx * 3 + 2] = nz; gConf[idx] = conf; }
}

// -------------------------------------------------------------------------------------------
// ProcessPixel in "latency mode": ONE WAVE PER PIXEL.  pm_sweep_kernel gives a wave 64 / G pixels and walks each pixel's hypotheses one after the
// other (<= 2 propagation candidates, then <= nRandomIters refinements or random restarts); with one depth map (BASELINE config 2) or a handful of
// them a diagonal launch cannot fill 1 024 SIMDs and its duration is one wave's serial chain of ~8 evaluations.  Here the eight 8-lane groups of the
// wave belong to the same pixel and each scores a DIFFERENT hypothesis in the same round (lane = candidate * 8 + source view):
//   round 1: group 0 / 1 = the two propagation candidates; groups 2..7 = the first six hypotheses of the stage that follows if both are rejected
//            (refinements of the current plane, or random restarts when conf >= thConfRand);
//   later rounds: the next <= 8 refinement / restart iterations from the then-current state.
// After a round every lane knows all eight scores and replays the reference's sequential accept rule (DepthMap.cpp:772-852) over them in order;
// candidates that were computed from a state an earlier accept has changed are discarded and recomputed in the next round.  Hypotheses, draws
// (counter-based: iteration index, not call order), scores and the order of the comparisons are those of the sequential code, so the result is
// the same bits; only evaluations whose outcome the reference would never look at are extra work.  Expected rounds = 1 + number of accepts.
// nSrc <= 8 (one source view per lane of a group).
template <bool GEO>
__global__ __launch_bounds__(64, PM_WIDE_MINWAVES) void pm_sweep_wide_kernel(const PMTask* __restrict__ tasks, PMKParams kp, int dir, int d, int xlo, int count, uint32_t pass) {
	constexpr int G = 8;
	constexpr int TC = PM_WIDE_TILES ? 1 + PM_TCX : 0;   // 0: no LDS windows, the tap rows read the quad image (as pm_sweep2_kernel)
	constexpr int TSTRIDE = PM_TR * (TC > 0 ? TC : 1) + PM_TILE_PAD;
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	PM_PROF_DECL;
	__shared__ float2 s_w[PM_NT + 1];
	__shared__ float s_tile[G * TSTRIDE];
	__shared__ int2 s_org[G];
	__shared__ double s_src[G * NBD];
	const PMTask& t = tasks[blockIdx.y];
	const int lane = threadIdx.x, c = lane >> 3, v = lane & 7;
	for (int i = lane; i < G * NBD; i += 64) s_src[i] = ((const double*)&t.src[i / NBD])[i % NBD];
	const double* hot = s_src + v * NBD;
	const int w = t.w, h = t.h;
	const int x = xlo + (int)blockIdx.x, y = d - x;            // grid.x == count: every wave has a pixel
	const size_t idx = (size_t)y * w + x;
	const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
	const int sgn = dir == 0 ? -1 : 1;
	// neighbour slots as in pm_sweep_kernel: slot0 (x+sgn,y), slot1 (x,y+sgn) are the propagation sources, slot2 (x-sgn,y), slot3 (x,y-sgn)
	size_t qis[4]; bool bok[4]; int qxs[4], qys[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int ox = (k == 0) ? sgn : (k == 2 ? -sgn : 0), oy = (k == 1) ? sgn : (k == 3 ? -sgn : 0);
		bool ok;
		if (ox == -1) ok = x > PM_HW; else if (ox == 1) ok = x < w - PM_HW; else if (oy == -1) ok = y > PM_HW; else ok = y < h - PM_HW;
		bok[k] = ok; qxs[k] = x + ox; qys[k] = y + oy;
		qis[k] = ok ? (size_t)(y + oy) * w + (x + ox) : idx;
	}
	float nds[4], prior = 0.f;
	unsigned char maskByte = 1;
	if (t.prior) prior = pm_glob(t.prior)[idx];
	if (t.mask != nullptr) maskByte = t.mask[idx];
#pragma unroll
	for (int k = 0; k < 4; ++k) nds[k] = gDepth[qis[k]];
	const int slot = v & 3;                                        // my smoothness slot (both quads of a group hold all four)
	const size_t qv = (slot == 0) ? qis[0] : (slot == 1) ? qis[1] : (slot == 2) ? qis[2] : qis[3];
	const float on0 = gNormal[qv * 3], on1 = gNormal[qv * 3 + 1], on2 = gNormal[qv * 3 + 2];
	const float oDepth = gDepth[idx], oNx = gNormal[idx * 3], oNy = gNormal[idx * 3 + 1], oNz = gNormal[idx * 3 + 2], oConf = gConf[idx];
	// the two propagation sources' estimates (they were updated one diagonal earlier and are not touched again before this launch ends)
	float pcf[2], pcd[2], pcn[2][3];
#pragma unroll
	for (int k = 0; k < 2; ++k) { const size_t q = qis[k]; pcf[k] = gConf[q]; pcd[k] = gDepth[q]; pcn[k][0] = gNormal[q * 3]; pcn[k][1] = gNormal[q * 3 + 1]; pcn[k][2] = gNormal[q * 3 + 2]; }
	float normSq0, sumW;
	pm_fill_patch<64, true>(t, true, x, y, lane, s_w, normSq0, sumW);
	const bool masked = maskByte == 0;
	const bool valid = !masked && !(normSq0 < kp.thMagnitudeSq && !(prior > 0));
	if (lane == 0) s_w[PM_NT] = make_float2(prior, prior > 0 ? pm_expf(normSq0 * (-1.f / (1.f * 0.02f))) : 0.f);
	__syncthreads();
	if (!valid) return;                                           // wave-uniform
	const double X0x = ((double)x - t.cx) / t.fx, X0y = ((double)y - t.cy) / t.fy;
	const float vx = (float)X0x, vy = (float)X0y, vz = 1.f;
	float depth = oDepth, nx = oNx, ny = oNy, nz = oNz, conf = oConf;
	bool pok[2] = {false, false};
	unsigned closeMask = 0u;
	float qX0 = 0.f, qX1 = 0.f, qX2 = 0.f, qn0 = 0.f, qn1 = 0.f, qn2 = 1.f;   // my smoothness slot (slot v, v < 4)
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const bool ok = bok[k] && nds[k] > 0;
		if (ok) closeMask |= 1u << k;
		if (k < 2) pok[k] = ok;
		if (ok && k == slot) {
			const double z = (double)nds[k];
			qX0 = (float)(((double)qxs[k] - t.cx) * z / t.fx);
			qX1 = (float)(((double)qys[k] - t.cy) * z / t.fy);
			qX2 = (float)z;
			qn0 = on0; qn1 = on1; qn2 = on2;
		}
	}
	// ---- windows: one per source view around the footprint of the current plane ----------------------------------------------------------------
	if (TC > 0) {
		int cs = 0, ctt = 0;
		if (v < t.nSrc) {
			float Hc[9];
			pm_homography(hot, t, X0x, X0y, depth, nx, ny, nz, Hc);
			const float fxp = (float)x, fyp = (float)y;
			const float c0 = Hc[0] * fxp + Hc[1] * fyp + Hc[2], c1 = Hc[3] * fxp + Hc[4] * fyp + Hc[5], c2 = Hc[6] * fxp + Hc[7] * fyp + Hc[8];
			const float cu = c0 / c2, cv = c1 / c2;
			if (cu > -1e6f && cu < 1e6f && cv > -1e6f && cv < 1e6f) { const int iu = (int)pm_floorf(cu), iv = (int)pm_floorf(cv); cs = iu + iv; ctt = iv; }
		}
		if (c == 0) s_org[v] = make_int2(cs - 8 - (PM_TR - 19) / 2, ctt - PM_HW - (PM_TCX - 9) / 2);
		__syncthreads();
		constexpr int NE = PM_TR * (TC > 0 ? TC : 1), NLD = (G * NE + 63) / 64;
		float vals[NLD];
#pragma unroll
		for (int k = 0; k < NLD; ++k) {
			const int e = lane + 64 * k;
			const int vv = e / NE, i = e - vv * NE;
			const int r = i / TC, cc = i - r * TC;
			float val = 0.f;
			if (vv < t.nSrc) {
				const int2 org = s_org[vv];
				const int sh = t.src[vv].h, sMax = t.src[vv].w + t.src[vv].h - 1;
				const int ss = org.x + r, tt = org.y + cc;
				if (ss >= 0 && ss < sMax && tt >= 0 && tt < sh) val = pm_glob(t.src[vv].imgS)[(size_t)ss * sh + tt];
			}
			vals[k] = val;
		}
#pragma unroll
		for (int k = 0; k < NLD; ++k) { const int e = lane + 64 * k; if (e < G * NE) { const int vv = e / NE; s_tile[vv * TSTRIDE + (e - vv * NE)] = vals[k]; } }
		__syncthreads();
	}
	const int2 myOrg = s_org[v];
	const float* tile = s_tile + v * TSTRIDE;
	const uint32_t k1 = t.k1base + pass;
	// ---- rounds -----------------------------------------------------------------------------------------------------------------------------------
	enum { W_PROPS = 0, W_REFINE = 1, W_RAND = 2, W_DONE = 3 };
	int stage = W_PROPS;
	unsigned it0 = 0, idxScale = 0;
	float scaleRange = 1.f, depthRange = 0.f, p0 = 0.f, p1 = 0.f;
	bool smooth = true, changed = false;
	PM_TICK(0); PM_COUNT(9, 1);
	while (stage != W_DONE) {
		// what the stage after the propagation candidates would be if they change nothing: RefineIters: (DepthMap.cpp:802-827) on the current state
		int specStage = W_REFINE; unsigned specIdx = idxScale; bool specSmooth = smooth;
		float specScale = scaleRange, specRange = depthRange, specP0 = p0, specP1 = p1;
		if (stage == W_PROPS) {
			if (conf <= kp.thConfSmall) specIdx = 2;
			else if (conf <= kp.thConfBig) specIdx = 1;
			else if (conf >= kp.thConfRand) { specSmooth = false; specStage = W_RAND; }
			if (specStage == W_REFINE) {
				specScale = pm_pow2neg(specIdx);
				specRange = depth * kp.depthRatio;
				specP0 = pm_atan2f(ny, nx); specP1 = pm_acosf(pm_clampf(nz, -1.f, 1.f)); // Normal2Dir
			}
		}
		// ---- my group's hypothesis ----
		bool need = false, useSmooth = smooth;
		float hd = 0.f, hnx = 0.f, hny = 0.f, hnz = 1.f, hp0 = 0.f, hp1 = 0.f;
		const int first = (stage == W_PROPS) ? 2 : 0;          // first group that holds an iteration candidate
		const int kind = (stage == W_PROPS) ? specStage : stage;
		if (stage == W_PROPS && c < 2) {
			const bool vert = (c == 1);
			const float cd = vert ? pcd[1] : pcd[0], cnx = vert ? pcn[1][0] : pcn[0][0], cny = vert ? pcn[1][1] : pcn[0][1], cnz = vert ? pcn[1][2] : pcn[0][2];
			const bool take = (vert ? pok[1] : pok[0]) && (vert ? pcf[1] : pcf[0]) < kp.thKeep;
			// InterpolatePixel, DepthMap.cpp:915-959
			float depthNew = cd; bool zero;
			if (vert) {
				const float nx1 = (float)(((double)y - t.cy) / t.fy);
				const float denom = cnz + nx1 * cny;
				zero = pm_fabsf(denom) < 0.0001f;
				const float x1 = (float)(((double)(y + sgn) - t.cy) / t.fy);
				const float nom = cd * (cnz + x1 * cny);
				if (!zero) depthNew = nom / denom;
			} else {
				const float nx1 = (float)(((double)x - t.cx) / t.fx);
				const float denom = cnz + nx1 * cnx;
				zero = pm_fabsf(denom) < 0.0001f;
				const float x1 = (float)(((double)(x + sgn) - t.cx) / t.fx);
				const float nom = cd * (cnz + x1 * cnx);
				if (!zero) depthNew = nom / denom;
			}
			hd = (!zero && pm_in_range(depthNew, t.dMin, t.dMax)) ? depthNew : cd;
			hnx = cnx; hny = cny; hnz = cnz;
			pm_correct_normal(vx, vy, vz, hnx, hny, hnz);
			need = take;
		} else {
			const unsigned itc = it0 + (unsigned)(c - first);
			if (c >= first && itc < kp.nRandomIters) {
				if (kind == W_RAND) {
					const PmPhilox4 r = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(PM_STREAM_RAND * 256) + itc, 0u, t.k0, k1);
					const float rr = t.dMinSqr + (t.dMaxSqr - t.dMinSqr) * pm_u32_to_unit(r.v[0]);
					hd = rr * rr;
					pm_random_normal(pm_u32_to_unit(r.v[1]), pm_u32_to_unit(r.v[2]), vx, vy, vz, hnx, hny, hnz);
					need = true; useSmooth = false;
				} else {
					const float sR = (stage == W_PROPS) ? specScale : scaleRange, dR = (stage == W_PROPS) ? specRange : depthRange;
					const float b0 = (stage == W_PROPS) ? specP0 : p0, b1 = (stage == W_PROPS) ? specP1 : p1;
					const PmPhilox4 r = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(PM_STREAM_REFINE * 256) + itc, 0u, t.k0, k1);
					const float ndepth = depth + (dR * sR) * (2.f * pm_u32_to_unit(r.v[0]) - 1.f);
					hp0 = b0 + (kp.angle1Range * sR) * (2.f * pm_u32_to_unit(r.v[1]) - 1.f);
					hp1 = b1 + (kp.angle2Range * sR) * (2.f * pm_u32_to_unit(r.v[2]) - 1.f);
					pm_dir2normal(hp0, hp1, hnx, hny, hnz);
					hd = ndepth;
					need = pm_in_range(ndepth, t.dMin, t.dMax) && !(hnx * vx + hny * vy + hnz * vz >= 0);
					useSmooth = (stage == W_PROPS) ? specSmooth : smooth;
				}
			}
		}
		PM_TICK(1); PM_COUNT(8, __popcll(__ballot(need)));
		// ---- smoothness factors of my group's plane (slot v for v < 4), DepthMap.cpp:524-533 ----
		float sf[4];
		{
			float myF = 1.f;
			if (need && useSmooth && ((closeMask >> slot) & 1u)) {
				const float planeD = -hd * (hnx * vx + hny * vy + hnz * vz);
				const float dist = (hnx * qX0 + (hny * qX1 + hnz * qX2)) + planeD;
				const float r = dist / hd;
				const float factorDepth = pm_expf((r * r) * kp.smoothSigmaDepth);
				const float ca = pm_clampf((hnx * qn0 + hny * qn1 + hnz * qn2) / pm_sqrtf((hnx * hnx + hny * hny + hnz * hnz) * (qn0 * qn0 + qn1 * qn1 + qn2 * qn2)), -1.f, 1.f);
				const float ac = pm_acosf(ca);
				const float factorNormal = pm_expf((ac * ac) * kp.smoothSigmaNormal);
				myF = (1.f - kp.smoothBonusDepth * factorDepth) * (1.f - kp.smoothBonusNormal * factorNormal);
			}
			sf[0] = pm_quad_bcast<0>(myF); sf[1] = pm_quad_bcast<1>(myF); sf[2] = pm_quad_bcast<2>(myF); sf[3] = pm_quad_bcast<3>(myF);
		}
		PM_TICK(2);
		float sc = PM_INF;
		if (need && v < t.nSrc)
			sc = pm_score_view<GEO, true, TC, true>(t.src[v], t, kp, x, y, X0x, X0y, normSq0, sumW, s_w, hd, hnx, hny, hnz, sf[0], sf[1], sf[2], sf[3], 0.f,
				tile, myOrg.x, myOrg.y, hot, hot + PM_SRC_HOT, nullptr PM_PROF_PASS);
		const float nconf = pm_aggregate<G>(sc, t.nSrc, kp.thRobust);
		// ---- every lane replays the sequential accept rule over the eight groups' results, in the reference's order ----
		bool restart = false;                                     // the state changed in a way that invalidates the remaining candidates
		if (stage == W_PROPS) {
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const int src = k * G;
				if (__shfl(need ? 1 : 0, src, 64) && conf > __shfl(nconf, src, 64)) {
					conf = __shfl(nconf, src, 64); depth = __shfl(hd, src, 64); nx = __shfl(hnx, src, 64); ny = __shfl(hny, src, 64); nz = __shfl(hnz, src, 64);
					changed = true; restart = true;
				}
			}
			// RefineIters: on the state the propagation left
			if (conf <= kp.thConfSmall) idxScale = 2;
			else if (conf <= kp.thConfBig) idxScale = 1;
			else if (conf >= kp.thConfRand) { smooth = false; stage = W_RAND; }
			if (stage != W_RAND) {
				scaleRange = pm_pow2neg(idxScale);
				depthRange = depth * kp.depthRatio;
				p0 = pm_atan2f(ny, nx); p1 = pm_acosf(pm_clampf(nz, -1.f, 1.f));
				stage = W_REFINE;
			}
			it0 = 0;
		}
		if (!restart) {
			// groups first .. 7 hold iterations it0, it0 + 1, ... of `stage` computed from exactly the present state
			for (int j = first; j < 8; ++j) {
				const unsigned itj = it0 + (unsigned)(j - first);
				if (itj >= kp.nRandomIters) break;
				const int src = j * G;
				const bool nd = __shfl(need ? 1 : 0, src, 64) != 0;
				const float ncj = __shfl(nconf, src, 64);
				if (nd && conf > ncj) {
					conf = ncj; depth = __shfl(hd, src, 64); nx = __shfl(hnx, src, 64); ny = __shfl(hny, src, 64); nz = __shfl(hnz, src, 64);
					changed = true;
					if (stage == W_REFINE) {
						p0 = __shfl(hp0, src, 64); p1 = __shfl(hp1, src, 64); scaleRange = pm_pow2neg(++idxScale);
						it0 = itj + 1; restart = true; break;     // the later refinements perturb the new plane: next round
					} else if (conf < kp.thConfRand) {
						// goto RefineIters (DepthMap.cpp:790-793): the remaining restarts are dropped
						if (conf <= kp.thConfSmall) idxScale = 2;
						else if (conf <= kp.thConfBig) idxScale = 1;
						scaleRange = pm_pow2neg(idxScale);
						depthRange = depth * kp.depthRatio;
						p0 = pm_atan2f(ny, nx); p1 = pm_acosf(pm_clampf(nz, -1.f, 1.f));
						stage = W_REFINE; it0 = 0; restart = true; break;
					}
				}
			}
			if (!restart) it0 += (unsigned)(8 - first);
		}
		if (it0 >= kp.nRandomIters) stage = W_DONE;                // DepthMap.cpp:833 / :781: the iteration budget of the stage is used up
		PM_TICK(6);
	}
	PM_PROF_FLUSH();
	if (changed && lane == 0) { gDepth[idx] = depth; gNormal[idx * 3] = nx; gNormal[idx * 3 + 1] = ny; gNormal[idx * 3 + 2] = nz; gConf[idx] = conf; }
}

// EndDepthMapTmp, SceneDensify.cpp:528-576
__global__ void pm_finalize_kernel(const PMTask* __restrict__ tasks, float thKeep) {
	const PMTask& t = tasks[blockIdx.y];
	const size_t n = (size_t)t.w * t.h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float depth = t.depth[i], conf = t.conf[i];
		if (depth <= 0 || conf >= thKeep) { t.conf[i] = 0.f; t.depth[i] = 0.f; t.normal[i * 3] = 0.f; t.normal[i * 3 + 1] = 0.f; t.normal[i * 3 + 2] = 0.f; }
		else t.conf[i] = conf >= 1.f ? 0.f : 1.f - conf;
	}
}

// ---- resampling (cv::resize restated; see oracle header for the conventions) ---------------
// INTER_AREA, integer factor f (ScaleDepthData, SceneDensify.cpp:586): dst[img][y][x], dw x dh = cvRound(sw/f) x cvRound(sh/f).
// Blocks cut by the right/bottom border average the available pixels; a cut bottom row takes that path for every pixel
// (OpenCV ResizeAreaFast_Invoker) -- see the oracle's resizeArea.
__global__ void pm_area_kernel(const float* __restrict__ src, float* __restrict__ dst, int sw, int sh, int dw, int dh, int f, int nImg) {
	const size_t n = (size_t)dw * dh * nImg;
	const float scale = 1.f / (float)(f * f);
	const int fullCols = sw / f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % dw), y = (int)((i / dw) % dh); const size_t im = i / ((size_t)dw * dh);
		const float* s = src + im * (size_t)sw * sh;
		const int sx0 = x * f, sy0 = y * f;
		float o;
		if (sy0 >= sh || sx0 >= sw) o = 0.f;
		else if (sy0 + f <= sh && x < fullCols) {
			if (f == 2) {
				const float* p = s + (size_t)sy0 * sw + sx0;
				o = ((p[0] + p[1]) + (p[sw] + p[sw + 1])) * 0.25f;
			} else {
				float sum = 0.f;
				for (int j = 0; j < f; ++j) for (int k = 0; k < f; ++k) sum += s[(size_t)(sy0 + j) * sw + (sx0 + k)];
				o = sum * scale;
			}
		} else {
			float sum = 0.f; int count = 0;
			for (int j = 0; j < f && sy0 + j < sh; ++j) for (int k = 0; k < f && sx0 + k < sw; ++k) { sum += s[(size_t)(sy0 + j) * sw + (sx0 + k)]; ++count; }
			o = sum / (float)count;
		}
		dst[i] = o;
	}
}
// anti-diagonal-major copy of nImg row-major images: dst[img][(u+v)*h + v] = src[img][v*w + u]
__global__ void pm_skew_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, int nImg) {
	const size_t n = (size_t)w * h * nImg, sp = (size_t)(w + h - 1) * h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int u = (int)(i % w), v = (int)((i / w) % h); const size_t im = i / ((size_t)w * h);
		dst[im * sp + (size_t)(u + v) * h + v] = src[i];
	}
}
// anti-diagonal-major quad image of nImg row-major images: dst[img][(u+v)*h + v] = {I(u,v), I(u+1,v), I(u,v+1), I(u+1,v+1)}, neighbours clamped at the
// border (a bilinear sample reads entries with u <= w-2, v <= h-2 only: TImage::sample after isInsideWithBorder<1>)
__global__ void pm_quad_kernel(const float* __restrict__ src, float4* __restrict__ dst, int w, int h, int nImg) {
	const size_t n = (size_t)w * h * nImg, sp = (size_t)(w + h - 1) * h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int u = (int)(i % w), v = (int)((i / w) % h); const size_t im = i / ((size_t)w * h);
		const float* s = src + im * (size_t)w * h;
		const int u1 = min(u + 1, w - 1), v1 = min(v + 1, h - 1);
		dst[im * sp + (size_t)(u + v) * h + v] = make_float4(s[(size_t)v * w + u], s[(size_t)v * w + u1], s[(size_t)v1 * w + u], s[(size_t)v1 * w + u1]);
	}
}
__device__ __forceinline__ void pm_linear_coef(int d, int dn, int sn, int& s, float& a) {
	const double scale = (double)sn / (double)dn;
	float f = (float)(((double)d + 0.5) * scale - 0.5);
	int si = (int)pm_floorf(f);
	f -= (float)si;
	if (si < 0) { f = 0.f; si = 0; }
	if (si >= sn - 1) { f = 0.f; si = sn - 1; }
	s = si; a = f;
}
// level hand-off (SceneDensify.cpp:660-664): depth INTER_LINEAR, normal INTER_NEAREST, prior = copy of depth
struct PMUpTask { const float* sdepth; const float* snormal; float* ddepth; float* dnormal; float* dprior; };
// nearestDepth != 0: the reference resizes the depth map with INTER_NEAREST too when ignore masks are enabled (SceneDensify.cpp:661)
__global__ void pm_upsample_kernel(const PMUpTask* __restrict__ ups, int sw, int sh, int dw, int dh, int nearestDepth) {
	const PMUpTask u = ups[blockIdx.y];
	const size_t n = (size_t)dw * dh;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % dw), y = (int)(i / dw);
		int x0, y0; float a1, b1;
		pm_linear_coef(x, dw, sw, x0, a1); pm_linear_coef(y, dh, sh, y0, b1);
		const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
		const float a0 = 1.f - a1, b0 = 1.f - b1;
		const float t0 = u.sdepth[(size_t)y0 * sw + x0] * a0 + u.sdepth[(size_t)y0 * sw + x1] * a1;
		const float t1 = u.sdepth[(size_t)y1 * sw + x0] * a0 + u.sdepth[(size_t)y1 * sw + x1] * a1;
		float dv = t0 * b0 + t1 * b1;
		const int nx = min((int)floor((double)x * ((double)sw / (double)dw)), sw - 1);
		const int ny = min((int)floor((double)y * ((double)sh / (double)dh)), sh - 1);
		const size_t si = (size_t)ny * sw + nx;
		if (nearestDepth) dv = u.sdepth[si];
		u.ddepth[i] = dv; u.dprior[i] = dv;
		u.dnormal[i * 3] = u.snormal[si * 3]; u.dnormal[i * 3 + 1] = u.snormal[si * 3 + 1]; u.dnormal[i * 3 + 2] = u.snormal[si * 3 + 2];
	}
}
// INTER_NEAREST down-sampling of the initial estimate to the coarsest level (ScaleDepthData, SceneDensify.cpp:596-599):
// fx = 1/f given, so the source index is min(dst*f, size-1)
__global__ void pm_nearest_down_kernel(const PMUpTask* __restrict__ ups, int sw, int sh, int dw, int dh, int f) {
	const PMUpTask u = ups[blockIdx.y];
	const size_t n = (size_t)dw * dh;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % dw), y = (int)(i / dw);
		const int nx = min(x * f, sw - 1), ny = min(y * f, sh - 1);
		const size_t si = (size_t)ny * sw + nx;
		u.ddepth[i] = u.sdepth[si];
		u.dnormal[i * 3] = u.snormal[si * 3]; u.dnormal[i * 3 + 1] = u.snormal[si * 3 + 1]; u.dnormal[i * 3 + 2] = u.snormal[si * 3 + 2];
	}
}
__global__ void pm_nearest_up_f_kernel(const float* __restrict__ s, float* __restrict__ d, int sw, int sh, int dw, int dh) {
	const size_t n = (size_t)dw * dh;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % dw), y = (int)(i / dw);
		const int nx = min((int)floor((double)x * ((double)sw / (double)dw)), sw - 1);
		const int ny = min((int)floor((double)y * ((double)sh / (double)dh)), sh - 1);
		d[i] = s[(size_t)ny * sw + nx];
	}
}

// ignore mask of one view at a pyramid level: cv::resize(mask, size, INTER_NEAREST) of the level-0 mask (ImportIgnoreMask, DepthMap.cpp:309)
__global__ void pm_mask_level_kernel(const unsigned char* __restrict__ s, unsigned char* __restrict__ d, int sw, int sh, int dw, int dh) {
	const size_t n = (size_t)dw * dh;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % dw), y = (int)(i / dw);
		const int nx = min((int)floor((double)x * ((double)sw / (double)dw)), sw - 1);
		const int ny = min((int)floor((double)y * ((double)sh / (double)dh)), sh - 1);
		d[i] = s[(size_t)ny * sw + nx];
	}
}

// pm_math.h on the device (self-test hook)
__global__ void pm_math_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		float s, c;
		switch (kind) {
		case 0: o[i] = pm_expf(a[i]); break;
		case 1: o[i] = pm_acosf(a[i]); break;
		case 2: o[i] = pm_atan2f(a[i], b[i]); break;
		case 3: pm_sincosf(a[i], &s, &c); o[i] = s; break;
		case 4: pm_sincosf(a[i], &s, &c); o[i] = c; break;
		case 5: o[i] = pm_sqrtf(a[i]); break;
		case 7: o[i] = pm_hypot_d(a[i], b[i]); break;
		case 8: { float qx, qy; pm_div2(a[i], b[i] * 3.0f, b[i], &qx, &qy); o[i] = qx; } break;
		case 9: { float qx, qy; pm_div2(b[i], a[i], b[i] + a[i], &qx, &qy); o[i] = qy; } break;
		default: o[i] = a[i] / b[i]; break;
		}
	}
}
