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
#define PM_BLOCK 64    // one wave per workgroup (init kernel)
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
__device__ __forceinline__ pm_f4v pm_loadq(pm_gcf4 base, unsigned off) { return base[off]; }
#else
typedef const float4* pm_gcf4;
__device__ __forceinline__ pm_gcf4 pm_glob4(const float4* p) { return p; }
__device__ __forceinline__ float4 pm_loadq(pm_gcf4 base, unsigned off) { return base[off]; }
#endif
__device__ __forceinline__ pm_gcf pm_glob(const float* p) { return (pm_gcf)p; }
__device__ __forceinline__ pm_gf pm_globw(float* p) { return (pm_gf)p; }
#define PM_HW 4      // nSizeHalfWindow, DepthMap.h:277
#define PM_NT 25     // nTexels, DepthMap.h:281

struct PMSrcView {
	// "hot" block, 14 doubles: what every hypothesis evaluation reads of its source view.  The sweep kernels copy it (and the geometric block)
	// into LDS once per visit; the layout is the copy's contract (see PM_SRC_HOT / PM_SRC_GEO below).
	double Hl[9];         // K_j R_j R_0^T          (ViewData::Init, DepthMap.h:175-185)
	double Hm[3];         // K_j R_j (C_0 - C_j)
	int w, h;             // size of the source image at this level
	unsigned qBase, qPad; // first entry of this view's quad image in the level's quad buffer (PMTask::qArr); unused for a view with its own image size
	// geometric block, 14 doubles: transforms of the consistency term and the source view's depth-map (nullable; geometric pass) with its own
	// size: the map is addressed through cameraDepthMap (Tl..Tn), not through the image's camera (DepthMap.h:170-171, DepthMap.cpp:535-551)
	float Tl[9], Tm[3], Tr[9], Tn[3];
	const float* depth;
	int dw, dh;
	const float* img;     // source image at this pyramid level, row-major
	const float4* imgQ;   // anti-diagonal-major "quad" image: entry (u,v) = {I(u,v), I(u+1,v), I(u,v+1), I(u+1,v+1)} -- the four texels of a bilinear
	                      // sample in ONE 16-byte load (the window-less tap rows are bound by the number of vector-memory instructions, not by bytes:
	                      // the stride-2 taps of a patch never share texels, so the quad image is read at the same byte rate as the plain one)
};
#define PM_SRC_HOT 14     // doubles
#define PM_SRC_GEO 14
static_assert(offsetof(PMSrcView, Hm) == 72 && offsetof(PMSrcView, w) == 96 && offsetof(PMSrcView, qBase) == 104 && offsetof(PMSrcView, Tl) == 8 * PM_SRC_HOT
	&& offsetof(PMSrcView, depth) == 8 * PM_SRC_HOT + 96 && offsetof(PMSrcView, dw) == 8 * PM_SRC_HOT + 104 && offsetof(PMSrcView, img) == 8 * (PM_SRC_HOT + PM_SRC_GEO), "PMSrcView layout");
struct PMTask {           // one reference view at one pyramid level
	float* depth; float* normal; float* conf;
	const float* prior;   // nullable: low-resolution depth prior at this level
	const float* ref;     // reference image at this level, row-major
	const float* refS;    // reference image, anti-diagonal-major and FOLDED: texel (u,v) at ((u+v) mod w)*h + v -- anti-diagonal d and d + w share row d mod w of the array
	                      // without colliding (d holds v <= d, d + w holds v > d), so the copy is w*h floats; pixels of one anti-diagonal are adjacent (what a wave's lanes read)
	const unsigned char* mask; // nullable: ignore mask at this level, 0 = pixel is not estimated (DepthData::ApplyIgnoreMask + masked MapMatrix2ZigzagIdx)
	// tiled sweeps only (pmhip_set_sweep_tiles; null otherwise): depth / normal / conf as the running sweep found them -- what a pixel reads of a neighbour in ANOTHER tile
	const float* depthOld; const float* normalOld; const float* confOld;
	int w, h, nSrc, pad0;
	double Hr[9];         // K_0^-1
	int hrUpper;          // 1 if Hr[1] == Hr[3] == Hr[6] == Hr[7] == 0 exactly (zero-skew K): products with those vanish exactly
	int pad1;
	double fx, fy, cx, cy;
	float dMin, dMax, dMinSqr, dMaxSqr;
	uint32_t k0, k1base;  // Philox key: (seed, viewID*0x9E3779B1 + pass)
	const float4* qArr;   // the quad images of ALL scene views at this level, one allocation (view i at qArr + i * (w + h - 1) * h): the buffer the sweep kernels' tap rows index
	unsigned qCount, qPad; // its entries
	PMSrcView src[PM_MAX_SRC];
};
struct PMKParams {        // DepthEstimator ctor constants, DepthMap.cpp:397-406
	float smoothBonusDepth, smoothBonusNormal, smoothSigmaDepth, smoothSigmaNormal;
	float thMagnitudeSq, angle1Range, angle2Range, thConfSmall, thConfBig, thConfRand, thRobust;
	float thKeep, geoWeight, depthRatio;
	uint32_t nRandomIters;
};

enum { PM_STREAM_INIT = 0, PM_STREAM_RAND = 1, PM_STREAM_REFINE = 2 };

// Which pixels one sweep launch visits.  The pixels that take part in the estimation, [HW, w - HW) x [HW, h - HW), are cut into ntx x nty tiles of tw x th pixels (the last
// column / row of tiles is what is left); launch k visits, in every tile, the k-th anti-diagonal counted from the tile's corner the sweep starts at (dir 0: top left, 1: bottom
// right).  The reference's sweep is ONE tile (ntx == nty == 1): launch k is anti-diagonal x + y == 2 HW + k of the map (or the last one minus k), pixels of one anti-diagonal
// never read each other, so launches in stream order reproduce the sequential result.  With more tiles (opt-in, pmhip_set_sweep_tiles) a neighbour in another
// tile is read from PMTask::depthOld / normalOld / confOld, the maps as this sweep found them -- see oracle/pm_oracle.cpp Opt::tileW for the definition both sides follow.
struct PMStep { int dir, k, tw, th, ntx, nty, len; };   // len: pixels of a full tile in this launch = min(k, tw - 1, th - 1, tw + th - 2 - k) + 1.  The launch's pixels are numbered
                                                        // tile by tile, len per tile (a tile cut by the border has fewer: the rest of its numbers are idle), and waves take consecutive numbers
struct PMStepPix { bool active; int x, y; unsigned oldMask; };   // oldMask bit s: neighbour slot s lies in another tile
// pixel number p of the launch; slots: 0 (x+sgn,y), 1 (x,y+sgn), 2 (x-sgn,y), 3 (x,y-sgn) with sgn = dir == 0 ? -1 : 1.  TILED = false: the reference's sweep, one tile.
template <bool TILED>
__device__ __forceinline__ PMStepPix pm_step_pixel(const PMStep& st, int w, int h, int p) {
	PMStepPix r;
	int tile = 0, pi = p, tx = 0, ty = 0;
	if (TILED) { tile = p / st.len; pi = p - tile * st.len; tx = tile % st.ntx; ty = tile / st.ntx; }
	const int ox = PM_HW + tx * st.tw, oy = PM_HW + ty * st.th;
	const int twc = TILED ? min(st.tw, w - PM_HW - ox) : st.tw, thc = TILED ? min(st.th, h - PM_HW - oy) : st.th;   // this tile's own size
	const int lo = max(0, st.k - (thc - 1)), hi = min(twc - 1, st.k);
	const int l = lo + pi;                                   // distance from the starting corner along x; st.k - l along y
	r.active = ty < st.nty && l <= hi;
	const int lx = st.dir == 0 ? l : twc - 1 - l, ly = st.dir == 0 ? st.k - l : thc - 1 - (st.k - l);
	r.x = r.active ? ox + lx : PM_HW; r.y = r.active ? oy + ly : PM_HW;
	// dir 0 (sgn -1): slot 0 = left, 1 = top, 2 = right, 3 = bottom; dir 1: slot 0 = right, 1 = bottom, 2 = left, 3 = top
	const bool oL = lx == 0, oT = ly == 0, oR = lx == twc - 1, oB = ly == thc - 1;
	r.oldMask = !TILED ? 0u : st.dir == 0 ? ((oL ? 1u : 0u) | (oT ? 2u : 0u) | (oR ? 4u : 0u) | (oB ? 8u : 0u)) : ((oR ? 1u : 0u) | (oB ? 2u : 0u) | (oL ? 4u : 0u) | (oT ? 8u : 0u));
	return r;
}

#define PM_INF __builtin_huge_valf()
// the value must exist in a register at this point (device only; nothing for the host build of the emulator)
#if defined(__HIP_DEVICE_COMPILE__)
#define PM_OPAQUE(x) asm volatile("" : "+v"(x))
#define PM_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)   // nothing is scheduled across this point
#else
#define PM_OPAQUE(x) do {} while (0)
#define PM_SCHED_BARRIER() do {} while (0)
#endif

// Optional in-kernel phase timing (build with -DPM_PROFILE): lane 0 of every wave accumulates s_memtime deltas
// per phase into pm_prof[]; read back with pmhip_prof_get.  Phases: 0 setup (weights, neighbour gather, tiles),
// 1 hypothesis generation, 2 smoothness factors, 3 homography, 4 taps, 5 score epilogue, 6 aggregation+accept,
// 7 number of outer trips, 8 trips x active pixel-lanes, 9 waves, 10 tap rows served from LDS, 11 tap rows total; the sweep kernel splits
// phase 0 further: 12 = head of the visit (own + neighbour estimates, patch texels, weights), 13 = neighbour set-up + window placement, 0 = window staging.
#ifdef PM_PROFILE
__device__ unsigned long long pm_prof[16];
__device__ unsigned long long pm_hist[17];   // trips of pm_visit by the number of pixels of the wave that score a hypothesis in the trip (0..16)
struct PmProfAcc { unsigned long long a[16]; unsigned long long t; };
#define PM_PROF_ARG , PmProfAcc& _pa
#define PM_PROF_PASS , _pa
#define PM_PROF_DECL PmProfAcc _pa; for (int _i = 0; _i < 16; ++_i) _pa.a[_i] = 0; _pa.t = __builtin_readcyclecounter()
#define PM_TICK(i) do { const unsigned long long _n = __builtin_readcyclecounter(); _pa.a[i] += _n - _pa.t; _pa.t = _n; } while (0)
#define PM_COUNT(i, n) do { _pa.a[i] += (unsigned long long)(n); } while (0)
#define PM_HIST(n) do { const int _n = (n); if ((threadIdx.x & 63) == 0) atomicAdd(&pm_hist[_n > 16 ? 16 : _n], 1ull); } while (0)
#define PM_PROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int _i = 0; _i < 16; ++_i) atomicAdd(&pm_prof[_i], _pa.a[_i]); } while (0)
#else
#define PM_PROF_ARG
#define PM_PROF_PASS
#define PM_PROF_DECL do {} while (0)
#define PM_TICK(i) do {} while (0)
#define PM_COUNT(i, n) do {} while (0)
#define PM_HIST(n) do {} while (0)
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
	pm_sincosf(p0, &sx, &cx); pm_sincosf(p1, &sy, &cy);
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

// Dir2Normal by the lanes of a pixel's quad: even lanes take p0, odd lanes p1 -- one sincos per lane instead of two, the four values travel through the quad's DPP
// network.  Every lane of the quad must call it (they share the pixel's state, so they do).
__device__ __forceinline__ void pm_dir2normal_quad(float p0, float p1, int v, float& nx, float& ny, float& nz) {
	float s, c;
	pm_sincosf((v & 1) ? p1 : p0, &s, &c);
	const float sx = pm_quad_bcast<0>(s), cx = pm_quad_bcast<0>(c), sy = pm_quad_bcast<1>(s), cy = pm_quad_bcast<1>(c);
	nx = cx * sy; ny = sx * sy; nz = cy;
}

// ScorePixelImage for this lane's source view, DepthMap.cpp:465-564.
// sf[]: the (view-independent) smoothness factors of the up-to-4 close neighbours, in insertion
// order; exactly 1.f for a neighbour that does not exist or does not take part (DepthMap.cpp:524-533).
// SKEW selects the image layout the 100 bilinear taps read: the sweep kernel walks an anti-diagonal, so
// the footprints of the pixels of one wave lie along an anti-diagonal of the source image too; in the
// anti-diagonal-major copy those texels are contiguous (one or two 128-B lines per view instead of one
// line per pixel), which is what the vector L1 / texture-address unit is bound by here.  Same values.
// ComputeHomographyMatrix, DepthMap.h:414-423: (Hl + Hm * (n^T / (n.X0 * depth))) * Hr in double, cast to float
// hlm: the view's Hl (9) and Hm (3), contiguous as in PMSrcView's hot block -- in HBM (init kernel) or in the wave's LDS copy (sweep kernel)
// its view-independent part: r = n / ((n . X0) depth)
__device__ __forceinline__ void pm_homography_plane(double X0x, double X0y, float depth, float nx, float ny, float nz, double& r0, double& r1, double& r2) {
	const double n0 = (double)nx, n1 = (double)ny, n2 = (double)nz;
	const double ndx = (n0 * X0x + n1 * X0y) + n2;
	const double den = ndx * (double)depth;
	const double inv = (den == 0.0) ? 1e+14 : 1.0 / den; // INVERT, Types.h:1234
	r0 = n0 * inv; r1 = n1 * inv; r2 = n2 * inv;
}
// rpre: the plane part already formed for this hypothesis (the sweep kernel: once per trip for all of the lane's views, in LDS), or null
__device__ __forceinline__ void pm_homography(const double* hlm, const PMTask& t, double X0x, double X0y,
		float depth, float nx, float ny, float nz, float* H, const double* rpre = nullptr) {
	// the twelve matrix entries are requested first, together: one round trip (overlapping the division below) instead of one per row
	double Hl[9], Hm[3];
#pragma unroll
	for (int i = 0; i < 9; ++i) Hl[i] = hlm[i];
#pragma unroll
	for (int i = 0; i < 3; ++i) Hm[i] = hlm[9 + i];
	double r0, r1, r2;
	if (rpre) { r0 = rpre[0]; r1 = rpre[1]; r2 = rpre[2]; }
	else pm_homography_plane(X0x, X0y, depth, nx, ny, nz, r0, r1, r2);
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

#ifndef PM_WIDE_MINWAVES
#define PM_WIDE_MINWAVES 3   // waves per SIMD the one-wave-per-pixel kernel is compiled for
#endif

// ---- the level's quad images as ONE buffer ---------------------------------------------------------------------------------------------------------
// All scene images of a pyramid level live in one allocation (PMTask::qArr, 16-byte entries).  Described to the memory pipeline as a buffer (V#: base,
// stride 16, record count) a lane addresses its sample with a 32-bit ENTRY INDEX (buffer_load_dwordx4 ... idxen): the scaling by 16, the 64-bit add and the
// range check are done by the address unit instead of by v_mad_u64_u32 / v_lshl_add_u64 / four clamps per tap (about a fifth of a tap's VALU time, measured
// from the ISA: DESIGN.md 4.2).  An index outside the buffer returns zeros and touches nothing, so the optimistic rows need no clamp at all: whatever a tap
// outside the image or a garbage position fetched is never used (the hypothesis is flagged or redone).
#if defined(__HIP_DEVICE_COMPILE__)
typedef int pm_rsrc __attribute__((ext_vector_type(4)));
// built from wave-uniform values only (one task per workgroup); readfirstlane makes that provable to the compiler so that the descriptor sits in SGPRs
__device__ __forceinline__ pm_rsrc pm_make_rsrc(const void* base, unsigned count, unsigned stride = 16) {
	const unsigned long long b = (unsigned long long)base;
	pm_rsrc r;
	r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
	r.y = __builtin_amdgcn_readfirstlane((int)(((unsigned)(b >> 32) & 0xffffu) | (stride << 16)));   // base[47:32], stride in bytes
	r.z = __builtin_amdgcn_readfirstlane((int)count);                                                // records (index >= count: out of range)
	r.w = 0x00020000;                                                                                // data format 32 bit, no swizzle
	return r;
}
// The five samples of a tap row: issued back to back; the destinations are the compiler's registers but the loads are not in its s_waitcnt bookkeeping, so
// pm_bufwait5 (which names them all read-write) must sit between this and their first use (cdna_hip_programming.md 5.7, form (ii)).
__device__ __forceinline__ void pm_bufload5(pm_f4v& q0, pm_f4v& q1, pm_f4v& q2, pm_f4v& q3, pm_f4v& q4, unsigned i0, unsigned i1, unsigned i2, unsigned i3, unsigned i4, pm_rsrc r) {
	asm volatile("buffer_load_dwordx4 %0, %5, %10, 0 idxen\n\tbuffer_load_dwordx4 %1, %6, %10, 0 idxen\n\tbuffer_load_dwordx4 %2, %7, %10, 0 idxen\n\t"
	             "buffer_load_dwordx4 %3, %8, %10, 0 idxen\n\tbuffer_load_dwordx4 %4, %9, %10, 0 idxen"
		: "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4) : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "s"(r) : "memory");
}
// LEFT = vector-memory operations that may still be outstanding afterwards: loads return in order, so with the NEXT row's five loads issued behind these, LEFT = 5
// waits for exactly this row
template <int LEFT>
__device__ __forceinline__ void pm_bufwait5(pm_f4v& q0, pm_f4v& q1, pm_f4v& q2, pm_f4v& q3, pm_f4v& q4) {
	asm volatile("s_waitcnt vmcnt(%5)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4) : "i"(LEFT));
}
// a * b + c on the low 24 bits of a and b, as ONE full-rate instruction (left to itself the compiler splits two chained ones into two multiplies and a three-operand add)
__device__ __forceinline__ unsigned pm_mad24(int a, int b, unsigned c) { unsigned r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#else
// host build (the CPU emulator of the tests): the same semantics -- entry index, zeros outside
typedef float4 pm_f4v;
struct pm_rsrc { const float4* base; unsigned count; };
__device__ __forceinline__ pm_rsrc pm_make_rsrc(const void* base, unsigned count) { return pm_rsrc{(const float4*)base, count}; }
__device__ __forceinline__ void pm_bufload5(pm_f4v& q0, pm_f4v& q1, pm_f4v& q2, pm_f4v& q3, pm_f4v& q4, unsigned i0, unsigned i1, unsigned i2, unsigned i3, unsigned i4, pm_rsrc r) {
	const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
	q0 = i0 < r.count ? r.base[i0] : z; q1 = i1 < r.count ? r.base[i1] : z; q2 = i2 < r.count ? r.base[i2] : z; q3 = i3 < r.count ? r.base[i3] : z; q4 = i4 < r.count ? r.base[i4] : z;
}
template <int LEFT> __device__ __forceinline__ void pm_bufwait5(pm_f4v&, pm_f4v&, pm_f4v&, pm_f4v&, pm_f4v&) {}
__device__ __forceinline__ unsigned pm_mad24(int a, int b, unsigned c) { return (unsigned)(((unsigned)a & 0xffffffu) * ((unsigned)b & 0xffffffu)) + c; }   // wraps as the hardware's low 32 bits do
#endif

// what the buffer path of the tap rows needs, wave-uniform: the descriptor of the level's quad buffer
struct PMImgBuf { pm_rsrc rs; };
__device__ __forceinline__ PMImgBuf pm_make_imgbuf(const PMTask& t) {
	PMImgBuf b;
	b.rs = pm_make_rsrc(t.qArr, t.qCount);
	return b;
}

// One tap row (5 taps) of ScorePixelImage through global loads, with the reference's per-tap tests; from the row-major image (QUAD = false) or from the view's
// anti-diagonal-major quad image read as plain floats (QUAD = true: the clamped position's entry (lx + ly) * sh + ly holds exactly the four texels of the sample).
// X = position of the row's first tap.  The reference returns thRobust at the first tap that leaves the image (DepthMap.cpp:484-485).  Here a tap outside
// only raises a flag and its address is clamped, so there is no branch between taps: the 20 loads of the row are issued back to back and the sums of a flagged
// hypothesis are simply discarded -- identical result, no load ever depends on a previous load.  This is the GUARDED path: every position is the IEEE quotient
// whatever the operands (pm_div2).  The init kernel scores with it (one evaluation per pixel), the sweep kernels only redo a patch with it (0 of 5.9 M evaluations in
// the emulator's census) -- as four 4-byte loads per sample also there: with 16-byte loads the redo path alone took pm_sweep2_kernel<4,2> from 127 to 136 VGPRs, i.e.
// from four waves per SIMD to three (measured 1 % on the 100-view benchmark, profiles/r05_call4_ab_100.log).
template <bool QUAD>
__device__ __forceinline__ void pm_tap_row_global(const pm_gcf img, int sw, int sh, float h0, float h3, float h6, float X0, float X1, float X2,
		const float2* wrow, float& sum, float& sumSq, float& num, bool& oob)
{
	const int lxMax = sw - 2, lyMax = sh - 2;
	float fxs[5], fys[5];
	unsigned offs[5];   // float offsets fit 32 bits (an image is < 2^32 floats, a quad image < 2^30 entries)
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		float ptx, pty;
		pm_div2(X0, X1, X2, &ptx, &pty); // == X0 / X2, X1 / X2 (TPoint2(Point3), Types.h:1291)
		oob = oob || !pm_inside1(ptx, pty, sw, sh);
		// TImage::sample, libs/Common/Types.inl:2273-2281
		int lx = (int)ptx, ly = (int)pty;
		fxs[j] = ptx - (float)lx; fys[j] = pty - (float)ly;
		lx = min(max(lx, 0), lxMax); ly = min(max(ly, 0), lyMax);
		offs[j] = QUAD ? 4u * ((unsigned)(lx + ly) * (unsigned)sh + (unsigned)ly) : (unsigned)ly * (unsigned)sw + (unsigned)lx;
		X0 += h0; X1 += h3; X2 += h6;
	}
	float v00[5], v01[5], v10[5], v11[5];
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		const pm_gcf p = img + offs[j];
		if (QUAD) { v00[j] = p[0]; v01[j] = p[1]; v10[j] = p[2]; v11[j] = p[3]; }   // texels (lx,ly), (lx+1,ly), (lx,ly+1), (lx+1,ly+1)
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

// The same row, OPTIMISTIC and branch-free, from the anti-diagonal-major quad image (one 16-byte entry = the four texels of a bilinear sample): the divisions are
// the unguarded reciprocal refinement (pm_div2_inrange), nothing is tested between the taps, the sums are committed unconditionally.  What the patch records
// instead (PMTapRange, below): position and depth of its four CORNER taps.  The caller decides afterwards (pm_score_view) whether the optimistic sums are the
// guarded path's, whether the reference would have left the patch at a tap outside the image, or whether the patch has to be redone through the guarded path.
// BUF: the sample is addressed as entry index qbase + (lx + ly) * sh + ly of the level's buffer (no clamp: out of range reads zeros, see above); otherwise through the
// view's own pointer with clamped coordinates (views that carry their own image size live outside the level's buffer).
struct PMRowPos { float ptx[5], pty[5]; };
struct PMRowQ { pm_f4v q0, q1, q2, q3, q4; };
// What a patch records for the decision afterwards, from its FOUR CORNER taps only (bit patterns, pm_f2i): zlo / zhi = extremes of the depth, plo = the smallest x or y,
// pxhi / pyhi = the largest x / y.  z: a row's z moves by the same float increment from tap to tap, a row's first z by the same increment from row to row, and x (+) h is
// monotone in x, so the extremes of z over the whole patch are exactly among the corners.  Positions: see pm_score_view.
struct PMTapRange { int zlo, zhi, plo, pxhi, pyhi; };
__device__ __forceinline__ void pm_range_corner(PMTapRange& rg, float px, float py, float z) {
	const int ix = pm_f2i(px), iy = pm_f2i(py), iz = pm_f2i(z);
	rg.plo = min(rg.plo, min(ix, iy)); rg.pxhi = max(rg.pxhi, ix); rg.pyhi = max(rg.pyhi, iy);
	rg.zlo = min(rg.zlo, iz); rg.zhi = max(rg.zhi, iz);
}
// positions of a row's five taps, their samples requested; nothing waits here.  CORNERS: the row is the patch's first one -- its taps 0 and 4 are recorded (the last row's
// are recorded by pm_taps_fast from the positions the row leaves behind, so that the loop over the rows has one body).
// Entry index of sample (lx, ly) of the view whose image starts at entry qbase: qbase + (lx + ly) * sh + ly == lx * sh + (ly * (sh + 1) + qbase): two v_mad_u32_u24.
template <bool BUF, bool CORNERS>
__device__ __forceinline__ void pm_row_issue(const PMImgBuf& rs, unsigned qbase, const pm_gcf4 imgQ, int sw, int sh, float h0, float h3, float h6, float X0, float X1, float X2,
		PMRowPos& p, PMRowQ& q, PMTapRange& rg)
{
	unsigned idx[5];
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		pm_div2_inrange(X0, X1, X2, &p.ptx[j], &p.pty[j]);
		const int lx = (int)p.ptx[j], ly = (int)p.pty[j];
		if (BUF) idx[j] = pm_mad24(lx, sh, pm_mad24(ly, sh + 1, qbase));
		else { const int lxc = min(max(lx, 0), sw - 2), lyc = min(max(ly, 0), sh - 2); idx[j] = (unsigned)(lxc + lyc) * (unsigned)sh + (unsigned)lyc; }
		if (CORNERS && (j == 0 || j == 4)) pm_range_corner(rg, p.ptx[j], p.pty[j], X2);
		X0 += h0; X1 += h3; X2 += h6;
	}
	if (BUF) pm_bufload5(q.q0, q.q1, q.q2, q.q3, q.q4, idx[0], idx[1], idx[2], idx[3], idx[4], rs.rs);
	else { q.q0 = pm_loadq(imgQ, idx[0]); q.q1 = pm_loadq(imgQ, idx[1]); q.q2 = pm_loadq(imgQ, idx[2]); q.q3 = pm_loadq(imgQ, idx[3]); q.q4 = pm_loadq(imgQ, idx[4]); }
}
// the row's samples have arrived (LEFT younger loads may still be under way): bilinear values and the three running sums, in the reference's order
template <bool BUF, int LEFT>
__device__ __forceinline__ void pm_row_consume(const PMRowPos& p, PMRowQ& q, const float2* wrow, float& sum, float& sumSq, float& num)
{
	float t00[5], t01[5], t10[5], t11[5];
	if (BUF) pm_bufwait5<LEFT>(q.q0, q.q1, q.q2, q.q3, q.q4);
	const pm_f4v qq[5] = {q.q0, q.q1, q.q2, q.q3, q.q4};
#pragma unroll
	for (int j = 0; j < 5; ++j) { t00[j] = qq[j].x; t01[j] = qq[j].y; t10[j] = qq[j].z; t11[j] = qq[j].w; }
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		const float fx = pm_fract_pos(p.ptx[j]), fx1 = 1.f - fx, fy = pm_fract_pos(p.pty[j]), fy1 = 1.f - fy;   // == ptx - (float)(int)ptx for the positions the row is accepted with (>= 1)
		const float v = (t00[j] * fx1 + t01[j] * fx) * fy1 + (t10[j] * fx1 + t11[j] * fx) * fy;
		const float2 pw = wrow[j];
		const float vw = v * pw.x;
		sum += vw;
		sumSq += v * vw;
		num += v * pw.y;
	}
}
// The 25 taps of a patch as a two-deep software pipeline over its five rows: the samples of row i + 1 are requested before row i is consumed, so a wave has ten
// 16-byte loads in flight instead of five and waits for memory three times per patch instead of five (a wave-visit is a chain of ~80 such round trips and the launch
// of a diagonal lasts as long as one wave-visit: DESIGN.md 4.2).  Two register sets (A, B) alternate; the rows are written out so that no set is ever copied.
template <bool BUF>
__device__ __forceinline__ void pm_taps_fast(const PMImgBuf& rs, unsigned qbase, const pm_gcf4 imgQ, int sw, int sh, const float* H, float bX0, float bX1, float bX2,
		const float2* wts, float& sum, float& sumSq, float& num, PMTapRange& rg)
{
	PMRowPos pa, pb; PMRowQ qa, qb;
	pm_row_issue<BUF, true>(rs, qbase, imgQ, sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, pa, qa, rg);
	// rows (0,1), (2,3) as one loop body with the A / B sets swapping roles (no copies), then row 4.  The scheduling barriers keep the compiler from pulling the next
	// row's divisions above the current row's arithmetic (which is what it does with the whole patch unrolled: every position of the patch live at once, 600 B of scratch)
#pragma unroll 1
	for (int i = 0; i < 4; i += 2) {
		bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
		pm_row_issue<BUF, false>(rs, qbase, imgQ, sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, pb, qb, rg);
		PM_SCHED_BARRIER();
		pm_row_consume<BUF, 5>(pa, qa, wts + i * 5, sum, sumSq, num);
		PM_SCHED_BARRIER();
		bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
		pm_row_issue<BUF, false>(rs, qbase, imgQ, sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, pa, qa, rg);
		PM_SCHED_BARRIER();
		pm_row_consume<BUF, 5>(pb, qb, wts + (i + 1) * 5, sum, sumSq, num);
		PM_SCHED_BARRIER();
	}
	// the last row's corners: its positions are still in `pa`, its z values are its first one (bX2) and that after the row's four increments (the same float additions)
	pm_range_corner(rg, pa.ptx[0], pa.pty[0], bX2);
	pm_range_corner(rg, pa.ptx[4], pa.pty[4], (((bX2 + H[6]) + H[6]) + H[6]) + H[6]);
	pm_row_consume<BUF, 0>(pa, qa, wts + 20, sum, sumSq, num);
}

// ScorePixelImage for this lane's source view, DepthMap.cpp:465-564.
// sf0..3: the (view-independent) smoothness factors of the up-to-4 close neighbours, in insertion order; exactly 1.f for a neighbour that does not exist or does
// not take part (DepthMap.cpp:524-533).
// MODE 0: guarded tap rows from the row-major image (init kernel: no anti-diagonal to exploit, one evaluation per pixel);
// MODE 1: optimistic rows from the view's quad image through its own pointer;  MODE 2: optimistic rows from the level's quad buffer (pm_tap_row_fast<true>).
// PF: the pixel's low-resolution prior and its blend factor exp(normSq0 * sigma) sit in the spare 26th entry of the pixel's weight row in LDS (written once per
// visit by the sweep kernels) instead of in two registers that are live across the whole hypothesis loop.
// hot / geoTab: the hot and geometric blocks of `s` (PMSrcView), in HBM (init kernel) or in the wave's LDS copy (sweep kernels); the image size and the view's
// first entry in the level's quad buffer travel with the homography entries.
// EARLY: test two opposite corner taps before any load (pm_sweep2_kernel, whose launches fill the GPU and run into the texture-address unit; the speculative kernels of the
// short launches are bound by a visit's dependent chain, where the test's two extra divisions cost 3 %: profiles/r06_call13/ab_13.log)
template <bool GEO, int MODE, bool PF = false, bool EARLY = false>
__device__ __forceinline__ float pm_score_view(const PMSrcView& s, const PMTask& t, const PMKParams& kp,
		int x, int y, double X0x, double X0y, float normSq0, float sumW, const float2* wts,
		float depth, float nx, float ny, float nz,
		float sf0, float sf1, float sf2, float sf3, float prior,
		const double* hot, const double* geoTab, const PMImgBuf& rs PM_PROF_ARG, const double* rpre = nullptr)
{
	const int sw = ((const int*)(hot + 12))[0], sh = ((const int*)(hot + 12))[1];
	float H[9];
	pm_homography(hot, t, X0x, X0y, depth, nx, ny, nz, H, rpre);
	PM_TICK(3);
	const float px = (float)(x - PM_HW), py = (float)(y - PM_HW);
	const float X0 = H[0] * px + H[1] * py + H[2];
	const float X1 = H[3] * px + H[4] * py + H[5];
	const float X2 = H[6] * px + H[7] * py + H[8];
	float bX0 = X0, bX1 = X1, bX2 = X2;
#pragma unroll
	for (int i = 0; i < 9; ++i) H[i] *= 2.f; // nSizeStep
	float sum = 0.f, sumSq = 0.f, num = 0.f;
	bool oob = false;
	if (MODE == 0) {
#pragma unroll 1
		for (int i = 0; i < 5; ++i) {
			pm_tap_row_global<false>(pm_glob(s.img), sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, wts + i * 5, sum, sumSq, num, oob);
			bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
		}
	} else {
		// |x|, |y| < 1e18 and |z| < 2^40 for every tap of the patch, from the first tap and the step sizes (8 steps along either axis at most)
		const bool sane = pm_fabsf(X0) + 8.f * (pm_fabsf(H[0]) + pm_fabsf(H[1])) < 5e17f && pm_fabsf(X1) + 8.f * (pm_fabsf(H[3]) + pm_fabsf(H[4])) < 5e17f
			&& pm_fabsf(X2) + 8.f * (pm_fabsf(H[6]) + pm_fabsf(H[7])) < 5e11f;
		// Two opposite corner taps first, without loads: the texture-address unit, which bounds the sweep kernels, is paid by the active lane, and a source view
		// that does not see the pixel at all -- a fifth of the (pixel, view) pairs of the benchmark's grid of cameras -- would otherwise gather all 25 samples for nothing.
		// Tap (0,0) is the reference's first tap and tap (4,4) its last; their positions are formed by the very additions the rows below perform (four row steps, then four tap
		// steps), and with z inside [2^-40, 2^40] and sane start values the quotient is the IEEE one (pm_div2_inrange): if either lies outside the image the reference has
		// returned thRobust at that tap at the latest (DepthMap.cpp:484-485), whatever the other taps do.  A corner whose z is out of range decides nothing here.
		if (EARLY && sane) {
			bool outside = false;
			float cx, cy;
			if (bX2 >= 9.094947e-13f && bX2 <= 1.0995116e12f) { pm_div2_inrange(bX0, bX1, bX2, &cx, &cy); outside = !pm_inside1(cx, cy, sw, sh); }
			float c0 = bX0, c1 = bX1, c2 = bX2;
#pragma unroll
			for (int i = 0; i < 4; ++i) { c0 += H[1]; c1 += H[4]; c2 += H[7]; }
#pragma unroll
			for (int j = 0; j < 4; ++j) { c0 += H[0]; c1 += H[3]; c2 += H[6]; }
			if (c2 >= 9.094947e-13f && c2 <= 1.0995116e12f) { pm_div2_inrange(c0, c1, c2, &cx, &cy); outside = outside || !pm_inside1(cx, cy, sw, sh); }
			if (outside) return kp.thRobust;
		}
		const unsigned qbase = ((const unsigned*)(hot + 13))[0];
		const pm_gcf4 imgQ = pm_glob4(s.imgQ);
		const float rX0 = bX0, rX1 = bX1, rX2 = bX2;
		PMTapRange rg = {0x7fffffff, (int)0x80000000, 0x7fffffff, (int)0x80000000, (int)0x80000000};
		pm_taps_fast<MODE == 2>(rs, qbase, imgQ, sw, sh, H, bX0, bX1, bX2, wts, sum, sumSq, num, rg);
		// (1) exact: 2^-40 <= z <= 2^40 on the whole patch (the extremes of z are among the corners, PMTapRange) and `sane` start values: every quotient of the patch was
		//     the correctly rounded one, so the sums are pm_tap_row_global's and a tap's position is the one the reference tests.  Otherwise (0 of 38 M evaluations in the
		//     emulator's census) the whole patch is redone through the guarded path.
		// (2) a corner fails isInsideWithBorder<1>: the reference returns thRobust at that tap at the latest -- outside.
		// (3) every corner inside by at least delta = 2^-17 max(w, h) and zhi <= 2 zlo: every tap is inside.  In real arithmetic on the float start values and steps the taps
		//     T(i,j) = b + i c + j r are affine in (i, j) with z > 0, so the positions T.xy / T.z lie in the convex hull of the four corner positions.  A float tap differs
		//     from T by the roundings of its <= 8 additions, each <= 2^-24 of a value that is itself a tap: |dx| <= 8 2^-24 P zhi, |dz| <= 8 2^-24 zhi with P = max(w, h) >=
		//     every position; the quotient adds one rounding: |position error| <= (16 zhi / zlo + 1) 2^-24 P <= 33 2^-24 P = eps.  A tap therefore lies within 2 eps =
		//     66 2^-24 P < delta = 128 2^-24 P of the corners' range (0.015 px at 1920).
		// (4) otherwise -- a corner within delta of the border band, or a plane that almost contains the viewing ray: the positions of all 25 taps again (the same float
		//     operations, no loads), tested one by one as the reference does.
		const bool exact = sane && rg.zlo >= pm_f2i(9.094947e-13f) && rg.zhi <= pm_f2i(1.0995116e12f);
		const bool cornersIn = rg.plo >= pm_f2i(1.f) && rg.pxhi <= pm_f2i((float)(sw - 2)) && rg.pyhi <= pm_f2i((float)(sh - 2));
		const float delta = (float)max(sw, sh) * 7.62939453125e-06f;   // 2^-17
		const bool allIn = rg.plo >= pm_f2i(1.f + delta) && rg.pxhi <= pm_f2i((float)(sw - 2) - delta) && rg.pyhi <= pm_f2i((float)(sh - 2) - delta)
			&& (unsigned)rg.zhi - (unsigned)rg.zlo <= 0x00800000u;   // zhi <= 2 zlo on the bit patterns of positive floats (one exponent step)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(PM_DEBUG_REDO)
		{ static unsigned long long c[3]; static bool reg = false; if (!reg) { reg = true; atexit([] { fprintf(stderr, "optimistic evaluations %llu, positions rechecked %llu, redone %llu\n", c[0], c[1], c[2]); }); }
		  c[0]++; if (exact && cornersIn && !allIn) c[1]++; if (!exact) c[2]++; }
#endif
		if (exact) {
			oob = !cornersIn;
			if (cornersIn && !allIn) {
				float cX0 = rX0, cX1 = rX1, cX2 = rX2;
#pragma unroll 1
				for (int i = 0; i < 5; ++i) {
					float X0t = cX0, X1t = cX1, X2t = cX2;
#pragma unroll 1
					for (int j = 0; j < 5; ++j) {
						float ptx, pty;
						pm_div2_inrange(X0t, X1t, X2t, &ptx, &pty);
						oob = oob || !pm_inside1(ptx, pty, sw, sh);
						X0t += H[0]; X1t += H[3]; X2t += H[6];
					}
					cX0 += H[1]; cX1 += H[4]; cX2 += H[7];
				}
			}
		} else {   // the whole patch through the guarded path
			sum = 0.f; sumSq = 0.f; num = 0.f;
			bX0 = rX0; bX1 = rX1; bX2 = rX2;
#pragma unroll 1
			for (int i = 0; i < 5; ++i) {
				pm_tap_row_global<true>(pm_glob((const float*)s.imgQ), sw, sh, H[0], H[3], H[6], bX0, bX1, bX2, wts + i * 5, sum, sumSq, num, oob);
				bX0 += H[1]; bX1 += H[4]; bX2 += H[7];
			}
		}
	}
	PM_TICK(4);
	if (oob) return kp.thRobust;
	const float normSq1 = sumSq - (sum * sum) / sumW;
	const float nrmSq = normSq0 * normSq1;
	if (nrmSq <= 1e-16f) return kp.thRobust;
	const float ncc = pm_clampf(num / pm_sqrtf(nrmSq), -1.f, 1.f);
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
		if (sdepth != nullptr) {
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

// row of anti-diagonal d in the folded anti-diagonal-major image (PMTask::refS): d mod w, for 0 <= d <= w + h - 2
__device__ __forceinline__ int pm_fold(int d, int w) { if (d >= w) { d -= w; if (d >= w) d %= w; } return d; }
// FillPixelPatch, DepthMap.cpp:422-462: cooperative weights into LDS; returns normSq0, sumW.
// Must be called by every thread of the workgroup (contains __syncthreads()).
template <int G, bool SKEW>
__device__ __forceinline__ void pm_fill_patch(const PMTask& t, bool inb, int x, int y, int v, float2* wts, float& normSq0, float& sumW) {
	const float sigmaColor = -1.f / (2.f * (0.1f * 0.1f));
	const float sigmaSpatial = -1.f / (2.f * 9.f);
	if (inb) {
		const pm_gcf refS = pm_glob(t.refS), ref = pm_glob(t.ref);
		const float colCenter = SKEW ? refS[(size_t)pm_fold(x + y, t.w) * t.h + y] : ref[(size_t)y * t.w + x];
		// the texels of this lane's taps are requested together (one memory round trip at the head of the visit, not one per tap)
		constexpr int NK = (PM_NT + G - 1) / G;
		float Is[NK];
#pragma unroll
		for (int q = 0; q < NK; ++q) {
			const int k = v + q * G, kk = k < PM_NT ? k : v % PM_NT;   // (a lane without a tap re-reads one that exists: no address outside the patch)
			const int i = (kk / 5) * 2 - PM_HW, j = (kk % 5) * 2 - PM_HW;
			Is[q] = SKEW ? refS[(size_t)pm_fold(x + j + y + i, t.w) * t.h + (y + i)] : ref[(size_t)(y + i) * t.w + (x + j)];
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
			wts[k] = make_float2(pm_expf(wColor + wSpatial), I);
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
// PM_INIT_MODE: how its one evaluation per pixel reads the source images -- 2 (default): the sweep kernels' optimistic rows from the level's quad buffer (one 16-byte load per
// sample, 34 VALU instructions per tap); 0: guarded tap rows from the row-major images (four 4-byte loads per sample, IEEE divisions and the image test per tap, ~60 per tap).
// With 100 views resident the pass is bound by VALU issue (85 % of the SIMD cycles, profiles/r06_call1): mode 2 is +0.8 % on the benchmark (profiles/r06_call4/ab_100.log).
// Batches that read source views of their own image size (outside the level's quad buffer) use mode 0.
#ifndef PM_INIT_MODE
#define PM_INIT_MODE 2
#endif
// G lanes per pixel, VPL source views per lane (lane v scores views v, v + G, ...): what a pixel's lanes compute redundantly -- the patch sums, the draw, the plane -- is
// 40 % of a wave's instructions with one view per lane (profiles/r06_final_pmc: 1 710 VALU instructions per wave of 8 pixels x 8 lanes), so batches with more than four
// source views give a pixel half the lanes and each lane two views.
template <int G, bool GEO, int MODE, int VPL = 1>
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
	float sc = PM_INF, sc2 = PM_INF;   // the lane's two smallest view scores
	PM_PROF_DECL;
	const PMImgBuf rs = MODE == 2 ? pm_make_imgbuf(t) : PMImgBuf();
#pragma unroll 1
	for (int u = 0; u < VPL; ++u) {
		const int vw = v + u * G;
		if (vw < t.nSrc) {
			const float s1 = pm_score_view<GEO, MODE>(t.src[vw], t, kp, x, y, X0x, X0y, normSq0, sumW, s_w[g], depth, nx, ny, nz, 1.f, 1.f, 1.f, 1.f, prior, t.src[vw].Hl, (const double*)t.src[vw].Tl,
				rs PM_PROF_PASS);
			if (s1 < sc) { sc2 = sc; sc = s1; } else if (s1 < sc2) sc2 = s1;
		}
	}
	const float conf = pm_aggregate<G>(sc, t.nSrc, kp.thRobust, sc2);
	if (v == 0) { gDepth[idx] = depth; gNormal[idx * 3] = nx; gNormal[idx * 3 + 1] = ny; gNormal[idx * 3 + 2] = nz; gConf[idx] = conf; }
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
template <bool GEO, bool BUF>
__global__ __launch_bounds__(64, PM_WIDE_MINWAVES) void pm_sweep_wide_kernel(const PMTask* __restrict__ tasks, PMKParams kp, int dir, int d, int xlo, int count, uint32_t pass) {
	constexpr int G = 8;
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	PM_PROF_DECL;
	__shared__ float2 s_w[PM_NT + 1];
	__shared__ double s_src[G * NBD];
	const PMTask& t = tasks[blockIdx.y];
	const PMImgBuf rs = pm_make_imgbuf(t);
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
			sc = pm_score_view<GEO, BUF ? 2 : 1, true>(t.src[v], t, kp, x, y, X0x, X0y, normSq0, sumW, s_w, hd, hnx, hny, hnz, sf[0], sf[1], sf[2], sf[3], 0.f,
				hot, hot + PM_SRC_HOT, rs PM_PROF_PASS);
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

// tiled sweeps (PMStep): the maps of the group's views as the sweep that follows finds them
__global__ void pm_snapshot_kernel(const PMTask* __restrict__ tasks, size_t n) {
	const PMTask& t = tasks[blockIdx.y];
	float* dO = const_cast<float*>(t.depthOld); float* nO = const_cast<float*>(t.normalOld); float* cO = const_cast<float*>(t.confOld);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		dO[i] = t.depth[i]; cO[i] = t.conf[i];
		nO[i * 3] = t.normal[i * 3]; nO[i * 3 + 1] = t.normal[i * 3 + 1]; nO[i * 3 + 2] = t.normal[i * 3 + 2];
	}
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
// folded anti-diagonal-major copy of nImg row-major images (PMTask::refS): dst[img][((u+v) mod w)*h + v] = src[img][v*w + u]
__global__ void pm_skew_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, int nImg) {
	const size_t n = (size_t)w * h * nImg;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int u = (int)(i % w), v = (int)((i / w) % h); const size_t im = i / ((size_t)w * h);
		dst[im * (size_t)w * h + (size_t)((u + v) % w) * h + v] = src[i];
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
