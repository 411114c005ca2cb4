// sgm_kernels_sub.hip -- the three Match kernels for the narrow, ragged disparity ranges of the tSGM loop: a wavefront is split
// into PW = 64 / LP sub-groups of LP lanes; a sub-group owns one pixel (WTA), one pair of pixels (cost volume) or one line (path aggregation) and loops
// over ceil(nD / LP) chunks of its disparity range, so a pixel with <= LP disparities costs one pass and the few wide ones cost more.  Same integer /
// float arithmetic, same order per pixel as the wide kernels of sgm_kernels.hip (and the reference, SemiGlobalMatcher.cpp:874-1301); only the mapping of
// work to lanes differs.  Included by sgm_engine.hip after sgm_kernels.hip.
#pragma once
#include "sgm_kernels.hip"

// minimum over the LP lanes of a sub-group (LP = 8, 16 or 32, aligned), returned in every lane: quad butterflies, then mirrors inside 8 and 16 lanes (DPP)
template <int LP>
__device__ __forceinline__ int sgm_sub_min(int v) {
	v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));          // quad_perm [1,0,3,2]
	v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));          // quad_perm [2,3,0,1]
	v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));         // row_half_mirror: lane i <-> 7-i inside 8 lanes
	if (LP >= 16) v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false)); // row_mirror: lane i <-> 15-i inside 16 lanes
	if (LP >= 32) v = min(v, __shfl_xor(v, 16, 64));                                     // the other row of the pair (LDS crossbar)
	return v;
}

// ---- winner-take-all: one pixel per sub-group ------------------------------------------------------------------------------------------------
template <int LP>
__global__ __launch_bounds__(256) void sgm_wta_sub_kernel(const SGMPixel* __restrict__ pixels, const unsigned short* __restrict__ accums,
		long nPix, short* __restrict__ disp, unsigned short* __restrict__ cost) {
	constexpr int PW = 64 / LP;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane / LP, kk = lane % LP;
	const long pix = ((long)blockIdx.x * 4 + wave) * PW + sub;
	SGMPixel px; px.idx = 0; px.minDisp = 0; px.maxDisp = 0; px.pad = 0;
	const bool have = pix < nPix;
	if (have) px = pixels[pix];
	const int nD = px.maxDisp - px.minDisp;
	// first minimum == lexicographic minimum of (value, index)
	unsigned key = 0xFFFFFFFFu;
	for (int k = kk; k < nD; k += LP) key = min(key, ((unsigned)accums[px.idx + k] << 16) | (unsigned)k);
	// unsigned keys < 2^31 would allow the signed DPP minimum; they are not (0xFFFF....): flip the top bit around it
	key = (unsigned)sgm_sub_min<LP>((int)(key ^ 0x80000000u)) ^ 0x80000000u;
	if (have && kk == 0) {
		if (nD <= 0) { disp[pix] = px.minDisp; cost[pix] = 0xFFFF; }
		else { disp[pix] = (short)(px.minDisp + (int)(key & 0xFFFFu)); cost[pix] = (unsigned short)(key >> 16); }
	}
}

// ---- cost volume: one pair of horizontally adjacent pixels per sub-group ------------------------------------------------------------------------
template <int LP>
__global__ __launch_bounds__(256, 2) void sgm_cost_sub_kernel(const unsigned char* __restrict__ colorL, const float* __restrict__ grayL,
		const float* __restrict__ grayR, int w, int h, int vw, int vh, const SGMPixel* __restrict__ pixels,
		const float4* __restrict__ setup, unsigned char* __restrict__ costs) {
	constexpr int PW = 64 / LP;
	__shared__ float4 s_w[4][PW][SGM_NT + 1];                          // (wA, wB, wA*(vA-meanA), wB*(vB-meanB)) per tap, per pair of the wave
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane / LP, kk = lane % LP;
	const int ppr = (vw + 1) >> 1;                                     // pairs per row
	const long nPairs = (long)ppr * vh;
	const long pair0 = ((long)blockIdx.x * 4 + wave) * PW;
	if (pair0 >= nPairs) return;                                       // (no workgroup barrier below: each wave owns its LDS rows)
	// weights of the PW pairs of this wave: PW * 49 entries spread over the 64 lanes
	for (int e = lane; e < PW * SGM_NT; e += 64) {
		const int s = e / SGM_NT, t = e % SGM_NT;
		const long pr = pair0 + s;
		float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
		if (pr < nPairs) {
			const int row = (int)(pr / ppr), colA = (int)(pr % ppr) * 2;
			const long pixA = (long)row * vw + colA;
			const bool hasB = colA + 1 < vw;
			const SGMPixel pa = pixels[pixA];
			const bool onA = pa.maxDisp > pa.minDisp;
			bool onB = false;
			if (hasB) { const SGMPixel pb = pixels[pixA + 1]; onB = pb.maxDisp > pb.minDisp; }
			const int ux = colA + SGM_HW, uy = row + SGM_HW, i = t / 7 - SGM_HW, j = t % 7 - SGM_HW;
			if (onA) { o.x = sgm_weight(colorL, w, ux, uy, i, j); o.z = o.x * (grayL[(size_t)(uy + i) * w + (ux + j)] - setup[pixA].y); }
			if (onB) { o.y = sgm_weight(colorL, w, ux + 1, uy, i, j); o.w = o.y * (grayL[(size_t)(uy + i) * w + (ux + 1 + j)] - setup[pixA + 1].y); }
		}
		s_w[wave][s][t] = o;
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const long pair = pair0 + sub;
	const bool have = pair < nPairs;
	const int row = have ? (int)(pair / ppr) : 0, colA = have ? (int)(pair % ppr) * 2 : 0;
	const long pixA = (long)row * vw + colA;
	const bool hasB = have && colA + 1 < vw;
	SGMPixel pxA, pxB; pxA.idx = 0; pxA.minDisp = 0; pxA.maxDisp = 0; pxA.pad = 0; pxB = pxA;
	if (have) pxA = pixels[pixA];
	if (hasB) pxB = pixels[pixA + 1];
	const int nDA = pxA.maxDisp > pxA.minDisp ? pxA.maxDisp - pxA.minDisp : 0;
	const int nDB = pxB.maxDisp > pxB.minDisp ? pxB.maxDisp - pxB.minDisp : 0;
	const int ux = colA + SGM_HW, uy = row + SGM_HW;
	const float4 sA = have ? setup[pixA] : make_float4(1.f, 0.f, 0.f, 0.f);
	const float4 sB = hasB ? setup[pixA + 1] : make_float4(1.f, 0.f, 0.f, 0.f);
	const int nDmax = nDA > nDB ? nDA : nDB;
#pragma unroll 1
	for (int k = kk; k < nDmax; k += LP) {
		asm volatile("" ::: "memory");
		const int dA = pxA.minDisp + k, dB = pxB.minDisp + k;
		const bool actA = k < nDA, actB = k < nDB;
		const bool inA = actA && !(ux - SGM_HW + dA < 0 || ux + SGM_HW + dA >= w);       // all taps inside the right image (:954-957)
		const bool inB = actB && !(ux + 1 - SGM_HW + dB < 0 || ux + 1 + SGM_HW + dB >= w);
		const int cA = inA ? ux + dA : SGM_HW, cB = inB ? ux + 1 + dB : SGM_HW;
		sgm_v2f sum = {0.f, 0.f}, sumSq = {0.f, 0.f}, nom = {0.f, 0.f};
		int n = 0;
		for (int i = -SGM_HW; i <= SGM_HW; ++i) {
			const float* rowA = grayR + (size_t)(uy + i) * w + cA;
			const float* rowB = grayR + (size_t)(uy + i) * w + cB;
#pragma unroll
			for (int j = -SGM_HW; j <= SGM_HW; ++j) {
				const sgm_v2f f = {rowA[j], rowB[j]};
				const float4 pw = s_w[wave][sub][n++];
				const sgm_v2f pww = {pw.x, pw.y}, pwt = {pw.z, pw.w};
				const sgm_v2f fw = f * pww;
				sum += fw; sumSq += f * fw; nom += f * pwt;
			}
		}
		if (actA) costs[pxA.idx + (unsigned)k] = inA ? sgm_cost_of(sum.x, sumSq.x, nom.x, sA.x, sA.z) : (unsigned char)255;
		if (actB) costs[pxB.idx + (unsigned)k] = inB ? sgm_cost_of(sum.y, sumSq.y, nom.y, sB.x, sB.z) : (unsigned char)255;
	}
}

// ---- path aggregation: one line per sub-group -----------------------------------------------------------------------------------------------------
// accums(d) += L(d): as sgm_accumulate, with "first lane of the instruction" = first lane of the sub-group (kk == 0) -- the low-half partner of that
// entry belongs to the previous chunk of the same pixel (another instruction) or does not exist.
template <int LP>
__device__ __forceinline__ void sgm_accumulate_sub(unsigned* wordsBase, unsigned par, int k, int kk, int nD, int L) {
	const int Lnext = __builtin_amdgcn_update_dpp(L, L, 0x130, 0xf, 0xf, false);   // wave_shl:1 (DPP): entry k+1 sits in lane+1 of the same sub-group whenever it is used (pair)
	const unsigned e = ((unsigned)k + par) & 1u;                   // 0: this entry is the low half of its word
	const bool act = k < nD;
	const bool pair = kk + 1 < LP && k + 1 < nD;                   // the last lane of a sub-group has its partner in the next chunk: it adds alone
	const unsigned val = e == 0u ? ((unsigned)L | (pair ? (unsigned)Lnext << 16 : 0u)) : ((unsigned)L << 16);
	if (act && (e == 0u || kk == 0)) atomicAdd(wordsBase + (((unsigned)k + par) >> 1), val);
}

// The grid holds, per direction, ceil(lines / PW) workgroups (first[] counts workgroups); sub-group s of workgroup g owns line g * PW + s.
// MD = capacity of a line buffer in disparities (>= maxNumDisp).
template <int LP, int MD, bool DELTA = false>   // DELTA: L - C into this direction's byte volume instead of atomic u16 sums (see sgm_step)
__global__ __launch_bounds__(64) void sgm_path_sub_kernel(const float* __restrict__ grayL, int w, int vw, int vh,
		const SGMPixel* __restrict__ pixels, const unsigned char* __restrict__ costs, unsigned* __restrict__ accumWords,
		const unsigned short* __restrict__ P2s, int P1, SGMDirs dirs, unsigned char* __restrict__ deltas = nullptr, unsigned long long numCosts = 0) {
	constexpr int PW = 64 / LP;
	__shared__ int s_L[PW][2][MD];                                     // previous / current L of each line, entries [0, nD)
	__shared__ unsigned short s_P2[256];
	__shared__ SGMPixel s_px[PW][2][LP];                               // pixel-table chunks: LP pixels of each line, double-buffered
	__shared__ float s_g[PW][2][LP];
	__shared__ unsigned char s_c[PW][2][LP][LP];                       // cost bytes of the first LP disparities of the chunk's pixels, prefetched a chunk ahead
	const int lane = threadIdx.x, sub = lane / LP, kk = lane % LP;
	int dir = 0;
#pragma unroll
	for (int i = 1; i < 8; ++i) dir += (int)blockIdx.x >= dirs.first[i] ? 1 : 0;
	const int line = ((int)blockIdx.x - dirs.first[dir]) * PW + sub;
	const int dx = dirs.dx[dir], dy = dirs.dy[dir];
	const SGMLines ln = dirs.ln[dir];
	unsigned char* dvol = DELTA ? deltas + (unsigned long long)dir * numCosts : nullptr;
	const bool haveLine = line < ln.nA + ln.nB;
	int x = -1, y = -1;                                                // a line that does not exist starts (and stays) outside
	if (haveLine) {
		if (line < ln.nA) { x = ln.ax + line * ln.adx; y = ln.ay + line * ln.ady; }
		else { const int i = line - ln.nA; x = ln.bx + i * ln.bdx; y = ln.by + i * ln.bdy; }
	}
	for (int k = lane; k < 256; k += 64) s_P2[k] = P2s[k];
	// lane kk of a sub-group owns entry kk of its line's chunk
	auto tableLoad = [&](int cx, int cy, SGMPixel& px, float& g) {
		const int tx = cx + kk * dx, ty = cy + kk * dy;
		px.idx = 0; px.minDisp = 0; px.maxDisp = 0; px.pad = 0; g = 0.f;
		if (haveLine && tx >= 0 && ty >= 0 && tx < vw && ty < vh) {
			px = pixels[(size_t)ty * vw + tx];
			g = grayL[(size_t)ty * w + tx]; // imageGray(u) with the valid-grid coordinate: the reference's quirk (:1078)
		}
	};
	// The recurrence is a dependent chain and few waves share a CU, so a global load inside a step would be paid in full: the cost bytes of a chunk
	// (entry kk of each of its LP pixels; wider pixels fetch the rest on demand) are requested one chunk ahead into registers and parked in LDS.
	auto costLoad = [&](int sl, int (&c)[LP]) {
#pragma unroll
		for (int t = 0; t < LP; ++t) {
			const SGMPixel p = s_px[sub][sl][t];
			c[t] = kk < p.maxDisp - p.minDisp ? (int)costs[p.idx + (unsigned)kk] : 0;
		}
	};
	auto costStore = [&](int sl, const int (&c)[LP]) {
#pragma unroll
		for (int t = 0; t < LP; ++t) s_c[sub][sl][t][kk] = (unsigned char)c[t];
	};
	SGMPixel tpx; float tg;
	tableLoad(x, y, tpx, tg);
	s_px[sub][0][kk] = tpx; s_g[sub][0][kk] = tg;
	tableLoad(x + LP * dx, y + LP * dy, tpx, tg);                      // chunk 1, in flight
	__syncthreads();
	int cr[LP];
	costLoad(0, cr); costStore(0, cr);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	int rpMin = 0, rpMax = 0, cur = 0; float Ip = 0.5f;               // state of the line (uniform inside the sub-group)
	int slot = 0;
	// the lines of a wave end at different pixels: loop while any of them is still inside
	while (__any(haveLine && x >= 0 && y >= 0 && x < vw && y < vh)) {
		WAVE_LOCKSTEP_POINT();                                        // every lane is done reading slot^1 (the chunk before this one)
		s_px[sub][slot ^ 1][kk] = tpx; s_g[sub][slot ^ 1][kk] = tg;
		tableLoad(x + 2 * LP * dx, y + 2 * LP * dy, tpx, tg);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		costLoad(slot ^ 1, cr);                                       // the next chunk's cost bytes: in flight during the LP steps below
#pragma unroll 1
		for (int t = 0; t < LP; ++t) {
			const SGMPixel px = s_px[sub][slot][t];
			const int rsMin = px.minDisp, rsMax = px.maxDisp, nD = rsMax - rsMin;
			const bool on = nD > 0;                                   // invalid pixels do not reset Lp / Ip (:1071-1072)
			if (!__any(on)) continue;
			const float g = s_g[sub][slot][t];
			const float DI = g - Ip;
			int ip = sgm_round2int(255.f * DI); ip = ip < 0 ? -ip : ip;
			const int P2 = s_P2[on ? ip : 0];
			const int lo = max(rpMin, rsMin), hi = min(rpMax, rsMax);
			const bool fresh = lo >= hi;                              // no common disparity (also the first pixel of a line)
			const int off = rsMin - rpMin, nDp = rpMax - rpMin;       // Lp(d) of entry k sits at k + off if that is inside [0, nDp)
			const int* Lp = &s_L[sub][cur][0];
			int* Ls = &s_L[sub][cur ^ 1][0];
			unsigned* wordsBase = accumWords + (px.idx >> 1);
			const unsigned par = (unsigned)(px.idx & 1ull);
			// pass 1: m = min over the entries of this pixel of Lp at the same disparity (SGM_INF outside the previous range)
			int m = SGM_INF;
			for (int k = kk; __any(on && !fresh && k - kk < nD); k += LP) {      // (uniform loop: every lane takes part in the vote)
				const int ipx = k + off;
				const int a0 = (on && !fresh && k < nD && (unsigned)ipx < (unsigned)nDp) ? Lp[ipx] : SGM_INF;
				m = min(m, a0);
			}
			m = sgm_sub_min<LP>(m);
			// pass 2: L of every entry, chunk by chunk
			for (int k = kk; __any(on && k - kk < nD); k += LP) {
				const bool mine = on && k < nD;
				const int c = !mine ? 0 : (k < LP ? (int)s_c[sub][slot][t][kk] : (int)costs[px.idx + (unsigned)k]);
				int L, dl = P2;                                          // dl = L - c
				if (fresh) L = c + P2;
				else {
					const int ipx = k + off;
					const int a0 = (mine && (unsigned)ipx < (unsigned)nDp) ? Lp[ipx] : SGM_INF;
					const int am = (mine && k > 0 && (unsigned)(ipx - 1) < (unsigned)nDp) ? Lp[ipx - 1] : SGM_INF;
					const int ap = (mine && k < nD - 1 && (unsigned)(ipx + 1) < (unsigned)nDp) ? Lp[ipx + 1] : SGM_INF;
					const int side = min(am, ap) + P1;
					const int best = min(min(m + P2, a0), side);
					dl = best - m; L = c + dl;
				}
				if (mine) Ls[k] = L;
				if (DELTA) { if (mine) dvol[px.idx + (unsigned)k] = (unsigned char)dl; }
				else sgm_accumulate_sub<LP>(wordsBase, par, k, kk, mine ? nD : 0, L);
			}
			if (on) { rpMin = rsMin; rpMax = rsMax; Ip = g; cur ^= 1; }
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
		costStore(slot ^ 1, cr);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		x += LP * dx; y += LP * dy; slot ^= 1;
	}
}
