// pm_band.hip -- the sweep as ONE resident launch per iteration ("band" kernel), instead of one launch per anti-diagonal.
//
// ProcessPixel (DepthMap.cpp:630-852) at (x, y) reads the estimates its sweep direction has already updated at (x -/+ 1, y) and (x, y -/+ 1) and the
// not yet updated ones on the other side; any schedule that honours those two edges gives the sequential result.  pm_sweep_kernel
// honours them with a kernel boundary per anti-diagonal: 2 984 launches per 1080p sweep, each as long as its slowest wave, and a batch too small to
// fill the machine pays one wave-life per diagonal.  Here a wavefront OWNS a band of PPW = 64 / G image rows of one view for the whole sweep and walks
// the diagonals itself: at step d its pixel group g sits at (d - y, y), y = first row + g.  Then
//   * the horizontal neighbour already updated is the group's own previous pixel and the vertical one is the neighbouring group's previous pixel: both are
//     taken from registers (one cross-lane move), not from memory;
//   * only the band's edge row needs another wave: the neighbouring band of the same view publishes the pixel of its last row after every step
//     (agent-scope write-through stores, then a progress counter) and this band waits for "one diagonal behind" before it reads that pixel
//     (agent-scope loads).  Everything else a step reads from memory -- its own pixel, the two not yet updated neighbours, the prior, the images --
//     is data no wave writes before this step in this launch, so plain (cached) loads are right;
//   * bands take tickets from a counter, every view's band k before any band k + 1: a band's predecessor always holds a smaller ticket, hence is
//     resident or finished -- no assumption about dispatch order, no deadlock whatever the occupancy; waits are bounded and report through ctl[1].
// The hypotheses, draws, scores and comparisons of a pixel are those of pm_sweep_kernel (the per-visit body below is its body), so the maps are the same bits.
#pragma once

#ifndef PM_BAND_MINWAVES
#define PM_BAND_MINWAVES 3
#endif
#ifndef PM_BAND_SPIN_LIMIT
#define PM_BAND_SPIN_LIMIT (1 << 21)
#endif

// agent-scope relaxed accesses (global_load / global_store ... sc1: served by / written through to the memory side, never by this CU's L1)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float pm_ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pm_st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int pm_ld_agent_i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pm_st_agent_i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pm_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pm_nap() { __builtin_amdgcn_s_sleep(8); }
__device__ __forceinline__ void pm_compiler_fence() { asm volatile("" ::: "memory"); }
#else
__device__ __forceinline__ void pm_compiler_fence() {}
__device__ __forceinline__ float pm_ld_agent(const float* p) { return *p; }
__device__ __forceinline__ void pm_st_agent(float* p, float v) { *p = v; }
__device__ __forceinline__ int pm_ld_agent_i(const int* p) { return *p; }
__device__ __forceinline__ void pm_st_agent_i(int* p, int v) { *p = v; }
__device__ __forceinline__ void pm_drain_stores() {}
__device__ __forceinline__ void pm_nap() {}
#endif

// Per-pixel state of a visit, in LDS.  The G lanes of a pixel all need it and all hold the same values, so one lane writes and all read: what lives in
// registers while a hypothesis is scored (the long part of a visit) is then only what the scoring itself needs -- the register budget that decides how
// many waves a SIMD holds.  Accessed through pm_launder() so that the compiler re-reads instead of carrying values across the scoring in registers.
struct PMPix {
	float depth, nx, ny, nz, conf;             // current estimate (DepthMap.cpp:767-769)
	float p0, p1, scaleRange, depthRange;      // refinement state (:828-852)
	int st, it, idxScale, flags;               // flags: bit 0 smooth, bit 1 changed, bit 2 / 3 propagation candidate 0 / 1 exists, bits 8..11 closeMask
	float hd, hnx, hny, hnz, hp0, hp1;         // hypothesis being scored
	int hst, pad0;
	float nb[2][5];                            // the two already-updated neighbours: depth, normal, conf
	float qX[4][3], qn[4][3];                  // neighborsClose: point and normal of slot k
	float vx, vy, normSq0, sumW;
	double X0x, X0y;
	int x, y;
};
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T* pm_launder(T* p) { asm volatile("" : "+v"(p)); return p; }
#else
template <class T> __device__ __forceinline__ T* pm_launder(T* p) { return p; }
#endif
enum { PMF_SMOOTH = 1, PMF_CHANGED = 2, PMF_POK0 = 4, PMF_POK1 = 8 };

// One ProcessPixel visit (DepthMap.cpp:630-852) of the G lanes of a pixel, shared by the band kernel and the per-diagonal kernel below.
// n0* / n1*: the two neighbours the sweep has already updated (depth, normal, conf), however the caller obtained them; bok / qxs / qys / qis: the four
// neighbour slots (bounds tests, coordinates, map indices).  afterPatch() runs once the visit's loads have been waited for (the band kernel publishes its
// previous step there).  Result: r* = what the maps hold at this pixel after the visit, wr = it changed.
template <int G, int VPL, bool GEO, bool VM, class AfterPatch>
__device__ __forceinline__ void pm_visit(const PMTask& t, const PMKParams& kp, uint32_t pass, int sgn, float2* s_wg, PMPix* s_pixg, const double* hotBase,
		int g, int v, int slot, bool active, int x, int y, int ySafe, size_t idx, const bool* bok, const int* qxs, const int* qys, const size_t* qis,
		float n0D, float n0N0, float n0N1, float n0N2, float n0C, float n1D, float n1N0, float n1N1, float n1N2, float n1C,
		AfterPatch afterPatch, float& rD, float& rN0, float& rN1, float& rN2, float& rC, bool& wr PM_PROF_ARG) {
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	constexpr int TC = 0;                     // no LDS windows: the optimistic tap rows read the quad image through the vector L1
	constexpr int PPW = 64 / G;               // VM: lane = v * PPW + g (view-major), else lane = g * G + v
	const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
	const int yTop = ySafe;
	// ---- what the visit reads from memory: its own estimate, the two not yet updated neighbours, prior, mask (none of it written earlier in this launch) ----
	float oDepth = 0.f, oNx = 0.f, oNy = 0.f, oNz = 0.f, oConf = 2.f, prior = 0.f;
	float myD = 0.f, myN0 = 0.f, myN1 = 0.f, myN2 = 1.f;     // depth and normal of my smoothness slot's pixel
	unsigned char maskByte = 1;
	if (active) {
		if (t.prior) prior = pm_glob(t.prior)[idx];
		if (t.mask != nullptr) maskByte = t.mask[idx];
		if (slot == 0) { myD = bok[0] ? n0D : 0.f; myN0 = n0N0; myN1 = n0N1; myN2 = n0N2; }
		else if (slot == 1) { myD = bok[1] ? n1D : 0.f; myN0 = n1N0; myN1 = n1N1; myN2 = n1N2; }
		else { const size_t qi = slot == 2 ? qis[2] : qis[3]; myD = gDepth[qi]; myN0 = gNormal[qi * 3]; myN1 = gNormal[qi * 3 + 1]; myN2 = gNormal[qi * 3 + 2]; }
		oDepth = gDepth[idx]; oNx = gNormal[idx * 3]; oNy = gNormal[idx * 3 + 1]; oNz = gNormal[idx * 3 + 2]; oConf = gConf[idx];
	}
	float normSq0, sumW;
	pm_fill_patch<G, true>(t, active, active ? x : PM_HW, active ? y : yTop, v, s_wg, normSq0, sumW);
	afterPatch();
	const bool masked = active && maskByte == 0;
	const bool valid = active && !masked && !(normSq0 < kp.thMagnitudeSq && !(prior > 0));
	if (v == 0) s_wg[PM_NT] = make_float2(prior, prior > 0 ? pm_expf(normSq0 * (-1.f / (1.f * 0.02f))) : 0.f);
	// ---- the visit's state goes to LDS: current estimate, neighbours, close-neighbour slots (lane `slot` of the first quad writes slot `slot`) ----
	{
		PMPix* P = pm_launder(s_pixg);
		const bool bk = slot == 0 ? bok[0] : slot == 1 ? bok[1] : slot == 2 ? bok[2] : bok[3];
		const bool okS = valid && bk && myD > 0;
		const unsigned long long bal = __ballot(okS);   // (ballot of the whole wave; my pixel's four bits -- its lanes v = 0..3 -- are picked here)
		const unsigned closeMask = VM ? (unsigned)(((bal >> g) & 1ull) | (((bal >> (PPW + g)) & 1ull) << 1) | (((bal >> (2 * PPW + g)) & 1ull) << 2) | (((bal >> (3 * PPW + g)) & 1ull) << 3))
		                              : (unsigned)((bal >> (g * G)) & 0xFull);
		if (v < 4) {
			// TransformPointI2C(Point3(nx, ndepth)) in double then Cast<float>, Camera.h:338-344
			const int qx = slot == 0 ? qxs[0] : slot == 1 ? qxs[1] : slot == 2 ? qxs[2] : qxs[3];
			const int qy = slot == 0 ? qys[0] : slot == 1 ? qys[1] : slot == 2 ? qys[2] : qys[3];
			const double z = (double)myD;
			P->qX[slot][0] = okS ? (float)(((double)qx - t.cx) * z / t.fx) : 0.f;
			P->qX[slot][1] = okS ? (float)(((double)qy - t.cy) * z / t.fy) : 0.f;
			P->qX[slot][2] = okS ? (float)z : 0.f;
			P->qn[slot][0] = okS ? myN0 : 0.f; P->qn[slot][1] = okS ? myN1 : 0.f; P->qn[slot][2] = okS ? myN2 : 1.f;
		}
		if (v == 0) {
			const double X0x = ((double)x - t.cx) / t.fx, X0y = ((double)y - t.cy) / t.fy;
			P->X0x = X0x; P->X0y = X0y; P->vx = (float)X0x; P->vy = (float)X0y; P->normSq0 = normSq0; P->sumW = sumW; P->x = x; P->y = y;
			P->depth = valid ? oDepth : 0.f; P->nx = valid ? oNx : 0.f; P->ny = valid ? oNy : 0.f; P->nz = valid ? oNz : 0.f; P->conf = valid ? oConf : 2.f;
			P->p0 = 0.f; P->p1 = 0.f; P->scaleRange = 1.f; P->depthRange = 0.f;
			enum { ST_PROP0 = 0, ST_DONE = 5 };
			P->st = valid ? ST_PROP0 : ST_DONE; P->it = 0; P->idxScale = 0;
			P->flags = PMF_SMOOTH | ((closeMask & 1u) ? PMF_POK0 : 0) | ((closeMask & 2u) ? PMF_POK1 : 0) | (int)(closeMask << 8);
			P->nb[0][0] = n0D; P->nb[0][1] = n0N0; P->nb[0][2] = n0N1; P->nb[0][3] = n0N2; P->nb[0][4] = n0C;
			P->nb[1][0] = n1D; P->nb[1][1] = n1N0; P->nb[1][2] = n1N1; P->nb[1][3] = n1N2; P->nb[1][4] = n1C;
		}
	}
	__syncthreads();
	// ---- ProcessPixel's control flow as a per-pixel state machine: every outer trip scores at most one hypothesis per pixel (as pm_sweep_kernel) ----
	enum { ST_PROP0 = 0, ST_PROP1 = 1, ST_DECIDE = 2, ST_RAND = 3, ST_REFINE = 4, ST_DONE = 5 };
	const uint32_t k1 = t.k1base + pass;
	for (;;) {
		bool need = false;
		float hd = 0.f, hnx = 0.f, hny = 0.f, hnz = 1.f;
		{	// -- next hypothesis of my pixel (every lane of the group computes the same; lane 0 records it)
			PMPix* P = pm_launder(s_pixg);
			int st = P->st; unsigned it = (unsigned)P->it, idxScale = (unsigned)P->idxScale; int flags = P->flags;
			const int px = P->x, py = P->y;
			const float vx = P->vx, vy = P->vy, vz = 1.f;
			float scaleRange = P->scaleRange, depthRange = P->depthRange, p0 = P->p0, p1 = P->p1;
			float hp0 = 0.f, hp1 = 0.f; int hst = ST_DONE;
			while (!need && st != ST_DONE) {
				if (st <= ST_PROP1) {
					const bool vert = (st == ST_PROP1); ++st; // slot 0: same row, slot 1: same column
					const bool pok = (flags & (vert ? PMF_POK1 : PMF_POK0)) != 0;
					const float* nbp = P->nb[vert ? 1 : 0];
					const float cd = nbp[0], cnx = nbp[1], cny = nbp[2], cnz = nbp[3], pconf = nbp[4];
					hd = cd; hnx = cnx; hny = cny; hnz = cnz;
					if (pok && pconf < kp.thKeep) {
						// InterpolatePixel, DepthMap.cpp:915-959
						float depthNew = cd; bool zero;
						if (vert) { // same column
							const float nx1 = (float)(((double)py - t.cy) / t.fy);
							const float denom = cnz + nx1 * cny;
							zero = pm_fabsf(denom) < 0.0001f;
							const float x1 = (float)(((double)(py + sgn) - t.cy) / t.fy);
							const float nom = cd * (cnz + x1 * cny);
							if (!zero) depthNew = nom / denom;
						} else {
							const float nx1 = (float)(((double)px - t.cx) / t.fx);
							const float denom = cnz + nx1 * cnx;
							zero = pm_fabsf(denom) < 0.0001f;
							const float x1 = (float)(((double)(px + sgn) - t.cx) / t.fx);
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
					const float conf = P->conf;
					if (conf <= kp.thConfSmall) idxScale = 2;
					else if (conf <= kp.thConfBig) idxScale = 1;
					else if (conf >= kp.thConfRand) { flags &= ~PMF_SMOOTH; st = ST_RAND; it = 0; continue; }
					scaleRange = pm_pow2neg(idxScale);
					depthRange = P->depth * kp.depthRatio;
					p0 = pm_atan2f(P->ny, P->nx); p1 = pm_acosf(pm_clampf(P->nz, -1.f, 1.f)); // Normal2Dir
					st = ST_REFINE; it = 0;
				} else if (st == ST_RAND) {
					if (it >= kp.nRandomIters) { st = ST_DONE; break; }
					const PmPhilox4 r = pm_philox4x32_10((uint32_t)px, (uint32_t)py, (uint32_t)(PM_STREAM_RAND * 256) + it, 0u, t.k0, k1);
					++it;
					const float rr = t.dMinSqr + (t.dMaxSqr - t.dMinSqr) * pm_u32_to_unit(r.v[0]);
					hd = rr * rr;
					pm_random_normal(pm_u32_to_unit(r.v[1]), pm_u32_to_unit(r.v[2]), vx, vy, vz, hnx, hny, hnz);
					need = true; hst = ST_RAND;
				} else { // ST_REFINE, DepthMap.cpp:832-852
					if (it >= kp.nRandomIters) { st = ST_DONE; break; }
					const PmPhilox4 r = pm_philox4x32_10((uint32_t)px, (uint32_t)py, (uint32_t)(PM_STREAM_REFINE * 256) + it, 0u, t.k0, k1);
					++it;
					const float ndepth = P->depth + (depthRange * scaleRange) * (2.f * pm_u32_to_unit(r.v[0]) - 1.f);
					if (!pm_in_range(ndepth, t.dMin, t.dMax)) continue;
					hp0 = p0 + (kp.angle1Range * scaleRange) * (2.f * pm_u32_to_unit(r.v[1]) - 1.f);
					hp1 = p1 + (kp.angle2Range * scaleRange) * (2.f * pm_u32_to_unit(r.v[2]) - 1.f);
					pm_dir2normal(hp0, hp1, hnx, hny, hnz);
					if (hnx * vx + hny * vy + hnz * vz >= 0) continue;
					hd = ndepth;
					need = true; hst = ST_REFINE;
				}
			}
			__builtin_amdgcn_wave_barrier();                     // every lane of the group has read the state before lane 0 advances it
			if (v == 0) {
				P->st = st; P->it = (int)it; P->idxScale = (int)idxScale; P->flags = flags;
				P->scaleRange = scaleRange; P->depthRange = depthRange; P->p0 = p0; P->p1 = p1;
				P->hd = hd; P->hnx = hnx; P->hny = hny; P->hnz = hnz; P->hp0 = hp0; P->hp1 = hp1; P->hst = hst;
			}
		}
		if (!__any(need)) break;
		// -- smoothness factors of the hypothesis plane w.r.t. the close neighbours, DepthMap.cpp:524-533, one neighbour per lane
		float sf0, sf1, sf2, sf3;
		{
			const PMPix* P = pm_launder(s_pixg);
			const int flags = P->flags;
			const bool on = need && (flags & PMF_SMOOTH) && ((flags >> (8 + slot)) & 1);
			float myF = 1.f;
			if (on) {
				const float vx = P->vx, vy = P->vy;
				const float q0 = P->qX[slot][0], q1 = P->qX[slot][1], q2 = P->qX[slot][2], m0 = P->qn[slot][0], m1 = P->qn[slot][1], m2 = P->qn[slot][2];
				const float planeD = -hd * (hnx * vx + hny * vy + hnz * 1.f); // InitPlane, DepthMap.cpp:963-971
				const float dist = (hnx * q0 + (hny * q1 + hnz * q2)) + planeD; // Planef::Distance, Eigen 3-dot order
				const float r = dist / hd;
				const float factorDepth = pm_expf((r * r) * kp.smoothSigmaDepth);
				const float ca = pm_clampf((hnx * m0 + hny * m1 + hnz * m2) / pm_sqrtf((hnx * hnx + hny * hny + hnz * hnz) * (m0 * m0 + m1 * m1 + m2 * m2)), -1.f, 1.f);
				const float ac = pm_acosf(ca);
				const float factorNormal = pm_expf((ac * ac) * kp.smoothSigmaNormal);
				myF = (1.f - kp.smoothBonusDepth * factorDepth) * (1.f - kp.smoothBonusNormal * factorNormal);
			}
			if (VM) { sf0 = __shfl(myF, g, 64); sf1 = __shfl(myF, PPW + g, 64); sf2 = __shfl(myF, 2 * PPW + g, 64); sf3 = __shfl(myF, 3 * PPW + g, 64); }   // slot k lives in lane v = k of my pixel
			else { sf0 = pm_quad_bcast<0>(myF); sf1 = pm_quad_bcast<1>(myF); sf2 = pm_quad_bcast<2>(myF); sf3 = pm_quad_bcast<3>(myF); }
		}
		// -- score against my source view(s)
		float sc = PM_INF, sc2 = PM_INF;   // the lane's two smallest view scores
		{
			const PMPix* P = pm_launder(s_pixg);
#pragma unroll 1
			for (int u = 0; u < VPL; ++u) {
				const int vw = v + u * G;
				if (need && vw < t.nSrc) {
					const float s1 = pm_score_view<GEO, true, TC, true>(t.src[vw], t, kp, P->x, P->y, P->X0x, P->X0y, P->normSq0, P->sumW, s_wg, hd, hnx, hny, hnz, sf0, sf1, sf2, sf3, 0.f,
						nullptr, 0, 0, hotBase + vw * NBD, hotBase + vw * NBD + PM_SRC_HOT, nullptr PM_PROF_PASS);
					if (s1 < sc) { sc2 = sc; sc = s1; } else if (s1 < sc2) sc2 = s1;
				}
			}
		}
		const float nconf = VM ? pm_aggregate_vm<G>(sc, t.nSrc, kp.thRobust, sc2) : pm_aggregate<G>(sc, t.nSrc, kp.thRobust, sc2);
		{	// -- accept (DepthMap.cpp:794-799, :784-793, :843-851)
			PMPix* P = pm_launder(s_pixg);
			if (need && v == 0 && P->conf > nconf) {
				P->conf = nconf; P->depth = P->hd; P->nx = P->hnx; P->ny = P->hny; P->nz = P->hnz;
				int flags = P->flags | PMF_CHANGED;
				P->flags = flags;
				const int hst = P->hst;
				if (hst == ST_RAND) { if (nconf < kp.thConfRand) P->st = ST_DECIDE; }
				else if (hst == ST_REFINE) { P->p0 = P->hp0; P->p1 = P->hp1; const int is = P->idxScale + 1; P->idxScale = is; P->scaleRange = pm_pow2neg((unsigned)is); }
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	{
		const PMPix* P = pm_launder(s_pixg);
		wr = (P->flags & PMF_CHANGED) && valid;
		rD = wr ? P->depth : oDepth; rN0 = wr ? P->nx : oNx; rN1 = wr ? P->ny : oNy; rN2 = wr ? P->nz : oNz; rC = wr ? P->conf : oConf;
	}
}

// ctl[0] = ticket counter, ctl[1] = error flag (a bounded wait gave up); progress[view * nBands + band] = 1 + sequence number of the band's last finished step
template <int G, int VPL, bool GEO>
__global__ __launch_bounds__(64, PM_BAND_MINWAVES) void pm_band_kernel(const PMTask* __restrict__ tasks, PMKParams kp, int dir, uint32_t pass, int nViews, int nBands,
		int nChunks, int chunkW, const unsigned* __restrict__ order, unsigned* __restrict__ ctl, int* __restrict__ progress) {
	constexpr int PPW = 64 / G;               // pixels (= rows) per wave
	constexpr int NV = G * VPL;
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	constexpr int TC = 0;                     // no LDS windows: the optimistic tap rows read the anti-diagonal-major image through the vector L1
	static_assert(G >= 4, "the band kernel gives every pixel a quad of lanes (one smoothness slot per lane); fewer sources are padded");
	PM_PROF_DECL;
	__shared__ float2 s_w[PPW][PM_NT + 1];
	__shared__ double s_src[NV * NBD];
	__shared__ PMPix s_pix[PPW];
	const int lane = threadIdx.x;
	unsigned ticket = 0;
	if (lane == 0) ticket = atomicAdd(&ctl[0], 1u);
	ticket = (unsigned)__shfl((int)ticket, 0, 64);
	// A task = (view, band of PPW rows, chunk of chunkW columns).  order[] lists (band, chunk) pairs, in sweep order, sorted so that both predecessors of a
	// task -- the band before it (same chunk) and the chunk before it (same band) -- come earlier; every view's k-th pair gets its ticket before any (k+1)-th.
	const int entry = (int)(ticket / (unsigned)nViews), view = (int)(ticket - (unsigned)entry * (unsigned)nViews);
	if (entry >= nBands * nChunks) return;
	const unsigned oe = order[entry];
	const int band = dir == 0 ? (int)(oe >> 16) : nBands - 1 - (int)(oe >> 16);
	const int chunk = dir == 0 ? (int)(oe & 0xffffu) : nChunks - 1 - (int)(oe & 0xffffu);
	const PMTask& t = tasks[view];
	for (int i = lane; i < NV * NBD; i += 64) s_src[i] = ((const double*)&t.src[i / NBD])[i % NBD];   // views >= nSrc: zeros (the task is memset), never used
	const double* hotBase = s_src;
	const int g = lane / G, v = lane % G;
	const int slot = v & 3;                                   // my smoothness slot (every quad of a group holds all four)
	const int sgn = dir == 0 ? -1 : 1;
	// the estimate this lane's pixel group left at its previous step (the horizontal "new" neighbour of the next one)
	float pvD = 0.f, pvN0 = 0.f, pvN1 = 0.f, pvN2 = 0.f, pvC = 2.f;
	__syncthreads();
	int* const progBase = progress + ((size_t)view * nBands) * nChunks;          // [band][chunk] of this view
	{	// the chunk before this one in the sweep direction must be complete: its last column is this chunk's first horizontal neighbour
		const int hp = chunk + sgn;
		if (hp >= 0 && hp < nChunks) {
			int spins = 0;
			while (pm_ld_agent_i(progBase + (size_t)band * nChunks + hp) != 0x7fffffff) {
				pm_nap();
				if (++spins > PM_BAND_SPIN_LIMIT) { if (lane == 0) atomicOr(&ctl[1], 1u); break; }
			}
			pm_compiler_fence();
		}
	}
	int pend = 0;                                                // progress value of the previous step, published once its stores have drained (below)
	int seen = 0;                                                // last progress value read of the preceding band: no poll while it is known to be ahead
	const int nSteps = [&]() { const int yT = PM_HW + band * PPW; const int r = min(PPW, (t.h - PM_HW) - yT);
		const int xa = PM_HW + chunk * chunkW, xb = min(xa + chunkW - 1, t.w - 1 - PM_HW); return (xb + (yT + r - 1)) - (xa + yT) + 1; }();
	for (int s = 0; s < nSteps; ++s) {
		const int w = t.w, h = t.h;
		const int yTop = PM_HW + band * PPW;
		const int rows = min(PPW, (h - PM_HW) - yTop);       // rows of this band that are processable (y <= h-1-HW)
		const int y = yTop + g;
		const int xc0 = PM_HW + chunk * chunkW, xc1 = min(xc0 + chunkW - 1, w - 1 - PM_HW);   // columns of this chunk
		const int dLo = xc0 + yTop, dHi = xc1 + (yTop + rows - 1);
		const int d = dir == 0 ? dLo + s : dHi - s;
		const int q = dir == 0 ? d : (w - 1 - PM_HW) + (h - 1 - PM_HW) - d;   // sequence number of this diagonal in sweep order
		const int x = d - y;
		const bool active = g < rows && x >= xc0 && x <= xc1;
		const size_t idx = active ? (size_t)y * w + x : (size_t)yTop * w + PM_HW;
		const int predBand = band + sgn, succBand = band - sgn;
		const bool predExists = predBand >= 0 && predBand < nBands, succExists = succBand >= 0 && succBand < nBands;
		const int gCons = dir == 0 ? 0 : rows - 1;          // the group whose vertical neighbour lives in the preceding band
		const int gPub = dir == 0 ? rows - 1 : 0;           // the group the following band reads
		const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
		// ---- the two already-updated neighbours: slot0 (x+sgn, y) = my previous pixel, slot1 (x, y+sgn) = the neighbouring group's previous pixel ----
		// bounds tests exactly as written in the reference: x > HW / y > HW / x < W-HW / y < H-HW
		bool bok[4]; int qxs[4], qys[4]; size_t qis[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int ox = (k == 0) ? sgn : (k == 2 ? -sgn : 0), oy = (k == 1) ? sgn : (k == 3 ? -sgn : 0);
			bool ok;
			if (ox == -1) ok = x > PM_HW; else if (ox == 1) ok = x < w - PM_HW; else if (oy == -1) ok = y > PM_HW; else ok = y < h - PM_HW;
			bok[k] = ok && active; qxs[k] = x + ox; qys[k] = y + oy;
			qis[k] = bok[k] ? (size_t)(y + oy) * w + (x + ox) : idx;
		}
		// vertical neighbour from the adjacent group's registers (its pixel of the previous step is exactly (x, y+sgn))
		const int srcLane = min(max(lane + sgn * G, 0), 63);
		float n1D = __shfl(pvD, srcLane, 64), n1N0 = __shfl(pvN0, srcLane, 64), n1N1 = __shfl(pvN1, srcLane, 64), n1N2 = __shfl(pvN2, srcLane, 64), n1C = __shfl(pvC, srcLane, 64);
		// ... or, for the band's edge row, from the preceding band: wait until it is at most one diagonal behind, then read what it published
		{
			const int xc = d - (yTop + gCons);
			const bool needPred = predExists && xc >= xc0 && xc <= xc1;   // wave-uniform
			if (needPred) {
				const int* const predProgress = progBase + (size_t)predBand * nChunks + chunk;
				if (seen < q) {
					int spins = 0;
					while ((seen = pm_ld_agent_i(predProgress)) < q) {
						pm_nap();
						if (++spins > PM_BAND_SPIN_LIMIT) { if (lane == 0) atomicOr(&ctl[1], 1u); break; }
					}
				}
				pm_compiler_fence();                                 // the reads below stay behind the successful poll
				if (g == gCons && active) {
					const size_t qi = qis[1];
					n1D = pm_ld_agent((const float*)t.depth + qi); n1C = pm_ld_agent((const float*)t.conf + qi);
					n1N0 = pm_ld_agent((const float*)t.normal + qi * 3); n1N1 = pm_ld_agent((const float*)t.normal + qi * 3 + 1); n1N2 = pm_ld_agent((const float*)t.normal + qi * 3 + 2);
				}
			}
		}
		float n0D = pvD, n0N0 = pvN0, n0N1 = pvN1, n0N2 = pvN2, n0C = pvC;
		// The reference's bounds tests (x < W-HW, y < H-HW) let the RB2LT sweep look at the first column / row BEYOND the processable area; no sweep
		// ever writes there, so those neighbours are what the init pass left in the maps: read them from memory.
		if (bok[0] && (x + sgn < PM_HW || x + sgn > w - 1 - PM_HW)) {
			const size_t qi = qis[0];
			n0D = gDepth[qi]; n0N0 = gNormal[qi * 3]; n0N1 = gNormal[qi * 3 + 1]; n0N2 = gNormal[qi * 3 + 2]; n0C = gConf[qi];
		} else if (bok[0] && (x + sgn < xc0 || x + sgn > xc1)) {
			// first column of the chunk: the horizontal neighbour is the last column of the preceding chunk (complete, written through)
			const size_t qi = qis[0];
			n0D = pm_ld_agent((const float*)t.depth + qi); n0C = pm_ld_agent((const float*)t.conf + qi);
			n0N0 = pm_ld_agent((const float*)t.normal + qi * 3); n0N1 = pm_ld_agent((const float*)t.normal + qi * 3 + 1); n0N2 = pm_ld_agent((const float*)t.normal + qi * 3 + 2);
		}
		if (bok[1] && (y + sgn < PM_HW || y + sgn > h - 1 - PM_HW)) {
			const size_t qi = qis[1];
			n1D = gDepth[qi]; n1N0 = gNormal[qi * 3]; n1N1 = gNormal[qi * 3 + 1]; n1N2 = gNormal[qi * 3 + 2]; n1C = gConf[qi];
		}
		float rD, rN0, rN1, rN2, rC; bool wr;
		pm_visit<G, VPL, GEO, false>(t, kp, pass, sgn, s_w[g], &s_pix[g], hotBase, g, v, slot, active, x, y, yTop, idx, bok, qxs, qys, qis,
			n0D, n0N0, n0N1, n0N2, n0C, n1D, n1N0, n1N1, n1N2, n1C,
			[&]() {
				// The previous step's result is published HERE: its stores were issued before this step's loads, which the patch set-up has just waited
				// for, so draining them costs nothing now (at the end of the previous step it was a full write-through round trip on the critical path).
				if (pend) {
					pm_drain_stores();
					if (lane == 0) pm_st_agent_i(progBase + (size_t)band * nChunks + chunk, pend);
					pend = 0;
				}
			}, rD, rN0, rN1, rN2, rC, wr PM_PROF_PASS);
		// ---- result of the step: what the map holds at this pixel from now on ----
		pvD = rD; pvN0 = rN0; pvN1 = rN1; pvN2 = rN2; pvC = rC;
		if (wr && v == 0) {
			const bool lastCol = dir == 0 ? (x == xc1 && chunk + 1 < nChunks) : (x == xc0 && chunk > 0);
			if ((succExists && g == gPub) || lastCol) {
				// the following band (or the following chunk) reads this pixel while the launch runs: write it through
				pm_st_agent((float*)t.depth + idx, rD); pm_st_agent((float*)t.normal + idx * 3, rN0); pm_st_agent((float*)t.normal + idx * 3 + 1, rN1);
				pm_st_agent((float*)t.normal + idx * 3 + 2, rN2); pm_st_agent((float*)t.conf + idx, rC);
			} else { gDepth[idx] = rD; gNormal[idx * 3] = rN0; gNormal[idx * 3 + 1] = rN1; gNormal[idx * 3 + 2] = rN2; gConf[idx] = rC; }
		}
		if (succExists) pend = q + 1;
		__syncthreads();                                         // the next step rewrites s_pix / s_w
	}
	pm_drain_stores();                                           // the last column has left this wave: the next chunk and the next band may read it
	if (lane == 0) pm_st_agent_i(progBase + (size_t)band * nChunks + chunk, 0x7fffffff);
	PM_PROF_FLUSH();
}

// The same visit, one launch per anti-diagonal (the schedule of pm_sweep_kernel): all pixels of diagonal x + y == d of every view of the group, the
// two already-updated neighbours read back from the maps (the previous launch wrote them).  For batches large enough to fill the machine with one
// diagonal this beats the resident band kernel (no hand-offs, no waiting on a preceding band); see DESIGN.md 4.2c for the measured crossover.
// VM ("view-major" lanes): lane = v * PPW + g instead of g * G + v.  The pixels of a wave lie on an anti-diagonal, so in the anti-diagonal-major quad image the taps
// of ADJACENT PIXELS against the SAME source view are adjacent 16-byte entries: with view-major lanes the four lanes of a quad read one 64-byte piece of one image
// (one request to the vector L1 per quad), with pixel-major lanes they read four different images (four requests).  Same values; the per-pixel exchanges
// (smoothness factors, MINMEAN) go through __shfl instead of DPP.
template <int G, int VPL, bool GEO, bool VM>
__global__ __launch_bounds__(64, PM_BAND_MINWAVES) void pm_sweep2_kernel(const PMTask* __restrict__ tasks, PMKParams kp, int dir, int d, int xlo, int count, uint32_t pass) {
	constexpr int PPW = 64 / G;
	constexpr int NV = G * VPL;
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	static_assert(G >= 4, "a pixel gets at least a quad of lanes");
	PM_PROF_DECL;
	__shared__ float2 s_w[PPW][PM_NT + 1];
	__shared__ double s_src[NV * NBD];
	__shared__ PMPix s_pix[PPW];
	// XCD-aware block mapping as in pm_sweep_kernel: contiguous (view, chunk) ranges per XCD
	unsigned vbx = blockIdx.x, vby = blockIdx.y;
	{
		const unsigned nbx = gridDim.x, nwg = nbx * gridDim.y, orig = blockIdx.y * nbx + blockIdx.x;
		const unsigned xcd = orig % 8u, q = nwg / 8u, r = nwg % 8u;
		const unsigned wgid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + orig / 8u;
		vby = wgid / nbx; vbx = wgid - vby * nbx;
	}
	const PMTask& t = tasks[vby];
	const int lane = threadIdx.x;
	for (int i = lane; i < NV * NBD; i += 64) s_src[i] = ((const double*)&t.src[i / NBD])[i % NBD];
	const int g = VM ? lane % PPW : lane / G, v = VM ? lane / PPW : lane % G, slot = v & 3;
	const int w = t.w, h = t.h;
	const int pi = (int)vbx * PPW + g;
	const bool active = pi < count;
	const int x = xlo + (active ? pi : 0), y = d - x;
	const size_t idx = (size_t)y * w + x;
	const int sgn = dir == 0 ? -1 : 1;
	bool bok[4]; int qxs[4], qys[4]; size_t qis[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int ox = (k == 0) ? sgn : (k == 2 ? -sgn : 0), oy = (k == 1) ? sgn : (k == 3 ? -sgn : 0);
		bool ok;
		if (ox == -1) ok = x > PM_HW; else if (ox == 1) ok = x < w - PM_HW; else if (oy == -1) ok = y > PM_HW; else ok = y < h - PM_HW;
		bok[k] = ok && active; qxs[k] = x + ox; qys[k] = y + oy;
		qis[k] = bok[k] ? (size_t)(y + oy) * w + (x + ox) : idx;
	}
	const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
	float n0D = 0.f, n0N0 = 0.f, n0N1 = 0.f, n0N2 = 0.f, n0C = 2.f, n1D = 0.f, n1N0 = 0.f, n1N1 = 0.f, n1N2 = 0.f, n1C = 2.f;
	if (active) {
		const size_t q0 = qis[0], q1 = qis[1];
		n0D = gDepth[q0]; n0N0 = gNormal[q0 * 3]; n0N1 = gNormal[q0 * 3 + 1]; n0N2 = gNormal[q0 * 3 + 2]; n0C = gConf[q0];
		n1D = gDepth[q1]; n1N0 = gNormal[q1 * 3]; n1N1 = gNormal[q1 * 3 + 1]; n1N2 = gNormal[q1 * 3 + 2]; n1C = gConf[q1];
	}
	__syncthreads();
	float rD, rN0, rN1, rN2, rC; bool wr;
	pm_visit<G, VPL, GEO, VM>(t, kp, pass, sgn, s_w[g], &s_pix[g], s_src, g, v, slot, active, x, y, active ? y : PM_HW, active ? idx : (size_t)PM_HW * w + PM_HW, bok, qxs, qys, qis,
		n0D, n0N0, n0N1, n0N2, n0C, n1D, n1N0, n1N1, n1N2, n1C, []() {}, rD, rN0, rN1, rN2, rC, wr PM_PROF_PASS);
	if (wr && v == 0) { gDepth[idx] = rD; gNormal[idx * 3] = rN0; gNormal[idx * 3 + 1] = rN1; gNormal[idx * 3 + 2] = rN2; gConf[idx] = rC; }
	PM_PROF_FLUSH();
}
