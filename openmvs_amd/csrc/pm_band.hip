// pm_band.hip -- the per-pixel visit (ProcessPixel, DepthMap.cpp:630-852) with its state in LDS, and the sweep kernel of large batches built on it:
// pm_sweep2_kernel, one launch per anti-diagonal, G lanes per pixel and VPL source views per lane.  (The file name is historical: round 3's resident "band"
// kernel -- one launch per sweep iteration with wave-to-wave hand-offs -- lived here; it was bit-exact and slower than per-diagonal launches everywhere it was
// measured, profiles/README.md, and is gone.)
#pragma once

#ifndef PM_NDRAW
#define PM_NDRAW 6     // refinement iterations whose draws a visit prepares up front (nRandomIters = 6 by default; later iterations draw on the spot)
#endif
#ifndef PM_BAND_MINWAVES
#define PM_BAND_MINWAVES 3
#endif
#ifndef PM_BAND_MINWAVES_PHOTO
#define PM_BAND_MINWAVES_PHOTO 4   // the photometric instantiations fit four waves per SIMD (128 VGPRs) without scratch
#endif
#ifndef PM_BAND_MINWAVES_PHOTO
#define PM_BAND_MINWAVES_PHOTO 4   // the photometric instantiations fit four waves per SIMD (128 VGPRs) without scratch
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
	float x1[2];                               // InterpolatePixel's ray coordinate of the already-updated neighbour: [0] same row (x + sgn), [1] same column (y + sgn)
	double hr[3];                              // n / ((n . X0) depth) of the hypothesis being scored: the view-independent part of ComputeHomographyMatrix
	float dr[PM_NDRAW][3];                     // the refinement stage's draws of iteration `it`, as 2 u - 1: they depend on (x, y, it, keys) only, so the G lanes of the
	                                           // pixel each run Philox for ANOTHER iteration at the head of the visit instead of all of them for the same one, six times
};
#if defined(__HIP_DEVICE_COMPILE__)
// The state lives in LDS and the pointer says so (address space 3): laundered as a GENERIC pointer -- what it was until round 6 -- every access became a flat_load / flat_store,
// i.e. a vector-memory instruction that takes its turn in the texture-address unit next to the tap rows' gathers, which is the unit that bounds the kernel:
// ~190 of a visit's 625 vector-memory reads were reads of this struct.
#define PM_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ PM_LDS T* pm_launder(T* p) { PM_LDS T* q = (PM_LDS T*)p; asm volatile("" : "+v"(q)); return q; }
#else
#define PM_LDS
template <class T> __device__ __forceinline__ T* pm_launder(T* p) { return p; }
#endif
enum { PMF_SMOOTH = 1, PMF_CHANGED = 2, PMF_POK0 = 4, PMF_POK1 = 8 };

// One ProcessPixel visit (DepthMap.cpp:630-852) of the G lanes of a pixel, shared by the band kernel and the per-diagonal kernel below.
// n0* / n1*: the two neighbours the sweep has already updated (depth, normal, conf), however the caller obtained them; bok / qxs / qys / qis: the four
// neighbour slots (bounds tests, coordinates, map indices).  afterPatch() runs once the visit's loads have been waited for (the band kernel publishes its
// previous step there).  Result: r* = what the maps hold at this pixel after the visit, wr = it changed.
template <int G, int VPL, bool GEO, bool BUF, bool TILED>
__device__ __forceinline__ void pm_visit(const PMTask& t, const PMKParams& kp, const PMImgBuf& rs, uint32_t pass, int sgn, float2* s_wg, PMPix* s_pixg, const double* hotBase,
		int g, int v, int slot, bool active, int x, int y, int ySafe, size_t idx, const bool* bok, const int* qxs, const int* qys, const size_t* qis, unsigned oldMask,
		float n0D, float n0N0, float n0N1, float n0N2, float n0C, float n1D, float n1N0, float n1N1, float n1N2, float n1C,
		float& rD, float& rN0, float& rN1, float& rN2, float& rC, bool& wr PM_PROF_ARG) {
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
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
		else {   // the two neighbours the sweep has not reached yet: the maps hold what the sweep found -- across a tile border (tiled sweeps) that is the snapshot
			const size_t qi = slot == 2 ? qis[2] : qis[3];
			const bool old = TILED && ((oldMask >> slot) & 1u);
			const pm_gcf sD = pm_glob(old ? t.depthOld : t.depth), sN = pm_glob(old ? t.normalOld : t.normal);
			myD = sD[qi]; myN0 = sN[qi * 3]; myN1 = sN[qi * 3 + 1]; myN2 = sN[qi * 3 + 2];
		}
		oDepth = gDepth[idx]; oNx = gNormal[idx * 3]; oNy = gNormal[idx * 3 + 1]; oNz = gNormal[idx * 3 + 2]; oConf = gConf[idx];
	}
	float normSq0, sumW;
	pm_fill_patch<G, true>(t, active, active ? x : PM_HW, active ? y : yTop, v, s_wg, normSq0, sumW);
	const bool masked = active && maskByte == 0;
	const bool valid = active && !masked && !(normSq0 < kp.thMagnitudeSq && !(prior > 0));
	if (v == 0) s_wg[PM_NT] = make_float2(prior, prior > 0 ? pm_expf(normSq0 * (-1.f / (1.f * 0.02f))) : 0.f);
	// ---- the visit's state goes to LDS: current estimate, neighbours, close-neighbour slots (lane `slot` of the first quad writes slot `slot`) ----
	{
		PM_LDS PMPix* P = pm_launder(s_pixg);
		const bool bk = slot == 0 ? bok[0] : slot == 1 ? bok[1] : slot == 2 ? bok[2] : bok[3];
		const bool okS = valid && bk && myD > 0;
		const unsigned long long bal = __ballot(okS);   // (ballot of the whole wave; my pixel's four bits -- its lanes v = 0..3 -- are picked here)
		const unsigned closeMask = (unsigned)((bal >> (g * G)) & 0xFull);
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
		if (v < 2)   // InterpolatePixel's x1 (DepthMap.cpp:915-959): lane 0 the same-row neighbour, lane 1 the same-column one
			P->x1[v] = v == 0 ? (float)(((double)(x + sgn) - t.cx) / t.fx) : (float)(((double)(y + sgn) - t.cy) / t.fy);
		// the refinement draws of iterations v, v + G, ...: one Philox per lane and round instead of one per hypothesis by every lane
		const unsigned nd = min((unsigned)PM_NDRAW, kp.nRandomIters);
		for (unsigned it0 = 0; it0 < nd; it0 += G) {
			const unsigned itv = it0 + (unsigned)v;
			const PmPhilox4 r = pm_philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)(PM_STREAM_REFINE * 256) + itv, 0u, t.k0, t.k1base + pass);
			if (itv < (unsigned)PM_NDRAW) {
				P->dr[itv][0] = 2.f * pm_u32_to_unit(r.v[0]) - 1.f; P->dr[itv][1] = 2.f * pm_u32_to_unit(r.v[1]) - 1.f; P->dr[itv][2] = 2.f * pm_u32_to_unit(r.v[2]) - 1.f;
			}
		}
	}
	__syncthreads();
	PM_TICK(12); PM_COUNT(9, 1);
	// ---- ProcessPixel's control flow as a per-pixel state machine: every outer trip scores at most one hypothesis per pixel (as pm_sweep_kernel) ----
	enum { ST_PROP0 = 0, ST_PROP1 = 1, ST_DECIDE = 2, ST_RAND = 3, ST_REFINE = 4, ST_DONE = 5 };
	const uint32_t k1 = t.k1base + pass;
	for (;;) {
		bool need = false;
		float hd = 0.f, hnx = 0.f, hny = 0.f, hnz = 1.f;
		{	// -- next hypothesis of my pixel (every lane of the group computes the same; lane 0 records it)
			PM_LDS PMPix* P = pm_launder(s_pixg);
			int st = P->st; unsigned it = (unsigned)P->it, idxScale = (unsigned)P->idxScale; int flags = P->flags;
			const int px = P->x, py = P->y;
			const float vx = P->vx, vy = P->vy, vz = 1.f;
			float scaleRange = P->scaleRange, depthRange = P->depthRange, p0 = P->p0, p1 = P->p1;
			float hp0 = 0.f, hp1 = 0.f; int hst = ST_DONE;
			while (!need && st != ST_DONE) {
				if (st <= ST_PROP1) {
					const bool vert = (st == ST_PROP1); ++st; // slot 0: same row, slot 1: same column
					const bool pok = (flags & (vert ? PMF_POK1 : PMF_POK0)) != 0;
					const PM_LDS float* nbp = P->nb[vert ? 1 : 0];
					const float cd = nbp[0], cnx = nbp[1], cny = nbp[2], cnz = nbp[3], pconf = nbp[4];
					hd = cd; hnx = cnx; hny = cny; hnz = cnz;
					if (pok && pconf < kp.thKeep) {
						// InterpolatePixel, DepthMap.cpp:915-959
						float depthNew = cd; bool zero;
						// (nx1 = (float)(((double)py - cy) / fy) is the pixel's own ray coordinate vy, x1 the neighbour's: both formed at the head of the visit)
						{
							const float nx1 = vert ? vy : vx, cn = vert ? cny : cnx;
							const float denom = cnz + nx1 * cn;
							zero = pm_fabsf(denom) < 0.0001f;
							const float x1 = P->x1[vert ? 1 : 0];
							const float nom = cd * (cnz + x1 * cn);
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
					float e0, e1, e2;
					if (it < (unsigned)PM_NDRAW) { e0 = P->dr[it][0]; e1 = P->dr[it][1]; e2 = P->dr[it][2]; }
					else {
						const PmPhilox4 r = pm_philox4x32_10((uint32_t)px, (uint32_t)py, (uint32_t)(PM_STREAM_REFINE * 256) + it, 0u, t.k0, k1);
						e0 = 2.f * pm_u32_to_unit(r.v[0]) - 1.f; e1 = 2.f * pm_u32_to_unit(r.v[1]) - 1.f; e2 = 2.f * pm_u32_to_unit(r.v[2]) - 1.f;
					}
					++it;
					const float ndepth = P->depth + (depthRange * scaleRange) * e0;
					if (!pm_in_range(ndepth, t.dMin, t.dMax)) continue;
					hp0 = p0 + (kp.angle1Range * scaleRange) * e1;
					hp1 = p1 + (kp.angle2Range * scaleRange) * e2;
					pm_dir2normal_quad(hp0, hp1, v, hnx, hny, hnz);
					if (hnx * vx + hny * vy + hnz * vz >= 0) continue;
					hd = ndepth;
					need = true; hst = ST_REFINE;
				}
			}
			double hr0 = 0.0, hr1 = 0.0, hr2 = 0.0;
			if (need) pm_homography_plane(P->X0x, P->X0y, hd, hnx, hny, hnz, hr0, hr1, hr2);
			__builtin_amdgcn_wave_barrier();                     // every lane of the group has read the state before lane 0 advances it
			if (v == 0) {
				P->hr[0] = hr0; P->hr[1] = hr1; P->hr[2] = hr2;
				P->st = st; P->it = (int)it; P->idxScale = (int)idxScale; P->flags = flags;
				P->scaleRange = scaleRange; P->depthRange = depthRange; P->p0 = p0; P->p1 = p1;
				P->hd = hd; P->hnx = hnx; P->hny = hny; P->hnz = hnz; P->hp0 = hp0; P->hp1 = hp1; P->hst = hst;
			}
		}
		const unsigned long long needBal = __ballot(need);
		if (needBal == 0ull) break;
		PM_TICK(1); PM_COUNT(8, __popcll(needBal)); PM_HIST(__popcll(needBal) / G);
		// -- smoothness factors of the hypothesis plane w.r.t. the close neighbours, DepthMap.cpp:524-533, one neighbour per lane
		float sf0, sf1, sf2, sf3;
		{
			const PM_LDS PMPix* P = pm_launder(s_pixg);
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
			sf0 = pm_quad_bcast<0>(myF); sf1 = pm_quad_bcast<1>(myF); sf2 = pm_quad_bcast<2>(myF); sf3 = pm_quad_bcast<3>(myF);
		}
		PM_TICK(2);
		// -- score against my source view(s)
		float sc = PM_INF, sc2 = PM_INF;   // the lane's two smallest view scores
		{
			const PM_LDS PMPix* P = pm_launder(s_pixg);
#pragma unroll 1
			for (int u = 0; u < VPL; ++u) {
				const int vw = v + u * G;
				if (need && vw < t.nSrc) {
					const float s1 = pm_score_view<GEO, BUF ? 2 : 1, true, true>(t.src[vw], t, kp, P->x, P->y, P->X0x, P->X0y, P->normSq0, P->sumW, s_wg, hd, hnx, hny, hnz, sf0, sf1, sf2, sf3, 0.f,
						hotBase + vw * NBD, hotBase + vw * NBD + PM_SRC_HOT, rs PM_PROF_PASS, (const double*)P->hr);
					if (s1 < sc) { sc2 = sc; sc = s1; } else if (s1 < sc2) sc2 = s1;
				}
			}
		}
		const float nconf = pm_aggregate<G>(sc, t.nSrc, kp.thRobust, sc2);
		{	// -- accept (DepthMap.cpp:794-799, :784-793, :843-851)
			PM_LDS PMPix* P = pm_launder(s_pixg);
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
		PM_TICK(6);
	}
	{
		const PM_LDS PMPix* P = pm_launder(s_pixg);
		wr = (P->flags & PMF_CHANGED) && valid;
		rD = wr ? P->depth : oDepth; rN0 = wr ? P->nx : oNx; rN1 = wr ? P->ny : oNy; rN2 = wr ? P->nz : oNz; rC = wr ? P->conf : oConf;
	}
}

// One launch of a sweep (PMStep): in place; the two already-updated neighbours are read back from the maps (the previous launch wrote them); launches on one stream order the
// anti-diagonals.  TILED: the opt-in tiled sweeps -- pixels numbered tile by tile, neighbours across a tile border read from the snapshot of the sweep's start.
// (Measured and dropped in round 4: "view-major" lanes -- lane = view * pixels-per-wave + pixel, so that the four lanes of a quad read adjacent entries of one quad
// image -- 43.1 vs 42.6 Mpix/s at 100 views, 28.1 vs 28.3 at 25: the order in which a wave's addresses reach the vector L1 is not what bounds the kernel.
// Round 6: 32 pixels resident per wave with a per-trip choice of the 16 that score, and wide tail trips -- pixels still refining get twice the lanes when at most half of the
// wave's pixels take part: bit-identical, -14 % and +-0 on the benchmark, whose every pixel scores exactly 8 hypotheses: there is no lock-step loss to recover there;
// profiles/r06_resident32_wide_trips_experiment.diff, profiles/r06_call3/ab_100.log.)
template <int G, int VPL, bool GEO, bool BUF, bool TILED>
__global__ __launch_bounds__(64, (GEO ? PM_BAND_MINWAVES : PM_BAND_MINWAVES_PHOTO)) void pm_sweep2_kernel(const PMTask* __restrict__ tasks, PMKParams kp, PMStep st, uint32_t pass) {
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
	const PMImgBuf rs = pm_make_imgbuf(t);
	const int lane = threadIdx.x;
	for (int i = lane; i < NV * NBD; i += 64) s_src[i] = ((const double*)&t.src[i / NBD])[i % NBD];
	const int g = lane / G, v = lane % G, slot = v & 3;
	const int w = t.w, h = t.h;
	const PMStepPix sp = pm_step_pixel<TILED>(st, w, h, (int)vbx * PPW + g);
	const bool active = sp.active;
	const int x = sp.x, y = sp.y;
	const size_t idx = (size_t)y * w + x;
	const int sgn = st.dir == 0 ? -1 : 1;
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
		if (TILED) {   // across a tile border: as the sweep found them
			const pm_gcf nD0 = pm_glob((sp.oldMask & 1u) ? t.depthOld : t.depth), nN0 = pm_glob((sp.oldMask & 1u) ? t.normalOld : t.normal), nC0 = pm_glob((sp.oldMask & 1u) ? t.confOld : t.conf);
			const pm_gcf nD1 = pm_glob((sp.oldMask & 2u) ? t.depthOld : t.depth), nN1 = pm_glob((sp.oldMask & 2u) ? t.normalOld : t.normal), nC1 = pm_glob((sp.oldMask & 2u) ? t.confOld : t.conf);
			n0D = nD0[q0]; n0N0 = nN0[q0 * 3]; n0N1 = nN0[q0 * 3 + 1]; n0N2 = nN0[q0 * 3 + 2]; n0C = nC0[q0];
			n1D = nD1[q1]; n1N0 = nN1[q1 * 3]; n1N1 = nN1[q1 * 3 + 1]; n1N2 = nN1[q1 * 3 + 2]; n1C = nC1[q1];
		} else {
			n0D = gDepth[q0]; n0N0 = gNormal[q0 * 3]; n0N1 = gNormal[q0 * 3 + 1]; n0N2 = gNormal[q0 * 3 + 2]; n0C = gConf[q0];
			n1D = gDepth[q1]; n1N0 = gNormal[q1 * 3]; n1N1 = gNormal[q1 * 3 + 1]; n1N2 = gNormal[q1 * 3 + 2]; n1C = gConf[q1];
		}
	}
	__syncthreads();
	float rD, rN0, rN1, rN2, rC; bool wr;
	pm_visit<G, VPL, GEO, BUF, TILED>(t, kp, rs, pass, sgn, s_w[g], &s_pix[g], s_src, g, v, slot, active, x, y, active ? y : PM_HW, active ? idx : (size_t)PM_HW * w + PM_HW, bok, qxs, qys, qis, sp.oldMask,
		n0D, n0N0, n0N1, n0N2, n0C, n1D, n1N0, n1N1, n1N2, n1C, rD, rN0, rN1, rN2, rC, wr PM_PROF_PASS);
	if (wr && v == 0) { gDepth[idx] = rD; gNormal[idx * 3] = rN0; gNormal[idx * 3 + 1] = rN1; gNormal[idx * 3 + 2] = rN2; gConf[idx] = rC; }
	PM_PROF_FLUSH();
}
