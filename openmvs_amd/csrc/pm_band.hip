// pm_band.hip -- the per-pixel visit (ProcessPixel, DepthMap.cpp:630-852) with its state in LDS, and the sweep kernel of large batches built on it:
// pm_sweep2_kernel, one launch per anti-diagonal, G lanes per pixel and VPL source views per lane.  (The file name is historical: round 3's resident "band"
// kernel -- one launch per sweep iteration with wave-to-wave hand-offs -- lived here; it was bit-exact and slower than per-diagonal launches everywhere it was
// measured, profiles/README.md, and is gone.)
#pragma once

#ifndef PM_BAND_MINWAVES
#define PM_BAND_MINWAVES 3
#endif
#ifndef PM_WIDE_TRIPS
#define PM_WIDE_TRIPS 1   // pm_sweep2_kernel: wide tail trips (see the kernel); 0 compiles them out
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
};
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T* pm_launder(T* p) { asm volatile("" : "+v"(p)); return p; }
#else
template <class T> __device__ __forceinline__ T* pm_launder(T* p) { return p; }
#endif
enum { PMF_SMOOTH = 1, PMF_CHANGED = 2, PMF_POK0 = 4, PMF_POK1 = 8 };
enum { ST_PROP0 = 0, ST_PROP1 = 1, ST_DECIDE = 2, ST_RAND = 3, ST_REFINE = 4, ST_DONE = 5 };   // ProcessPixel's control flow as states (PMPix::st)

// bit g * G of the result is set for every pixel slot g of a wave: the lanes with v == 0
template <int G> __device__ __forceinline__ unsigned long long pm_lane0_mask() { unsigned long long m = 0; for (int i = 0; i < 64; i += G) m |= 1ull << i; return m; }

// The scoring part of one trip of pm_visit for the G lanes [v = 0 .. G) of one pixel: the smoothness factors of the pixel's recorded hypothesis (PMPix::hd, hn*), its score
// against the lane's VPL source views v, v + G, ..., ScorePixel's aggregation over the pixel's lanes, and the accept (DepthMap.cpp:794-799, :784-793, :843-851) by lane 0.
// on: these lanes have a pixel in this trip.  s_wg / s_pixg: the pixel's weights and state in LDS.
template <int G, int VPL, bool GEO, bool BUF>
__device__ __forceinline__ void pm_trip_score(const PMTask& t, const PMKParams& kp, const PMImgBuf& rs, const float2* s_wg, PMPix* s_pixg, const double* hotBase, int v, bool on PM_PROF_ARG) {
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	const int slot = v & 3;
	float hd, hnx, hny, hnz;
	{ const PMPix* P = pm_launder(s_pixg); hd = P->hd; hnx = P->hnx; hny = P->hny; hnz = P->hnz; }
	// -- smoothness factors of the hypothesis plane w.r.t. the close neighbours, DepthMap.cpp:524-533, one neighbour per lane
	float sf0, sf1, sf2, sf3;
	{
		const PMPix* P = pm_launder(s_pixg);
		const int flags = P->flags;
		const bool sm = on && (flags & PMF_SMOOTH) && ((flags >> (8 + slot)) & 1);
		float myF = 1.f;
		if (sm) {
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
		const PMPix* P = pm_launder(s_pixg);
#pragma unroll 1
		for (int u = 0; u < VPL; ++u) {
			const int vw = v + u * G;
			if (on && vw < t.nSrc) {
				const float s1 = pm_score_view<GEO, BUF ? 2 : 1, true>(t.src[vw], t, kp, P->x, P->y, P->X0x, P->X0y, P->normSq0, P->sumW, s_wg, hd, hnx, hny, hnz, sf0, sf1, sf2, sf3, 0.f,
					hotBase + vw * NBD, hotBase + vw * NBD + PM_SRC_HOT, rs PM_PROF_PASS);
				if (s1 < sc) { sc2 = sc; sc = s1; } else if (s1 < sc2) sc2 = s1;
			}
		}
	}
	const float nconf = pm_aggregate<G>(sc, t.nSrc, kp.thRobust, sc2);
	{	// -- accept (DepthMap.cpp:794-799, :784-793, :843-851)
		PMPix* P = pm_launder(s_pixg);
		if (on && v == 0 && P->conf > nconf) {
			P->conf = nconf; P->depth = P->hd; P->nx = P->hnx; P->ny = P->hny; P->nz = P->hnz;
			int flags = P->flags | PMF_CHANGED;
			P->flags = flags;
			const int hst = P->hst;
			if (hst == ST_RAND) { if (nconf < kp.thConfRand) P->st = ST_DECIDE; }
			else if (hst == ST_REFINE) { P->p0 = P->hp0; P->p1 = P->hp1; const int is = P->idxScale + 1; P->idxScale = is; P->scaleRange = pm_pow2neg((unsigned)is); }
		}
	}
}

// ---- a visit (ProcessPixel, DepthMap.cpp:630-852) in three parts: head, trips, write-back -----------------------------------------------------------------------------------
// Head of the visit of ONE pixel by its G lanes: what the visit reads from memory (its own estimate, the four neighbours, prior, mask, the reference patch), the patch weights
// (FillPixelPatch) and the visit's state, all of it into LDS (s_wg, s_pixg).  Must be called by every thread of the workgroup (pm_fill_patch synchronises).
// sp: the pixel (PMStep); slots: 0 (x+sgn,y), 1 (x,y+sgn) = the neighbours the sweep has already updated, 2 (x-sgn,y), 3 (x,y-sgn) = those it has not reached yet; across a
// tile border (sp.oldMask) a neighbour is read from the snapshot the sweep started from.
template <int G>
__device__ __forceinline__ void pm_visit_head(const PMTask& t, const PMKParams& kp, const PMStepPix& sp, int sgn, float2* s_wg, PMPix* s_pixg, int g, int v) {
	const int slot = v & 3;
	const bool active = sp.active;
	const int x = sp.x, y = sp.y, w = t.w, h = t.h;
	const size_t idx = (size_t)y * w + x;
	const pm_gcf gDepth = pm_glob(t.depth), gNormal = pm_glob(t.normal), gConf = pm_glob(t.conf);
	// my smoothness slot's neighbour (lane v looks after slot v & 3) and, for every lane, the two already-updated ones
	bool bok[4]; int qxs[4], qys[4]; size_t qis[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int ox = (k == 0) ? sgn : (k == 2 ? -sgn : 0), oy = (k == 1) ? sgn : (k == 3 ? -sgn : 0);
		bool ok;
		if (ox == -1) ok = x > PM_HW; else if (ox == 1) ok = x < w - PM_HW; else if (oy == -1) ok = y > PM_HW; else ok = y < h - PM_HW;
		bok[k] = ok && active; qxs[k] = x + ox; qys[k] = y + oy;
		qis[k] = bok[k] ? (size_t)(y + oy) * w + (x + ox) : idx;
	}
	const pm_gcf nD0 = pm_glob((sp.oldMask & 1u) ? t.depthOld : t.depth), nN0 = pm_glob((sp.oldMask & 1u) ? t.normalOld : t.normal), nC0 = pm_glob((sp.oldMask & 1u) ? t.confOld : t.conf);
	const pm_gcf nD1 = pm_glob((sp.oldMask & 2u) ? t.depthOld : t.depth), nN1 = pm_glob((sp.oldMask & 2u) ? t.normalOld : t.normal), nC1 = pm_glob((sp.oldMask & 2u) ? t.confOld : t.conf);
	float n0D = 0.f, n0N0 = 0.f, n0N1 = 0.f, n0N2 = 0.f, n0C = 2.f, n1D = 0.f, n1N0 = 0.f, n1N1 = 0.f, n1N2 = 0.f, n1C = 2.f;
	float oDepth = 0.f, oNx = 0.f, oNy = 0.f, oNz = 0.f, oConf = 2.f, prior = 0.f;
	float myD = 0.f, myN0 = 0.f, myN1 = 0.f, myN2 = 1.f;     // depth and normal of my smoothness slot's pixel
	unsigned char maskByte = 1;
	if (active) {
		const size_t q0 = qis[0], q1 = qis[1];
		n0D = nD0[q0]; n0N0 = nN0[q0 * 3]; n0N1 = nN0[q0 * 3 + 1]; n0N2 = nN0[q0 * 3 + 2]; n0C = nC0[q0];
		n1D = nD1[q1]; n1N0 = nN1[q1 * 3]; n1N1 = nN1[q1 * 3 + 1]; n1N2 = nN1[q1 * 3 + 2]; n1C = nC1[q1];
		if (t.prior) prior = pm_glob(t.prior)[idx];
		if (t.mask != nullptr) maskByte = t.mask[idx];
		if (slot == 0) { myD = bok[0] ? n0D : 0.f; myN0 = n0N0; myN1 = n0N1; myN2 = n0N2; }
		else if (slot == 1) { myD = bok[1] ? n1D : 0.f; myN0 = n1N0; myN1 = n1N1; myN2 = n1N2; }
		else {   // the two neighbours the sweep has not reached yet: in this tile the maps hold what the sweep found; across a tile border that is the snapshot
			const size_t qi = slot == 2 ? qis[2] : qis[3];
			const bool old = (sp.oldMask >> slot) & 1u;
			const pm_gcf sD = pm_glob(old ? t.depthOld : t.depth), sN = pm_glob(old ? t.normalOld : t.normal);
			myD = sD[qi]; myN0 = sN[qi * 3]; myN1 = sN[qi * 3 + 1]; myN2 = sN[qi * 3 + 2];
		}
		oDepth = gDepth[idx]; oNx = gNormal[idx * 3]; oNy = gNormal[idx * 3 + 1]; oNz = gNormal[idx * 3 + 2]; oConf = gConf[idx];
	}
	float normSq0, sumW;
	pm_fill_patch<G, true>(t, active, x, y, v, s_wg, normSq0, sumW);
	const bool masked = active && maskByte == 0;
	const bool valid = active && !masked && !(normSq0 < kp.thMagnitudeSq && !(prior > 0));
	if (v == 0) s_wg[PM_NT] = make_float2(prior, prior > 0 ? pm_expf(normSq0 * (-1.f / (1.f * 0.02f))) : 0.f);
	// ---- the visit's state goes to LDS: current estimate, neighbours, close-neighbour slots (lane `slot` of the first quad writes slot `slot`) ----
	PMPix* P = pm_launder(s_pixg);
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
		P->st = valid ? ST_PROP0 : ST_DONE; P->it = 0; P->idxScale = 0; P->pad0 = 0;
		P->flags = PMF_SMOOTH | ((closeMask & 1u) ? PMF_POK0 : 0) | ((closeMask & 2u) ? PMF_POK1 : 0) | (int)(closeMask << 8);
		P->nb[0][0] = n0D; P->nb[0][1] = n0N0; P->nb[0][2] = n0N1; P->nb[0][3] = n0N2; P->nb[0][4] = n0C;
		P->nb[1][0] = n1D; P->nb[1][1] = n1N0; P->nb[1][2] = n1N1; P->nb[1][3] = n1N2; P->nb[1][4] = n1C;
	}
}

// The next hypothesis of the pixel whose state is *s_pixg, by the G lanes of a group (every lane computes the same; lane 0 records it in the state): ProcessPixel's control
// flow as a state machine -- <= 2 propagation candidates (DepthMap.cpp:775-799), then <= nRandomIters refinements (:832-852) or random restarts (:806-826).  Returns whether
// there is one (false: the visit of this pixel is over).
template <int G>
__device__ __forceinline__ bool pm_next_hypothesis(const PMTask& t, const PMKParams& kp, uint32_t k1, int sgn, PMPix* s_pixg, int v, bool on) {
	bool need = false;
	float hd = 0.f, hnx = 0.f, hny = 0.f, hnz = 1.f;
	PMPix* P = pm_launder(s_pixg);
	int st = on ? P->st : (int)ST_DONE; unsigned it = (unsigned)P->it, idxScale = (unsigned)P->idxScale; int flags = P->flags;
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
	if (on && v == 0) {
		P->st = st; P->it = (int)it; P->idxScale = (int)idxScale; P->flags = flags;
		P->scaleRange = scaleRange; P->depthRange = depthRange; P->p0 = p0; P->p1 = p1;
		P->hd = hd; P->hnx = hnx; P->hny = hny; P->hnz = hnz; P->hp0 = hp0; P->hp1 = hp1; P->hst = hst;
#ifdef PM_PROFILE
		if (need) P->pad0 += 1;
#endif
	}
	return need;
}

// One launch of a sweep (PMStep): in place; launches on one stream order the anti-diagonals.  G lanes per pixel, VPL source views per lane, 64 / G lane groups.
// NP pixels are RESIDENT in a wave (their visits' state and patch weights in LDS): NP == 64 / G is one pixel per lane group for the whole visit.  With NP a multiple of that,
// every trip picks the pixels that score a hypothesis in it among all that are not done -- those on the long path first (a random restart that succeeds earns a second round
// of refinements: 14 trips against the usual 8), the halves of the range taking turns otherwise.  The lock-step of a wave's pixels costs a third of its trips (a wave of 16
// pixels runs 12.8 trips for 7.9 hypotheses per pixel: the tail belongs to one or two pixels); with twice the pixels resident the lane groups stay busy until the range runs
// out, and a wave-visit of 32 pixels takes ~17 trips instead of 2 x 12.8.  It pays where the launch has more waves than the GPU holds (the time is then the sum of the
// wave-visits, VALU issue: profiles/r06_call1); a launch that leaves SIMDs idle is better off with short waves.  Which pixel a lane group scores when is scheduling only:
// the pixels of a launch do not read each other, and a pixel's hypotheses, scores and accepts are its own -- the same bits.
// (Measured and dropped in round 4: "view-major" lanes -- lane = view * pixels-per-wave + pixel, so that the four lanes of a quad read adjacent entries of one quad
// image -- 43.1 vs 42.6 Mpix/s at 100 views, 28.1 vs 28.3 at 25: the order in which a wave's addresses reach the vector L1 is not what bounds the kernel.)
template <int G, int VPL, bool GEO, bool BUF, int NP>
__global__ __launch_bounds__(64, (GEO ? PM_BAND_MINWAVES : PM_BAND_MINWAVES_PHOTO)) void pm_sweep2_kernel(const PMTask* __restrict__ tasks, PMKParams kp, PMStep st, uint32_t pass) {
	constexpr int PPW = 64 / G;
	constexpr int ROUNDS = NP / PPW;
	constexpr int NV = G * VPL;
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	static_assert(G >= 4, "a pixel gets at least a quad of lanes");
	static_assert(NP % PPW == 0 && (ROUNDS == 1 || ROUNDS == 2) && NP <= 64, "resident pixels: one or two per lane group");
	PM_PROF_DECL;
	__shared__ float2 s_w[NP][PM_NT + 1];
	__shared__ double s_src[NV * NBD];
	__shared__ PMPix s_pix[NP];
	__shared__ int s_order[ROUNDS > 1 ? PPW : 1];
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
	const int g = lane / G, v = lane % G;
	const int w = t.w, h = t.h;
	const int sgn = st.dir == 0 ? -1 : 1;
	const int tile = (int)vbx / st.cpt, first = ((int)vbx % st.cpt) * NP;     // this wave's pixels: [first, first + NP) of the tile's anti-diagonal
	float2* const s_wBase = &s_w[0][0]; PMPix* const s_pixBase = &s_pix[0];
	// ---- heads ----
#pragma unroll 1
	for (int r = 0; r < ROUNDS; ++r) {
		const int q = r * PPW + g;
		const PMStepPix sp = pm_step_pixel(st, w, h, tile, first + q);
		pm_visit_head<G>(t, kp, sp, sgn, s_wBase + (size_t)q * (PM_NT + 1), s_pixBase + q, g, v);
	}
	__syncthreads();
	PM_TICK(12); PM_COUNT(9, 1);
	// ---- trips: every trip scores at most one hypothesis per lane group ----
	const uint32_t k1 = t.k1base + pass;
#pragma unroll 1
	for (int trip = 0; ; ++trip) {
		int pid = g; bool on = true;
		if constexpr (ROUNDS > 1) {
			// which pixels take part: every resident state is looked after by one lane (lane l < NP, state l)
			int key = -1;                                        // -1: done.  0 / 1: on the long path, 2 / 3: not; even: in the half of the range whose turn it is
			if (lane < NP) {
				const PMPix* Q = pm_launder(s_pixBase + lane);
				if (Q->st != ST_DONE) key = ((Q->flags & PMF_SMOOTH) ? 2 : 0) + ((((lane / PPW) ^ trip) & 1) ? 1 : 0);
			}
			const unsigned long long m0 = __ballot(key == 0), m1 = __ballot(key == 1), m2 = __ballot(key == 2), m3 = __ballot(key == 3);
			const int total = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
			if (total == 0) break;
			const unsigned long long below = (1ull << lane) - 1ull;
			const unsigned long long mine = key == 0 ? m0 : key == 1 ? m1 : key == 2 ? m2 : m3;
			const int rank = (key > 0 ? __popcll(m0) : 0) + (key > 1 ? __popcll(m1) : 0) + (key > 2 ? __popcll(m2) : 0) + __popcll(mine & below);
			if (key >= 0 && rank < PPW) s_order[rank] = lane;
			__syncthreads();
			on = g < total;
			pid = on ? pm_launder(&s_order[0])[g] : 0;
			__syncthreads();                                     // (the list is rewritten in the next trip)
		}
		PMPix* const s_pixg = s_pixBase + pid; float2* const s_wg = s_wBase + (size_t)pid * (PM_NT + 1);
		const bool need = pm_next_hypothesis<G>(t, kp, k1, sgn, s_pixg, v, on);
		const unsigned long long needBal = __ballot(need);
		if (needBal == 0ull) { if constexpr (ROUNDS > 1) continue; else break; }
		PM_TICK(1); PM_COUNT(8, __popcll(needBal)); PM_HIST(__popcll(needBal) / G);
		// -- the rest of the trip: smoothness factors, scores against the source views, accept.  Everything it needs of a pixel is in LDS (state, hypothesis, weights), so ANY
		// lanes can do it for ANY pixel of the wave.  When at most half of the lane groups have a hypothesis to score (the tail of a wave's visits), each of those pixels
		// gets TWICE the lanes -- the k-th of them lanes [k 2G, (k+1) 2G), half the source views per lane -- and the trip takes about two thirds of the time.  Same
		// hypotheses, same view scores, the same two smallest of them (an exact selection, whatever the pairing), same accept: the same bits.
		bool wideTrip = false;
		if constexpr (VPL >= 2 && PM_WIDE_TRIPS) {
			wideTrip = __popcll(needBal) <= (PPW / 2) * G;            // (wave-uniform)
			if (wideTrip) {
				constexpr int G2 = G * 2, VPL2 = VPL / 2;
				const int grp = lane / G2, v2 = lane % G2;
				unsigned long long m = needBal & pm_lane0_mask<G>();     // one bit per lane group that takes part (its lane v == 0)
				const bool on2 = grp < __popcll(m);
				for (int i = 0; i < grp; ++i) m &= m - 1ull;
				const int srcLane = on2 ? (int)(__ffsll((long long)m) - 1) : 0;
				const int pid2 = __shfl(pid, srcLane, 64);                // that lane group's pixel
				pm_trip_score<G2, VPL2, GEO, BUF>(t, kp, rs, s_wBase + (size_t)pid2 * (PM_NT + 1), s_pixBase + pid2, s_src, v2, on2 PM_PROF_PASS);
			}
		}
		if (!wideTrip)
			pm_trip_score<G, VPL, GEO, BUF>(t, kp, rs, s_wg, s_pixg, s_src, v, need PM_PROF_PASS);
		__builtin_amdgcn_wave_barrier();
		PM_TICK(6);
	}
	// ---- write-back: what the visits changed ----
	const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
#pragma unroll 1
	for (int r = 0; r < ROUNDS; ++r) {
		const PMPix* P = pm_launder(s_pixBase + (r * PPW + g));
		PM_HIST2(v == 0 && P->pad0 > 0, P->pad0);
		if (v == 0 && (P->flags & PMF_CHANGED)) {
			const size_t idx = (size_t)P->y * w + P->x;
			gDepth[idx] = P->depth; gNormal[idx * 3] = P->nx; gNormal[idx * 3 + 1] = P->ny; gNormal[idx * 3 + 2] = P->nz; gConf[idx] = P->conf;
		}
	}
	PM_PROF_FLUSH();
}
