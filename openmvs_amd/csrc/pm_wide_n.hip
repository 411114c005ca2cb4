// pm_wide_n.hip -- the speculative sweep kernel of pm_kernels.hip (pm_sweep_wide_kernel: one wave per pixel, eight hypotheses of a pixel scored side by
// side, the reference's sequential accept rule replayed over them) with the width of the speculation as a template parameter: NH = 4 or 2 hypotheses per
// round, 8 * NH lanes per pixel, 8 / NH pixels per wave.
//
// Why: for a batch of 13-25 reference views (what a rank of an 8- or 4-GPU split of BASELINE's 100-view job holds) the batch time is the critical path of
// 21 563 dependent diagonal launches x the duration of one pixel visit.  One view per lane (pm_sweep2_kernel<8,1>) has the longest
// visit (67 us at 13 views) on a machine it fills to a quarter; the eight-wide kernel has the shortest (37 us) but needs 13 x 1 080 waves per diagonal, 4.6
// rounds of the machine.  Four- or two-wide speculation needs 7 020 / 3 510 waves per 13-view diagonal and should sit between the two.
//
// Same hypotheses, draws (counter-based: iteration index, not call order), scores and comparison order as the sequential code: same bits -- under the CPU
// emulator of the test-suite (tests/test_emu_kernels.py) and on the device (full schedule at 1920x1080, batches of 1 / 2 / 4 / 8 / 13 views, both widths, against
// the regular kernel; profiles/r03_small_batches_call24_26_narrower_speculation.log).  Measured: the two-wide kernel is the fastest of all mappings from 4 to at
// least 25 views (13 views: 23.6 Mpix/s against 17.9 with one view per lane and 17.5 eight-wide; 25 views: 32.5 against 28.2), the engine's default there.
//
// (NH = 8 of this template would be pm_sweep_wide_kernel without its LDS-window option; folding the two is left for the round that next touches pm_kernels.hip, whose
// text is tied to the committed counter measurement of this round.)
// Differences from pm_sweep_wide_kernel, all forced by several pixels sharing a wave: a pixel that is masked / done does not leave (its lanes idle to the end of
// the wave's loop); every cross-lane read (the other groups' scores and planes) is done by all lanes before the per-pixel replay, never inside its
// data-dependent control flow; window-less tap rows only (the quad images).
#pragma once
#include "pm_kernels.hip"

#ifndef PM_WIDEN_MINWAVES
#define PM_WIDEN_MINWAVES 3   // waves per SIMD the kernel is compiled for (168 VGPRs, 12-44 B of scratch; 4 has not been tried on the device)
#endif
template <bool GEO, int NH, bool BUF, bool TILED>
__global__ __launch_bounds__(64, PM_WIDEN_MINWAVES) void pm_sweep_widen_kernel(const PMTask* __restrict__ tasks, PMKParams kp, PMStep st, uint32_t pass) {
	static_assert(NH == 2 || NH == 4, "two propagation candidates need two groups; eight groups are pm_sweep_wide_kernel");
	constexpr int G = 8, LPP = G * NH, PPW = 64 / LPP;
	constexpr int NBD = PM_SRC_HOT + (GEO ? PM_SRC_GEO : 0);
	PM_PROF_DECL;
	__shared__ float2 s_w[PPW][PM_NT + 1];
	__shared__ double s_src[G * NBD];
	const PMTask& t = tasks[blockIdx.y];
	const PMImgBuf rs = pm_make_imgbuf(t);
	const int lane = threadIdx.x, p = lane / LPP, sub = lane % LPP, c = sub >> 3, v = lane & 7, seg = p * LPP;
	for (int i = lane; i < G * NBD; i += 64) s_src[i] = ((const double*)&t.src[i / NBD])[i % NBD];
	const double* hot = s_src + v * NBD;
	const int w = t.w, h = t.h;
	const PMStepPix sp = pm_step_pixel<TILED>(st, w, h, (int)blockIdx.x * PPW + p);
	const bool active = sp.active;
	const int x = sp.x, y = sp.y;                                 // a segment without a pixel takes the map's first one and never writes
	const size_t idx = (size_t)y * w + x;
	const pm_gf gDepth = pm_globw(t.depth), gNormal = pm_globw(t.normal), gConf = pm_globw(t.conf);
	const pm_gcf oDepthM = pm_glob(t.depthOld), oNormalM = pm_glob(t.normalOld), oConfM = pm_glob(t.confOld);   // tiled sweeps: neighbours in another tile (PMStep)
	const int sgn = st.dir == 0 ? -1 : 1;
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
	for (int k = 0; k < 4; ++k) nds[k] = (TILED && ((sp.oldMask >> k) & 1u)) ? oDepthM[qis[k]] : gDepth[qis[k]];
	const int slot = v & 3;                                        // my smoothness slot (both quads of a group hold all four)
	const size_t qv = (slot == 0) ? qis[0] : (slot == 1) ? qis[1] : (slot == 2) ? qis[2] : qis[3];
	const bool oldV = TILED && ((sp.oldMask >> slot) & 1u);
	const float on0 = oldV ? oNormalM[qv * 3] : gNormal[qv * 3], on1 = oldV ? oNormalM[qv * 3 + 1] : gNormal[qv * 3 + 1], on2 = oldV ? oNormalM[qv * 3 + 2] : gNormal[qv * 3 + 2];
	const float oDepth = gDepth[idx], oNx = gNormal[idx * 3], oNy = gNormal[idx * 3 + 1], oNz = gNormal[idx * 3 + 2], oConf = gConf[idx];
	// the two propagation sources' estimates (they were updated one diagonal earlier and are not touched again before this launch ends)
	float pcf[2], pcd[2], pcn[2][3];
#pragma unroll
	for (int k = 0; k < 2; ++k) {
		const size_t q = qis[k];
		if (TILED && ((sp.oldMask >> k) & 1u)) { pcf[k] = oConfM[q]; pcd[k] = oDepthM[q]; pcn[k][0] = oNormalM[q * 3]; pcn[k][1] = oNormalM[q * 3 + 1]; pcn[k][2] = oNormalM[q * 3 + 2]; }
		else { pcf[k] = gConf[q]; pcd[k] = gDepth[q]; pcn[k][0] = gNormal[q * 3]; pcn[k][1] = gNormal[q * 3 + 1]; pcn[k][2] = gNormal[q * 3 + 2]; }
	}
	float normSq0, sumW;
	pm_fill_patch<LPP, true>(t, true, x, y, sub, s_w[p], normSq0, sumW);
	const bool masked = maskByte == 0;
	const bool valid = active && !masked && !(normSq0 < kp.thMagnitudeSq && !(prior > 0));
	if (sub == 0) s_w[p][PM_NT] = make_float2(prior, prior > 0 ? pm_expf(normSq0 * (-1.f / (1.f * 0.02f))) : 0.f);
	__syncthreads();
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
	int stage = valid ? W_PROPS : W_DONE;
	unsigned it0 = 0, idxScale = 0;
	float scaleRange = 1.f, depthRange = 0.f, p0 = 0.f, p1 = 0.f;
	bool smooth = true, changed = false;
	PM_TICK(0); PM_COUNT(9, 1);
	while (__any(stage != W_DONE)) {
		const bool on = stage != W_DONE;                           // (uniform inside a pixel's segment)
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
		if (on && stage == W_PROPS && c < 2) {
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
		} else if (on) {
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
			sc = pm_score_view<GEO, BUF ? 2 : 1, true>(t.src[v], t, kp, x, y, X0x, X0y, normSq0, sumW, s_w[p], hd, hnx, hny, hnz, sf[0], sf[1], sf[2], sf[3], 0.f,
				hot, hot + PM_SRC_HOT, rs PM_PROF_PASS);
		const float nconf = pm_aggregate<G>(sc, t.nSrc, kp.thRobust);
		// ---- what the NH groups of my pixel found: read by every lane, outside any per-pixel control flow ----
		bool gNeed[NH]; float gConf_[NH], gD[NH], gNx[NH], gNy[NH], gNz[NH], gP0[NH], gP1[NH];
#pragma unroll
		for (int j = 0; j < NH; ++j) {
			const int src = seg + j * G;
			gNeed[j] = __shfl(need ? 1 : 0, src, 64) != 0; gConf_[j] = __shfl(nconf, src, 64); gD[j] = __shfl(hd, src, 64);
			gNx[j] = __shfl(hnx, src, 64); gNy[j] = __shfl(hny, src, 64); gNz[j] = __shfl(hnz, src, 64); gP0[j] = __shfl(hp0, src, 64); gP1[j] = __shfl(hp1, src, 64);
		}
		// ---- every lane replays the sequential accept rule over its pixel's groups, in the reference's order ----
		if (on) {
			bool restart = false;                                 // the state changed in a way that invalidates the remaining candidates
			if (stage == W_PROPS) {
#pragma unroll
				for (int k = 0; k < 2; ++k) {
					if (gNeed[k] && conf > gConf_[k]) {
						conf = gConf_[k]; depth = gD[k]; nx = gNx[k]; ny = gNy[k]; nz = gNz[k];
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
				// groups first .. NH-1 hold iterations it0, it0 + 1, ... of `stage` computed from exactly the present state
#pragma unroll
				for (int j = 0; j < NH; ++j) {
					if (restart || j < first) continue;
					const unsigned itj = it0 + (unsigned)(j - first);
					if (itj >= kp.nRandomIters) { restart = true; it0 = kp.nRandomIters; continue; }   // budget used up (the stage ends below)
					if (gNeed[j] && conf > gConf_[j]) {
						conf = gConf_[j]; depth = gD[j]; nx = gNx[j]; ny = gNy[j]; nz = gNz[j];
						changed = true;
						if (stage == W_REFINE) {
							p0 = gP0[j]; p1 = gP1[j]; scaleRange = pm_pow2neg(++idxScale);
							it0 = itj + 1; restart = true;            // the later refinements perturb the new plane: next round
						} else if (conf < kp.thConfRand) {
							// goto RefineIters (DepthMap.cpp:790-793): the remaining restarts are dropped
							if (conf <= kp.thConfSmall) idxScale = 2;
							else if (conf <= kp.thConfBig) idxScale = 1;
							scaleRange = pm_pow2neg(idxScale);
							depthRange = depth * kp.depthRatio;
							p0 = pm_atan2f(ny, nx); p1 = pm_acosf(pm_clampf(nz, -1.f, 1.f));
							stage = W_REFINE; it0 = 0; restart = true;
						}
					}
				}
				if (!restart) it0 += (unsigned)(NH - first);
			}
			if (it0 >= kp.nRandomIters) stage = W_DONE;            // DepthMap.cpp:833 / :781: the iteration budget of the stage is used up
		}
		PM_TICK(6);
	}
	PM_PROF_FLUSH();
	if (valid && changed && sub == 0) { gDepth[idx] = depth; gNormal[idx * 3] = nx; gNormal[idx * 3 + 1] = ny; gNormal[idx * 3 + 2] = nz; gConf[idx] = conf; }
}
