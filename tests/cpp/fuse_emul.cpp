// fuse_emul.cpp -- host emulation of the device fusion (openmvs_amd/csrc/pm_fuse.h): the same per-seed functions the HIP kernels
// call, driven by a loop that plays the role of the GPU's thread scheduler with a different pseudo-random thread order in every
// phase of every round.  tests/test_fuse.py compares its output with the sequential oracle (oracle/fuse_oracle.cpp): that is the
// check that the deterministic-reservation scheme reproduces FuseDepthMaps' sequential semantics, runnable without a GPU.
// Test code: not part of the product, never loaded by openmvs_amd/.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../openmvs_amd/csrc/pm_fuse.h"

extern "C" {
struct EmuFuseView { const float* depth; const float* normal; const float* conf; const uint8_t* bgr; double K[9], R[9], C[3]; const uint32_t* neighbors; uint32_t nNeighbors; int w, h; };   // (w, h: carried by the oracle's layout; this harness runs the uniform-size case)
struct EmuFuseCloud { uint64_t nPoints, nDepths, nViews; float* points; uint32_t* viewStart; uint32_t* views; float* weights; uint16_t* projs; uint8_t* colors; float* normals; };

static uint64_t g_rounds = 0, g_seeds = 0;
void emu_fuse_stats(uint64_t* rounds, uint64_t* seeds) { *rounds = g_rounds; *seeds = g_seeds; }

static void shuffle(std::vector<uint32_t>& v, uint64_t& st, int mode) {
	if (mode == 0) return;
	if (mode == 1) { std::reverse(v.begin(), v.end()); return; }
	for (size_t i = v.size(); i > 1; --i) { st = st * 6364136223846793005ull + 1442695040888963407ull; std::swap(v[i-1], v[(st >> 33) % i]); }
}

int emu_fuse_depth_maps(const EmuFuseView* vs, int nImages, int w, int h, const uint32_t* order, int nOrder, unsigned nMinViewsFuse,
		float fDepthDiffThreshold, float normalError, int bEstimateColor, int bEstimateNormal, EmuFuseCloud* out) {
	memset(out, 0, sizeof(*out));
	const bool merge = nMinViewsFuse < 2;          // MergeDepthMaps, as the engine routes it
	const char* m = getenv("EMU_FUSE_ORDER");
	const int mode = m ? atoi(m) : 2;
	uint64_t st = 12345;
	if ((unsigned)nImages < nMinViewsFuse) nMinViewsFuse = (unsigned)nImages;
	const size_t P = (size_t)w * h;
	std::vector<PMFuseCam> cams(nImages);
	std::vector<float> depth(P * nImages, 0.f), normal, conf;
	std::vector<uint8_t> bgr;
	bool bNormalMap = true, hasConf = true, hasBgr = true;
	for (int i = 0; i < nImages; ++i) if (vs[i].depth) { if (!vs[i].normal) bNormalMap = false; if (!vs[i].conf) hasConf = false; }
	for (int i = 0; i < nImages; ++i) if (!vs[i].bgr) hasBgr = false;
	if (bNormalMap) normal.assign(P * 3 * nImages, 0.f);
	if (hasConf) conf.assign(P * nImages, 0.f);
	if (hasBgr) bgr.assign(P * 3 * nImages, 0);
	for (int i = 0; i < nImages; ++i) {
		memcpy(cams[i].K, vs[i].K, 72); memcpy(cams[i].R, vs[i].R, 72); memcpy(cams[i].C, vs[i].C, 24); pmfu_composeP(cams[i]);
		if (vs[i].depth) {
			memcpy(&depth[P * i], vs[i].depth, P * 4);
			if (bNormalMap) memcpy(&normal[P * 3 * i], vs[i].normal, P * 12);
			if (hasConf) memcpy(&conf[P * i], vs[i].conf, P * 4);
		}
		if (hasBgr) memcpy(&bgr[P * 3 * i], vs[i].bgr, P * 3);
	}
	if (bEstimateNormal && !bNormalMap) bEstimateNormal = 0;
	std::vector<uint32_t> claimed(P * nImages, PMFU_NO_ID), resv(P * nImages, PMFU_FREE);
	std::vector<uint8_t> recN(P), recColor(3 * P);
	std::vector<float> recX(3 * P), recWeight(PMFU_MAXV * P), recNormal(3 * P);
	std::vector<uint32_t> recView(PMFU_MAXV * P), recProj(PMFU_MAXV * P);
	std::vector<float> oPoints, oWeights, oNormals; std::vector<uint32_t> oStart, oViews; std::vector<uint16_t> oProjs; std::vector<uint8_t> oColors;
	uint64_t nDepths = 0;
	g_rounds = g_seeds = 0;
	for (int o = 0; o < nOrder; ++o) {
		const int A = (int)order[o];
		if (!vs[A].depth) continue;
		PMFuseCtx c; memset(&c, 0, sizeof(c));
		c.w = w; c.h = h; c.nImages = nImages; c.A = A; c.nNb = 0;
		for (uint32_t n = 0; n < vs[A].nNeighbors && c.nNb < PMFU_MAXNB; ++n) { const uint32_t b = vs[A].neighbors[n]; if ((int)b != A && vs[b].depth) c.nb[c.nNb++] = (int)b; }
		c.depth = depth.data(); c.normal = bNormalMap ? normal.data() : nullptr; c.conf = hasConf ? conf.data() : nullptr; c.bgr = hasBgr ? bgr.data() : nullptr;
		c.claimed = claimed.data(); c.resv = resv.data(); c.cams = cams.data();
		c.nMinViewsFuse = nMinViewsFuse; c.fDepthDiffThreshold = fDepthDiffThreshold; c.normalError = normalError;
		c.bEstimateColor = bEstimateColor; c.bEstimateNormal = bEstimateNormal;
		c.recN = recN.data(); c.recX = recX.data(); c.recView = recView.data(); c.recWeight = recWeight.data(); c.recProj = recProj.data();
		c.recColor = recColor.data(); c.recNormal = recNormal.data();
		std::fill(recN.begin(), recN.end(), 0);
		std::vector<uint32_t> pending, next;
		if (merge) {                                           // the merge kernel, threads in a shuffled order
			std::vector<uint32_t> all(P); for (uint32_t p = 0; p < (uint32_t)P; ++p) all[p] = p;
			shuffle(all, st, mode);
			for (uint32_t p : all) { pmfu_merge(c, p); nDepths += recN[p]; }
		} else
		for (uint32_t p = 0; p < (uint32_t)P; ++p) {          // the seed kernel
			if (depth[P * A + p] == 0) continue;
			++nDepths;
			if (claimed[P * A + p] != PMFU_NO_ID) continue;
			pending.push_back(p);
		}
		g_seeds += pending.size();
		while (!pending.empty()) {
			++g_rounds;
			shuffle(pending, st, mode);
			for (uint32_t p : pending) pmfu_reserve(c, p);      // reserve kernel
			shuffle(pending, st, mode);
			next.clear();
			for (uint32_t p : pending) {                        // commit kernel
				if (pmfu_owns(c, p)) pmfu_commit(c, p); else next.push_back(p);
			}
			if (next.size() == pending.size()) return 7;        // no progress: the scheme is broken
			pending.swap(next);
		}
		for (size_t i = 0; i < resv.size(); ++i) if (resv[i] != PMFU_FREE) return 8;   // every reservation must have been released
		for (uint32_t p = 0; p < (uint32_t)P; ++p) {           // compaction (scan + scatter kernels)
			const int nv = recN[p];
			if (!nv) continue;
			oStart.push_back((uint32_t)oViews.size());
			for (int k = 0; k < 3; ++k) oPoints.push_back(recX[k * P + p]);
			for (int v = 0; v < nv; ++v) {
				oViews.push_back(recView[v * P + p]); oWeights.push_back(recWeight[v * P + p]);
				oProjs.push_back((uint16_t)(recProj[v * P + p] & 0xFFFF)); oProjs.push_back((uint16_t)(recProj[v * P + p] >> 16));
			}
			if (bEstimateColor) for (int k = 0; k < 3; ++k) oColors.push_back(recColor[k * P + p]);
			if (bEstimateNormal) for (int k = 0; k < 3; ++k) oNormals.push_back(recNormal[k * P + p]);
		}
	}
	oStart.push_back((uint32_t)oViews.size());
	auto dup = [](const void* src, size_t bytes) { void* d = malloc(bytes + 8); memcpy(d, src, bytes); return d; };
	out->nPoints = oStart.size() - 1; out->nDepths = nDepths; out->nViews = oViews.size();
	out->points = (float*)dup(oPoints.data(), oPoints.size() * 4); out->viewStart = (uint32_t*)dup(oStart.data(), oStart.size() * 4);
	out->views = (uint32_t*)dup(oViews.data(), oViews.size() * 4); out->weights = (float*)dup(oWeights.data(), oWeights.size() * 4);
	out->projs = (uint16_t*)dup(oProjs.data(), oProjs.size() * 2);
	out->colors = bEstimateColor ? (uint8_t*)dup(oColors.data(), oColors.size()) : nullptr;
	out->normals = bEstimateNormal ? (float*)dup(oNormals.data(), oNormals.size() * 4) : nullptr;
	return 0;
}
}
