// pm_fuse.h -- depth-map fusion (DepthMapsData::FuseDepthMaps, libs/MVS/SceneDensify.cpp:1372-1650 in /root/reference) as a
// data-parallel algorithm with the reference's sequential result.
//
// The reference visits the pixels of an image in raster order; each unclaimed depth seeds a point which claims agreeing pixels in
// the neighbour views and, if kept, zeroes the neighbour depths it occludes.  All the state one seed reads or writes outside its
// own image lives in the neighbour cells its 3D point projects to, and those cells depend on the seed's depth only (never on
// what other seeds did).  Two seeds of one image therefore interact only if they share a cell, and the seeds of one image can be
// run as "deterministic reservations": every pending seed writes its raster index into its cells with atomicMin; a seed that
// holds all of its cells has no earlier pending seed it could depend on, so it runs the reference's per-seed body right away; the
// others wait for the next round.  The outcome is the sequential one whatever the thread schedule is.  Images are still taken
// one after the other (the reference's best-connected-first order); surviving seeds are compacted in raster order afterwards, which
// reproduces the reference's point numbering.
//
// This header holds the per-seed logic as host+device inline functions: pm_fuse.hip wraps them in kernels, and
// tests/cpp/fuse_emul.cpp runs the very same functions on the host under adversarial thread orders to check the scheme against the
// sequential oracle without a GPU.  Arithmetic follows the reference operation by operation (see oracle/fuse_oracle.cpp for the
// cv:: operator semantics this relies on); both sides are compiled with -ffp-contract=off.
#pragma once
#include <stdint.h>
#include "pm_math.h"

#define PMFU_MAXNB 16            // = PM_MAX_SRC: neighbours per view
#define PMFU_MAXV (PMFU_MAXNB + 1)
#define PMFU_NO_ID 0xFFFFFFFFu
#define PMFU_FREE 0xFFFFFFFFu    // reservation cell not held

#if defined(__HIP_DEVICE_COMPILE__)
#define PMFU_ATOMIC_MIN(p, v) atomicMin((p), (v))
#else
#define PMFU_ATOMIC_MIN(p, v) do { if (*(p) > (v)) *(p) = (v); } while (0)
#endif

struct PMFuseCam { double K[9], R[9], C[3], P[12]; };

// everything a seed of image A needs; arrays are [nImages][...] slabs of the scene
struct PMFuseCtx {
	int w, h, nImages;             // w, h: the size of every image's maps -- unless iw / ih give each image its own (the reference sizes every depth map on its image)
	const int* iw; const int* ih;  // [nImages] or null
	size_t slab;                   // pixels between two images in the [nImages][...] arrays and in the record arrays; 0 = w*h
	int A;                         // current image
	int nNb; int nb[PMFU_MAXNB];   // its neighbours that have a depth map, in neighbour-list order
	float* depth;                  // working copies (zeroed by occlusion), [nImages][w*h]
	const float* normal;           // [nImages][w*h*3] or null (then every normal is (0,0,-1))
	const float* conf;             // [nImages][w*h] or null (weight uses 1)
	const uint8_t* bgr;            // [nImages][w*h*3] or null
	uint32_t* claimed;             // [nImages][w*h]: PMFU_NO_ID or the claiming seed's tag
	uint32_t* resv;                // [nImages][w*h]: reservation cells
	const PMFuseCam* cams;         // [nImages]
	unsigned nMinViewsFuse; float fDepthDiffThreshold, normalError; int bEstimateColor, bEstimateNormal;
	// per-seed records of the current image, struct-of-arrays over the pixel index
	uint8_t* recN;                 // [w*h] number of views of the kept point, 0 = no point
	float* recX;                   // [3][w*h]
	uint32_t* recView;             // [PMFU_MAXV][w*h]
	float* recWeight;              // [PMFU_MAXV][w*h]
	uint32_t* recProj;             // [PMFU_MAXV][w*h]  x | y<<16
	uint8_t* recColor;             // [3][w*h]
	float* recNormal;              // [3][w*h]
};

PM_HD int pmfu_w(const PMFuseCtx& c, int img) { return c.iw ? c.iw[img] : c.w; }
PM_HD int pmfu_h(const PMFuseCtx& c, int img) { return c.ih ? c.ih[img] : c.h; }
PM_HD size_t pmfu_slab(const PMFuseCtx& c) { return c.slab ? c.slab : (size_t)c.w * c.h; }

// Camera::ComposeP -> AssembleProjectionMatrix (libs/MVS/Camera.cpp:173-180): M = K*R, P = [M | M*(-C)], sums left to right
inline void pmfu_composeP(PMFuseCam& c) {
	double M[9];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += c.K[i*3+k] * c.R[k*3+j]; M[i*3+j] = s; }
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) c.P[i*4+j] = M[i*3+j];
		c.P[i*4+3] = M[i*3+0] * (-c.C[0]) + M[i*3+1] * (-c.C[1]) + M[i*3+2] * (-c.C[2]);
	}
}

PM_HD void pmfu_I2W(const PMFuseCam& c, double x, double y, double z, double* X) {
	const double ci0 = (x - c.K[2]) * z / c.K[0], ci1 = (y - c.K[5]) * z / c.K[4], ci2 = z;
	X[0] = ((0.0 + c.R[0] * ci0) + c.R[3] * ci1) + c.R[6] * ci2 + c.C[0];
	X[1] = ((0.0 + c.R[1] * ci0) + c.R[4] * ci1) + c.R[7] * ci2 + c.C[1];
	X[2] = ((0.0 + c.R[2] * ci0) + c.R[5] * ci1) + c.R[8] * ci2 + c.C[2];
}
PM_HD void pmfu_projectP3(const PMFuseCam& c, const float* X, float* q) {
	q[0] = (float)(c.P[0] * X[0] + c.P[1] * X[1] + c.P[2] * X[2] + c.P[3]);
	q[1] = (float)(c.P[4] * X[0] + c.P[5] * X[1] + c.P[6] * X[2] + c.P[7]);
	q[2] = (float)(c.P[8] * X[0] + c.P[9] * X[1] + c.P[10] * X[2] + c.P[11]);
}
PM_HD void pmfu_normalW(const PMFuseCam& c, const float* n, float* o) {
	o[0] = (float)(((0.0 + c.R[0] * (double)n[0]) + c.R[3] * (double)n[1]) + c.R[6] * (double)n[2]);
	o[1] = (float)(((0.0 + c.R[1] * (double)n[0]) + c.R[4] * (double)n[1]) + c.R[7] * (double)n[2]);
	o[2] = (float)(((0.0 + c.R[2] * (double)n[0]) + c.R[5] * (double)n[1]) + c.R[8] * (double)n[2]);
}
PM_HD float pmfu_conf2weight(float conf, float depth) { const float a = 1.f - conf; return 1.f / ((a > 0.03f ? a : 0.03f) * depth * depth); }
PM_HD int pmfu_round2int(float x) { return (int)pm_floorf(x + .5f); }
PM_HD uint8_t pmfu_toU8(float v) { const int i = pmfu_round2int(v); return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i); }

// the seed's 3D point (float, as stored in PointCloud::points) from pixel p of image A
PM_HD void pmfu_seed_point(const PMFuseCtx& c, uint32_t p, float depth, float* point) {
	const uint32_t wA = (uint32_t)pmfu_w(c, c.A);
	const int i = (int)(p / wA), j = (int)(p % wA);
	double Xw[3]; pmfu_I2W(c.cams[c.A], (double)(float)j, (double)(float)i, (double)depth, Xw);
	point[0] = (float)Xw[0]; point[1] = (float)Xw[1]; point[2] = (float)Xw[2];
}

// cell of neighbour n the point projects to, or -1 (behind the camera / outside); q = its projection
PM_HD int64_t pmfu_target(const PMFuseCtx& c, int n, const float* point, float* q, int* xb, int* yb) {
	pmfu_projectP3(c.cams[c.nb[n]], point, q);
	if (q[2] <= 0) return -1;
	*xb = pmfu_round2int(q[0] / q[2]); *yb = pmfu_round2int(q[1] / q[2]);
	const int wB = pmfu_w(c, c.nb[n]);
	if (!(*xb >= 0 && *yb >= 0 && *xb < wB && *yb < pmfu_h(c, c.nb[n]))) return -1;   // depthMapB.isInside(xB), SceneDensify.cpp:1548
	return (int64_t)c.nb[n] * (int64_t)pmfu_slab(c) + (int64_t)*yb * wB + *xb;
}

// phase 1 of a round: reserve every cell the seed may touch
PM_HD void pmfu_reserve(const PMFuseCtx& c, uint32_t p) {
	const float depth = c.depth[(size_t)c.A * pmfu_slab(c) + p];
	float point[3]; pmfu_seed_point(c, p, depth, point);
	for (int n = 0; n < c.nNb; ++n) {
		float q[3]; int xb, yb;
		const int64_t cell = pmfu_target(c, n, point, q, &xb, &yb);
		if (cell >= 0) PMFU_ATOMIC_MIN(c.resv + cell, p);
	}
}

// phase 2: true if the seed holds all its cells (then it must run pmfu_commit now)
PM_HD bool pmfu_owns(const PMFuseCtx& c, uint32_t p) {
	const float depth = c.depth[(size_t)c.A * pmfu_slab(c) + p];
	float point[3]; pmfu_seed_point(c, p, depth, point);
	for (int n = 0; n < c.nNb; ++n) {
		float q[3]; int xb, yb;
		const int64_t cell = pmfu_target(c, n, point, q, &xb, &yb);
		if (cell >= 0 && c.resv[cell] != p) return false;
	}
	return true;
}

// the reference's per-seed body (SceneDensify.cpp:1513-1606) for seed pixel p of image A, which owns its cells
PM_HD void pmfu_commit(const PMFuseCtx& c, uint32_t p) {
	const size_t P = pmfu_slab(c);
	const uint32_t wA = (uint32_t)pmfu_w(c, c.A);
	const size_t xa = (size_t)c.A * P + p;
	const float depth = c.depth[xa];
	const PMFuseCam& camA = c.cams[c.A];
	float point[3]; pmfu_seed_point(c, p, depth, point);
	uint32_t views[PMFU_MAXV]; float weights[PMFU_MAXV]; uint32_t projs[PMFU_MAXV]; int nViews = 1;
	uint32_t cells[PMFU_MAXV];           // pixel index inside the view, for the rollback / claim
	int64_t invalid[PMFU_MAXNB]; int nInvalid = 0;
	views[0] = (uint32_t)c.A;
	const float w0 = pmfu_conf2weight(c.conf ? c.conf[xa] : 1.f, depth);
	weights[0] = w0; projs[0] = (p % wA) | ((p / wA) << 16); cells[0] = p;
	double confidence = (double)w0;
	float normal[3] = {0.f, 0.f, -1.f};
	if (c.normal) pmfu_normalW(camA, c.normal + xa * 3, normal);
	double X[3]; float Cc[3] = {0.f, 0.f, 0.f}, N[3];
	for (int k = 0; k < 3; ++k) {
		X[k] = (double)(float)((double)point[k] * confidence);
		if (c.bgr) Cc[k] = (float)(confidence * (double)(float)c.bgr[xa * 3 + k]);
		N[k] = (float)((double)normal[k] * confidence);
	}
	const uint32_t tag = p;              // any value != PMFU_NO_ID marks the cell as claimed
	c.claimed[xa] = tag;
	for (int n = 0; n < c.nNb; ++n) {
		const int B = c.nb[n];
		float q[3]; int xb, yb;
		const int64_t cell = pmfu_target(c, n, point, q, &xb, &yb);
		// release the reservation as we go: nobody else can hold it this round, and later rounds must find it free
		if (cell < 0) continue;
		c.resv[cell] = PMFU_FREE;
		const float depthB = c.depth[cell];
		if (depthB == 0) continue;
		if (c.claimed[cell] != PMFU_NO_ID) continue;
		if (pm_fabsf(q[2] - depthB) / q[2] < c.fDepthDiffThreshold) {
			float normalB[3] = {0.f, 0.f, -1.f};
			if (c.normal) pmfu_normalW(c.cams[B], c.normal + (size_t)cell * 3, normalB);
			if (normal[0] * normalB[0] + normal[1] * normalB[1] + normal[2] * normalB[2] > c.normalError) {
				const float confidenceB = pmfu_conf2weight(c.conf ? c.conf[cell] : 1.f, depthB);
				int idx = 0; while (idx < nViews && views[idx] < (uint32_t)B) ++idx;
				for (int m = nViews; m > idx; --m) { views[m] = views[m-1]; weights[m] = weights[m-1]; projs[m] = projs[m-1]; cells[m] = cells[m-1]; }
				views[idx] = (uint32_t)B; weights[idx] = confidenceB; projs[idx] = (uint32_t)xb | ((uint32_t)yb << 16);
				cells[idx] = (uint32_t)((int64_t)yb * pmfu_w(c, B) + xb);
				++nViews;
				c.claimed[cell] = tag;
				double XB[3]; pmfu_I2W(c.cams[B], (double)(float)xb, (double)(float)yb, (double)depthB, XB);
				for (int k = 0; k < 3; ++k) {
					X[k] += XB[k] * (double)confidenceB;
					if (c.bgr && c.bEstimateColor) Cc[k] += (float)c.bgr[(size_t)cell * 3 + k] * confidenceB;
					if (c.bEstimateNormal) N[k] += normalB[k] * confidenceB;
				}
				confidence += (double)confidenceB;
				continue;
			}
		}
		if (q[2] < depthB) invalid[nInvalid++] = cell;
	}
	if ((unsigned)nViews < c.nMinViewsFuse) {
		for (int v = 0; v < nViews; ++v) c.claimed[(size_t)views[v] * P + cells[v]] = PMFU_NO_ID;
		c.recN[p] = 0;
		return;
	}
	const double nrm = 1.0 / confidence;
	c.recN[p] = (uint8_t)nViews;
	for (int k = 0; k < 3; ++k) c.recX[(size_t)k * P + p] = (float)(X[k] * nrm);
	for (int v = 0; v < nViews; ++v) {
		c.recView[(size_t)v * P + p] = views[v]; c.recWeight[(size_t)v * P + p] = weights[v]; c.recProj[(size_t)v * P + p] = projs[v];
	}
	if (c.bEstimateColor) for (int k = 0; k < 3; ++k) c.recColor[(size_t)k * P + p] = pmfu_toU8((float)nrm * Cc[k]);
	if (c.bEstimateNormal) {
		float v[3]; for (int k = 0; k < 3; ++k) v[k] = N[k] * (float)nrm;
		double s = 0; for (int k = 0; k < 3; ++k) s += (double)v[k] * (double)v[k];
		const double nv = __builtin_sqrt(s);
		const double inv = nv ? 1. / nv : 0.;
		for (int k = 0; k < 3; ++k) c.recNormal[(size_t)k * P + p] = (float)((double)v[k] * inv);
	}
	for (int m = 0; m < nInvalid; ++m) c.depth[invalid[m]] = 0.f;
}

// DepthMapsData::MergeDepthMaps (SceneDensify.cpp:1305-1368), used when nMinViewsFuse < 2: every valid depth of image A becomes a
// point with that single view, the image's own colour and the normal of DepthData::GetNormal (R^T n, DepthMap.cpp:137-146); no
// weights are produced (the reference leaves PointCloud::pointWeights empty; the record carries 0).
PM_HD void pmfu_merge(const PMFuseCtx& c, uint32_t p) {
	const size_t P = pmfu_slab(c);
	const uint32_t wA = (uint32_t)pmfu_w(c, c.A);
	const size_t xa = (size_t)c.A * P + p;
	const float depth = c.depth[xa];
	if (depth == 0) { c.recN[p] = 0; return; }
	float point[3]; pmfu_seed_point(c, p, depth, point);
	c.recN[p] = 1;
	for (int k = 0; k < 3; ++k) c.recX[(size_t)k * P + p] = point[k];
	c.recView[p] = (uint32_t)c.A; c.recWeight[p] = 0.f; c.recProj[p] = (p % wA) | ((p / wA) << 16);
	if (c.bEstimateColor) for (int k = 0; k < 3; ++k) c.recColor[(size_t)k * P + p] = c.bgr ? c.bgr[xa * 3 + k] : (uint8_t)0;
	if (c.bEstimateNormal) {
		float n[3] = {0.f, 0.f, -1.f};
		if (c.normal) pmfu_normalW(c.cams[c.A], c.normal + xa * 3, n);
		for (int k = 0; k < 3; ++k) c.recNormal[(size_t)k * P + p] = n[k];
	}
}
