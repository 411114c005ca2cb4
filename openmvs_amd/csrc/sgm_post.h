// sgm_post.h -- the per-pixel / per-row steps around SemiGlobalMatcher::Match in the tSGM loop (libs/MVS/SemiGlobalMatcher.cpp in
// /root/reference): ConsistencyCrossCheck (:1449-1489), FilterByCost (:1491-1514), ExtractMask (:1516-1573), FlipDirection (:1628-1655),
// UpscaleMask (:1657-1690), RefineDisparityMap (:1693-1811).  Host+device inline functions: sgm_post.hip wraps them in kernels and
// tests/cpp/sgm_post_emul.cpp drives the same functions from a host loop with scrambled thread orders, so that the parallel forms are
// checked against the sequential oracle (oracle/sgm_post_oracle.cpp) on the CPU.
// Integer work is exact.  RefineDisparityMap's SINE / COSINE / LC_BLEND fits call cos/sin: both sides use pm_math.h's Cephes kernels
// instead of libm (the documented transcendental deviation of this repository, see pm_math.h).
#pragma once
#include <stdint.h>
#include "pm_math.h"

#define SGMP_NO_DISP ((int16_t)32767)       // SemiGlobalMatcher::NO_DISP, SemiGlobalMatcher.h:68
#define SGMP_NO_ACCUM ((uint16_t)65535)     // NO_ACCUMCOST, :69
#define SGMP_VALID ((uint8_t)255)           // MaskMap values, :67
#define SGMP_INVALID ((uint8_t)0)
#define SGMP_HW 3                           // halfWindowSizeX/Y

#if defined(__HIP_DEVICE_COMPILE__)
#define SGMP_ATOMIC_MAX(p, v) atomicMax((p), (v))
#else
#define SGMP_ATOMIC_MAX(p, v) do { if (*(p) < (v)) *(p) = (v); } while (0)
#endif

// ConsistencyCrossCheck: one pixel of l2r (wl x h) against r2l (wr x h)
PM_HD void sgmp_cross_check(int16_t* l2r, const int16_t* r2l, int wl, int wr, int r, int c, int thCross) {
	int16_t& ld = l2r[(size_t)r * wl + c];
	if (ld == SGMP_NO_DISP) return;
	const int vx = c + ld;
	if (vx < 0 || vx >= wr) { ld = SGMP_NO_DISP; return; }
	const int16_t rd = r2l[(size_t)r * wr + vx];
	if (rd == SGMP_NO_DISP) { ld = SGMP_NO_DISP; return; }
	const int s = (int)ld + (int)rd;
	if ((s < 0 ? -s : s) > thCross) ld = SGMP_NO_DISP;
}

// FilterByCost
PM_HD void sgmp_filter_by_cost(int16_t* disp, const uint16_t* cost, size_t i, uint16_t th) {
	if (disp[i] != SGMP_NO_DISP && cost[i] > th) disp[i] = SGMP_NO_DISP;
}

// ExtractMask, one row: left-to-right then right-to-left scan (the two passes of the reference touch disjoint... not necessarily
// disjoint pixels, so the order left-then-right is kept inside the row's thread)
PM_HD void sgmp_extract_mask_row(const int16_t* disp, uint8_t* mask, int w, int r, int thValid) {
	const int16_t* d = disp + (size_t)r * w; uint8_t* m = mask + (size_t)r * w;
	int numValid = 0;
	for (int c = 0; c < w; ++c) {
		if (m[c] == SGMP_INVALID) continue;
		m[c] = SGMP_INVALID;
		if (d[c] == SGMP_NO_DISP) continue;
		if (++numValid >= thValid) break;
	}
	numValid = 0;
	for (int c = w; --c >= 0; ) {
		if (m[c] == SGMP_INVALID) continue;
		m[c] = SGMP_INVALID;
		if (d[c] == SGMP_NO_DISP) continue;
		if (++numValid >= thValid) break;
	}
}

// UpscaleMask as a gather: destination pixel (r2, c2) of the w2 x h2 mask
PM_HD uint8_t sgmp_upscale_mask(const uint8_t* mask, int w, int h, int r2, int c2) {
	if (r2 < SGMP_HW || c2 < SGMP_HW) return SGMP_INVALID;
	const int r = (r2 - SGMP_HW) >> 1, c = (c2 - SGMP_HW) >> 1;
	if (r >= h || c >= w) return SGMP_INVALID;
	return mask[(size_t)r * w + c];
}

// FlipDirection as a scatter with a deterministic winner: the sequential loop lets the pixel with the largest column overwrite the
// others, so each source pixel offers (column + 1) << 16 | (uint16)(-d) to its up-to-three targets with an atomic max.
PM_HD void sgmp_flip_scatter(const int16_t* l2r, uint32_t* keys, int w, int r, int c) {
	const int16_t d = l2r[(size_t)r * w + c];
	if (d == SGMP_NO_DISP) return;
	const int x0 = c + d - 1 > 0 ? c + d - 1 : 0, x1 = c + d + 2 < w ? c + d + 2 : w;
	const uint32_t key = ((uint32_t)(c + 1) << 16) | (uint32_t)(uint16_t)(int16_t)(-d);
	for (int x = x0; x < x1; ++x) SGMP_ATOMIC_MAX(keys + (size_t)r * w + x, key);
}
PM_HD int16_t sgmp_flip_decode(uint32_t key) { return key == 0u ? SGMP_NO_DISP : (int16_t)(uint16_t)(key & 0xFFFFu); }

// RefineDisparityMap, the sub-pixel fits (:1719-1771)
enum { SGMP_SUBPIXEL_NA = 0, SGMP_SUBPIXEL_LINEAR, SGMP_SUBPIXEL_POLY4, SGMP_SUBPIXEL_PARABOLA, SGMP_SUBPIXEL_SINE, SGMP_SUBPIXEL_COSINE, SGMP_SUBPIXEL_LC_BLEND };
PM_HD float sgmp_cos(float x) { float s, c; pm_sincosf(x, &s, &c); return c; }
PM_HD float sgmp_sin(float x) { float s, c; pm_sincosf(x, &s, &c); return s; }
PM_HD float sgmp_semisubpixel(uint16_t primary, uint16_t other) { return 0.5f * ((float)primary / (float)other); }
PM_HD float sgmp_fit_linear(float x) { return x / 2.f; }
PM_HD float sgmp_fit_cosine(float x) { return 1.f - sgmp_cos(x * (float)(3.1415926535897932384626433832795 / 3.0)); }
PM_HD float sgmp_subpixel(uint16_t prev, uint16_t center, uint16_t next, int mode) {
	if (prev == center) return center == next ? 0.f : sgmp_semisubpixel(center, next);
	if (center == next) return prev == center ? 0.f : -sgmp_semisubpixel(center, prev);
	const uint16_t ld = (uint16_t)(prev - center), rd = (uint16_t)(next - center);   // AccumCost arithmetic is uint16 (wraps like the reference)
	float x, mult;
	if (ld < rd) { x = (float)ld / (float)rd; mult = 1.f; } else { x = (float)rd / (float)ld; mult = -1.f; }
	float value = 0.f;
	switch (mode) {
	case SGMP_SUBPIXEL_LINEAR: value = sgmp_fit_linear(x); break;
	case SGMP_SUBPIXEL_POLY4: value = (x * x * x * x + x) / 4.f; break;
	case SGMP_SUBPIXEL_PARABOLA: value = x / (x + 1.f); break;
	case SGMP_SUBPIXEL_SINE: value = 0.5f * (sgmp_sin((x - 1.f) * (float)1.5707963267948966192313216916398) + 1.f); break;
	case SGMP_SUBPIXEL_COSINE: value = sgmp_fit_cosine(x); break;
	case SGMP_SUBPIXEL_LC_BLEND: {
		const float factor = 1.195f - sgmp_cos(x * (float)(3.1415926535897932384626433832795 / 2.3));
		value = sgmp_fit_cosine(x) * factor + sgmp_fit_linear(x) * (1.f - factor);
	} break;
	default: break;
	}
	return (value - 0.5f) * mult;
}
// one pixel: disparity in/out, accums = the 8-path sums of this pixel (numDisp entries starting at its minDisp)
PM_HD int16_t sgmp_refine(int16_t d, int minDisp, int maxDisp, const uint16_t* accums, int mode, int steps) {
	if (d == SGMP_NO_DISP) return d;
	if (mode == SGMP_SUBPIXEL_NA) return (int16_t)(d * steps);
	if (maxDisp - minDisp < 2) return d;
	const int i = d - minDisp;
	float disparity = (float)d;
	if (d == minDisp) disparity += sgmp_semisubpixel(accums[i], accums[i + 1]);
	else if (d + 1 == maxDisp) disparity -= sgmp_semisubpixel(accums[i], accums[i - 1]);
	else disparity += sgmp_subpixel(accums[i - 1], accums[i], accums[i + 1], mode);
	return (int16_t)(int)pm_floorf(disparity * (float)steps + .5f);
}
