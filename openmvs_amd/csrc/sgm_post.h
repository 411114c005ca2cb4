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

// ---- Disparity2RangeMap (:1350-1444): per low-resolution pixel, the disparity search range for the next tSGM level from the median / minimum /
// maximum of the valid disparities in a 7x7 window (41x41 where the pixel itself has no disparity).  The k-th smallest of up to 1681 int16 values
// is found by bisection on the value (16 counting passes over the window) so that a thread needs no storage.
PM_HD int sgmp_window_count_le(const int16_t* disp, int w, int h, int r, int c, int hw, int v, int* nValid) {
	int n = 0, le = 0;
	for (int i = -hw; i <= hw; ++i) { const int y = r + i; if (y < 0 || y >= h) continue;
		for (int j = -hw; j <= hw; ++j) { const int x = c + j; if (x < 0 || x >= w) continue;
			const int16_t d = disp[(size_t)y * w + x];
			if (d == SGMP_NO_DISP) continue;
			++n; if (d <= v) ++le; } }
	*nValid = n;
	return le;
}
PM_HD int sgmp_window_kth(const int16_t* disp, int w, int h, int r, int c, int hw, int k) {   // k-th smallest (0-based) of the valid values
	int lo = -32768, hi = 32766;     // NO_DISP = 32767 is never a value
	while (lo < hi) {
		const int mid = lo + ((hi - lo) >> 1);
		int n; const int le = sgmp_window_count_le(disp, w, h, r, c, hw, mid, &n);
		if (le > k) hi = mid; else lo = mid + 1;
	}
	return lo;
}
// out: minDisp, maxDisp (both NO_DISP for a masked-out pixel); returns numDisp
PM_HD int sgmp_range_of(const int16_t* disp, int w, int h, const uint8_t* mask2x, int w2, int r, int c, int minNumDisp, int minNumDispInvalid,
		int16_t* oMin, int16_t* oMax) {
	if (mask2x[(size_t)(r * 2 + SGMP_HW) * w2 + (SGMP_HW + 2 * c)] == SGMP_INVALID) { *oMin = SGMP_NO_DISP; *oMax = SGMP_NO_DISP; return 0; }
	const bool bInvalid = disp[(size_t)r * w + c] == SGMP_NO_DISP;
	const int hw = bInvalid ? 20 : 3;
	int n = 0, mn = 32767, mx = -32768;
	for (int i = -hw; i <= hw; ++i) { const int y = r + i; if (y < 0 || y >= h) continue;
		for (int j = -hw; j <= hw; ++j) { const int x = c + j; if (x < 0 || x >= w) continue;
			const int d = disp[(size_t)y * w + x];
			if (d == SGMP_NO_DISP) continue;
			++n; mn = d < mn ? d : mn; mx = d > mx ? d : mx; } }
	int rmin, rmax, numDisp;
	if (n < 3) {
		const int a = (int16_t)(w * 2 / 3);
		rmax = a < minNumDispInvalid ? a : minNumDispInvalid; rmin = (int16_t)(-rmax); numDisp = (int16_t)(rmax - rmin);
	} else {
		int med;
		if (n & 1) med = sgmp_window_kth(disp, w, h, r, c, hw, n >> 1);
		else med = (int16_t)((sgmp_window_kth(disp, w, h, r, c, hw, (n >> 1) - 1) + sgmp_window_kth(disp, w, h, r, c, hw, n >> 1)) / 2);
		const int d2 = (int16_t)(med * 2);
		numDisp = (int16_t)((mx - mn) * 2);
		if (numDisp < minNumDisp) {
			numDisp = minNumDisp; rmin = (int16_t)(d2 - numDisp / 2); rmax = (int16_t)(d2 + (numDisp + 1) / 2);
		} else {
			const int maxNum = bInvalid ? 64 : 32;
			if (numDisp > maxNum) {
				rmin = (int16_t)(d2 - (maxNum * (d2 - mn * 2) + 1) / numDisp);
				rmax = (int16_t)(d2 + (maxNum * (mx * 2 + 1 - d2) + 1) / numDisp);
				numDisp = (int16_t)(rmax - rmin);
			} else { rmin = (int16_t)(d2 - numDisp / 2); rmax = (int16_t)(d2 + (numDisp + 1) / 2); }
		}
	}
	*oMin = (int16_t)rmin; *oMax = (int16_t)rmax;
	return numDisp;
}

// ---- disparity <-> depth between the rectified and the original image (:1815-1923; Image::Disparity2Depth / Depth2Disparity,
// libs/MVS/Image.cpp:367-433; TImage::sampleSafe with a validity functor, libs/Common/Types.inl:2315-2332) -------------------------------
PM_HD int sgmp_clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }
// bilinear sample of the valid neighbours only; valid(v): float maps v > 0, int16 maps v != NO_DISP.  T = float or int16_t / uint16_t
template <typename T, typename VALID>
PM_HD bool sgmp_sample_safe(const T* img, int w, int h, float px, float py, VALID valid, float* out) {
	const int lx = (int)px, ly = (int)py;
	const float x = px - (float)lx, x1 = 1.f - x, y = py - (float)ly, y1 = 1.f - y;
	const T v00 = img[(size_t)sgmp_clampi(ly, h) * w + sgmp_clampi(lx, w)], v10 = img[(size_t)sgmp_clampi(ly, h) * w + sgmp_clampi(lx + 1, w)];
	const T v01 = img[(size_t)sgmp_clampi(ly + 1, h) * w + sgmp_clampi(lx, w)], v11 = img[(size_t)sgmp_clampi(ly + 1, h) * w + sgmp_clampi(lx + 1, w)];
	const bool b00 = valid(v00), b10 = valid(v10), b01 = valid(v01), b11 = valid(v11);
	if (!b00 && !b10 && !b01 && !b11) return false;
	const float a = (float)(b00 ? v00 : (b10 ? v10 : (b01 ? v01 : v11))), b = (float)(b10 ? v10 : (b00 ? v00 : (b11 ? v11 : v01)));
	const float cc = (float)(b01 ? v01 : (b11 ? v11 : (b00 ? v00 : v10))), dd = (float)(b11 ? v11 : (b01 ? v01 : (b10 ? v10 : v00)));
	*out = y1 * (x1 * a + x * b) + y * (x1 * cc + x * dd);
	return true;
}
struct SgmpValidDepth { PM_HD bool operator()(float d) const { return d > 0; } };
struct SgmpValidDisp { PM_HD bool operator()(int16_t d) const { return d != SGMP_NO_DISP; } };
struct SgmpValidCost { PM_HD bool operator()(uint16_t c) const { return c != SGMP_NO_ACCUM; } };
// ProjectVertex_3x3_2_2 with a double matrix and integer input, float output (libs/Common/Util.inl:389-393; INVERT as the reference)
PM_HD void sgmp_project_h(const double* H, int x, int y, float* u) {
	const double z = H[6] * x + H[7] * y + H[8];
	const double invZ = z == 0.0 ? 1e+14 : 1.0 / z;   // INVERT(0) = INV_ZERO, libs/Common/Types.h:562,1234
	u[0] = (float)((H[0] * x + H[1] * y + H[2]) * invZ);
	u[1] = (float)((H[3] * x + H[4] * y + H[5]) * invZ);
}
// Image::Depth2Disparity
PM_HD bool sgmp_depth2disparity(const double* Q, float ux, float uy, float d, float* disparity) {
	const double w = (Q[12] * ux + Q[13] * uy + Q[14]) * d + Q[15];
	if ((w < 0 ? -w : w) < 1e-7) return false;
	const double z = (Q[8] * ux + Q[9] * uy + Q[10]) * d + Q[11];
	*disparity = -(float)(z / w);
	return true;
}
// TDisparity2Depth(Q, u, d)
PM_HD float sgmp_disparity2depth(const double* Q, float ux, float uy, float d) {
	const double w = Q[12] * ux + Q[13] * uy - Q[14] * d + Q[15];
	if ((w < 0 ? -w : w) < 1e-7) return 0.f;
	const double z = Q[8] * ux + Q[9] * uy - Q[10] * d + Q[11];
	const float depth = (float)(z / w);
	return depth < 0.0001f ? 0.f : depth;
}
// Depth2DisparityMap, one pixel (r,c) of the valid-size disparity map
PM_HD int16_t sgmp_depth2disparity_px(const float* depth, int dw, int dh, const double* invH, const double* invQ, int steps, int r, int c) {
	float u[2]; sgmp_project_h(invH, c + SGMP_HW, r + SGMP_HW, u);
	float dep, disp;
	if (!sgmp_sample_safe(depth, dw, dh, u[0], u[1], SgmpValidDepth(), &dep) || !sgmp_depth2disparity(invQ, u[0], u[1], dep, &disp)) return SGMP_NO_DISP;
	return (int16_t)(int)pm_floorf(disp * (float)steps + .5f);
}
// Disparity2DepthMap, one pixel (r,c) of the depth map; cost may be null
PM_HD void sgmp_disparity2depth_px(const int16_t* disp, const uint16_t* cost, int w, int h, const double* H, const double* Q, int steps, int r, int c, float* depth, float* conf) {
	float u[2]; sgmp_project_h(H, c, r, u);
	u[0] -= (float)SGMP_HW; u[1] -= (float)SGMP_HW;
	float d;
	if (!sgmp_sample_safe(disp, w, h, u[0], u[1], SgmpValidDisp(), &d)) { *depth = 0.f; if (cost) *conf = 0.f; return; }
	if (cost) {
		float cst = 0.f;      // (left untouched by the reference when no neighbour is valid; costs of valid disparities are valid)
		sgmp_sample_safe(cost, w, h, u[0], u[1], SgmpValidCost(), &cst);
		*conf = 1.f / (cst + 1.f);
	}
	*depth = sgmp_disparity2depth(Q, u[0], u[1], d / (float)steps);
}

// ---- ProjectDisparity2DepthMap (:1925-2039): every valid disparity is projected into the un-rectified image and offered to the 3x3 pixels
// around it; per pixel and per quadrant (left/right x top/bottom of the projection) the NEAREST projection is kept, the first one in raster
// order on ties.  Parallel form: one 64-bit atomicMin per offer on (float bits of the squared distance << 32 | source index); the resolve pass
// recomputes the winner's values from its index, so nothing but the keys is stored.
#if defined(__HIP_DEVICE_COMPILE__)
#define SGMP_ATOMIC_MIN64(p, v) atomicMin((p), (v))
#else
#define SGMP_ATOMIC_MIN64(p, v) do { if (*(p) > (v)) *(p) = (v); } while (0)
#endif
#define SGMP_KEY_NONE 0xFFFFFFFFFFFFFFFFull
// TDisparity2Depth(Q, ImageRef u, d, pt) (libs/MVS/Image.cpp:391-404)
PM_HD float sgmp_disparity2depth_pt(const double* Q, int ux, int uy, float d, float* px, float* py) {
	const double w = Q[12] * ux + Q[13] * uy - Q[14] * d + Q[15];
	if ((w < 0 ? -w : w) < 1e-7) return 0.f;
	const double z = (Q[8] * ux + Q[9] * uy - Q[10] * d + Q[11]) / w;
	if (z < 1e-7) return 0.f;
	const double nrm = 1.0 / (w * z);
	*px = (float)((Q[0] * ux + Q[1] * uy - Q[2] * d + Q[3]) * nrm);
	*py = (float)((Q[4] * ux + Q[5] * uy - Q[6] * d + Q[7]) * nrm);
	return (float)z;
}
// TDisparity2Depth(Q, ImageRef u, d)
PM_HD float sgmp_disparity2depth_i(const double* Q, int ux, int uy, float d) {
	const double w = Q[12] * ux + Q[13] * uy - Q[14] * d + Q[15];
	if ((w < 0 ? -w : w) < 1e-7) return 0.f;
	const double z = Q[8] * ux + Q[9] * uy - Q[10] * d + Q[11];
	const float depth = (float)(z / w);
	return depth < 0.0001f ? 0.f : depth;
}
struct SgmpProj { float depth, rangeX, rangeY, conf, ux, uy; int x, y; };
// the values source pixel (r,c) of the disparity map contributes; false if it contributes nothing
PM_HD bool sgmp_proj_source(const int16_t* disp, const uint16_t* cost, int w, const double* Q, int steps, int r, int c, SgmpProj* o) {
	const int16_t di = disp[(size_t)r * w + c];
	if (di == SGMP_NO_DISP) return false;
	const float disparity = (float)di / (float)steps;
	const int dx = c + SGMP_HW, dy = r + SGMP_HW;
	float ux = 0.f, uy = 0.f;
	const float depth = sgmp_disparity2depth_pt(Q, dx, dy, disparity, &ux, &uy);
	if (depth <= 0) return false;
	const int center = (int16_t)(int)pm_floorf(disparity);
	o->depth = depth;
	o->rangeX = sgmp_disparity2depth_i(Q, dx, dy, (float)(center - 1));
	o->rangeY = sgmp_disparity2depth_i(Q, dx, dy, (float)(center + 1));
	o->conf = cost ? 1.f / (float)((int)cost[(size_t)r * w + c] + 1) : 0.f;
	o->x = (int)pm_floorf(ux); o->y = (int)pm_floorf(uy);
	o->ux = ux - 0.5f; o->uy = uy - 0.5f;
	return true;
}
PM_HD void sgmp_proj_splat(const int16_t* disp, const uint16_t* cost, int w, const double* Q, int steps, int r, int c, unsigned long long* keys, int dw, int dh) {
	SgmpProj s;
	if (!sgmp_proj_source(disp, cost, w, Q, steps, r, c, &s)) return;
	const unsigned long long src = (unsigned long long)((size_t)r * w + c);
	for (int i = -1; i <= 1; ++i) for (int j = -1; j <= 1; ++j) {
		const int nx = s.x + j, ny = s.y + i;
		if (nx < 0 || ny < 0 || nx >= dw || ny >= dh) continue;
		const float ddx = (float)nx - s.ux, ddy = (float)ny - s.uy;
		if (pm_fabsf(ddx) > 0.75f || pm_fabsf(ddy) > 0.75f) continue;
		const int q = (ddx < 0 ? 1 : 0) + (ddy < 0 ? 2 : 0);
		const float distSq = ddx * ddx + ddy * ddy;
		SGMP_ATOMIC_MIN64(keys + ((size_t)q * dh + ny) * dw + nx, ((unsigned long long)pm_f2u(distSq) << 32) | src);
	}
}
// one pixel of the output maps; returns 1 if a depth was produced
PM_HD int sgmp_proj_resolve(const int16_t* disp, const uint16_t* cost, int w, const double* Q, int steps, const unsigned long long* keys, int dw, int dh,
		int r, int c, float* depth, float* range2, float* conf) {
	float qd[4], qrx[4], qry[4], qc[4], qdist[4];
	for (int q = 0; q < 4; ++q) {
		qd[q] = 0.f; qrx[q] = qry[q] = qc[q] = qdist[q] = 0.f;
		const unsigned long long key = keys[((size_t)q * dh + r) * dw + c];
		if (key == SGMP_KEY_NONE) continue;
		const size_t src = (size_t)(key & 0xFFFFFFFFull);
		SgmpProj s;
		sgmp_proj_source(disp, cost, w, Q, steps, (int)(src / w), (int)(src % w), &s);
		qd[q] = s.depth; qrx[q] = s.rangeX; qry[q] = s.rangeY; qc[q] = s.conf; qdist[q] = pm_u2f((uint32_t)(key >> 32));
	}
	float distCenter = 3.402823466e+38f, depthCenter = 0.f;
	for (int q = 0; q < 4; ++q) { if (qd[q] <= 0) continue; if (distCenter > qdist[q]) { distCenter = qdist[q]; depthCenter = qd[q]; } }
	if (distCenter > 0.75f * 0.75f) { *depth = 0.f; return 0; }
	float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, wsum = 0.f;
	for (int q = 0; q < 4; ++q) {
		if (qd[q] <= 0 || !(pm_fabsf(depthCenter - qd[q]) / depthCenter < 0.02f)) continue;
		const float wq = pm_sqrtf(qdist[q]);
		v0 += qd[q] * wq; v1 += qrx[q] * wq; v2 += qry[q] * wq; v3 += qc[q] * wq; wsum += wq;
	}
	const float inv = 1.f / wsum;
	*depth = v0 * inv; range2[0] = v1 * inv; range2[1] = v2 * inv; *conf = v3 * inv;
	return 1;
}

// ---- SemiGlobalMatcher::Fuse, the per-pixel part (:797-849): cluster the pair depths whose trust ranges contain them, average the largest cluster
#define SGMP_MAX_PAIRS 32
PM_HD void sgmp_fuse_pairs_px(const float* const* depthMaps, const float* const* rangeMaps, const float* const* confMaps, int nPairs, size_t i, unsigned minViews,
		float* depth, float* conf) {
	float cx[SGMP_MAX_PAIRS], cy[SGMP_MAX_PAIRS]; unsigned members[SGMP_MAX_PAIRS]; int cnt[SGMP_MAX_PAIRS]; int nC = 0;   // members: bit p = pair p in the cluster (insertion = ascending p)
	for (int p = 0; p < nPairs; ++p) {
		const float d = depthMaps[p][i];
		if (d <= 0) continue;
		const float rx = rangeMaps[p][i * 2], ry = rangeMaps[p][i * 2 + 1];
		unsigned numClusters = 0;
		for (int k = 0; k < nC; ++k) {
			if (!(cx[k] <= d && d < cy[k])) continue;
			members[k] |= 1u << p; ++cnt[k];
			if (cx[k] < rx) cx[k] = rx;
			if (cy[k] > ry) cy[k] = ry;
			++numClusters;
		}
		if (numClusters == 0) { cx[nC] = rx; cy[nC] = ry; members[nC] = 1u << p; cnt[nC] = 1; ++nC; }
	}
	if (nC == 0) { *depth = 0.f; *conf = 0.f; return; }
	int best = 0;
	for (int k = 1; k < nC; ++k) if (cnt[best] < cnt[k]) best = k;          // std::max_element: first of the largest
	if ((unsigned)cnt[best] < minViews) { *depth = 0.f; *conf = 0.f; return; }
	float ds = 0.f, cs = 0.f; unsigned n = 0;
	for (int p = 0; p < nPairs; ++p) if ((members[best] >> p) & 1u) { ds += depthMaps[p][i]; cs += confMaps[p][i]; ++n; }
	*depth = ds / (float)n; *conf = cs / (float)n;
}

// ---- cv::filterSpeckles(disparityMap, NO_DISP, maxSpeckleSize, maxDiff) as the tSGM loop calls it on the first level (:687-688; OpenCV
// calib3d stereosgbm.cpp filterSpecklesImpl): regions grown over 4-neighbours whose disparities differ by at most maxDiff, step by step; a
// region of at most maxSpeckleSize pixels is erased.  The growth condition is symmetric, so regions are plain connected components:
// lock-free union-find (hook the larger root under the smaller with an atomic min, as in pm_filter.hip), flatten, count, erase.
#if defined(__HIP_DEVICE_COMPILE__)
#define SGMP_ATOMIC_MIN_I(p, v) atomicMin((p), (v))
#define SGMP_ATOMIC_ADD_I(p, v) atomicAdd((p), (v))
#else
PM_HD int sgmp_host_min(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
PM_HD int sgmp_host_add(int* p, int v) { const int o = *p; *p = o + v; return o; }
#define SGMP_ATOMIC_MIN_I(p, v) sgmp_host_min((p), (v))
#define SGMP_ATOMIC_ADD_I(p, v) sgmp_host_add((p), (v))
#endif
PM_HD int sgmp_cc_find(int* parent, int a) {
	int p = parent[a];
	while (p != a) { const int gp = parent[p]; if (gp != p) parent[a] = gp; a = p; p = parent[a]; }   // parents only decrease: safe under races
	return a;
}
PM_HD int sgmp_cc_find_ro(const int* parent, int a) { int p = parent[a]; while (p != a) { a = p; p = parent[a]; } return a; }
PM_HD void sgmp_cc_union(int* parent, int a, int b) {
	for (;;) {
		a = sgmp_cc_find(parent, a); b = sgmp_cc_find(parent, b);
		if (a == b) return;
		if (a > b) { const int t = a; a = b; b = t; }
		const int old = SGMP_ATOMIC_MIN_I(&parent[b], a);
		if (old == b) return;
		b = old;
	}
}
PM_HD void sgmp_speckle_hook(const int16_t* disp, int* parent, int w, int h, int i, int maxDiff) {
	const int16_t d = disp[i];
	if (d == SGMP_NO_DISP) return;
	const int x = i % w, y = i / w;
	if (x + 1 < w) { const int16_t e = disp[i + 1]; if (e != SGMP_NO_DISP) { const int df = (int)d - (int)e; if ((df < 0 ? -df : df) <= maxDiff) sgmp_cc_union(parent, i, i + 1); } }
	if (y + 1 < h) { const int16_t e = disp[i + w]; if (e != SGMP_NO_DISP) { const int df = (int)d - (int)e; if ((df < 0 ? -df : df) <= maxDiff) sgmp_cc_union(parent, i, i + w); } }
}
PM_HD void sgmp_speckle_flatten(int* parent, int* size, int i) { const int r = sgmp_cc_find_ro(parent, i); parent[i] = r; SGMP_ATOMIC_ADD_I(&size[r], 1); }
PM_HD void sgmp_speckle_apply(int16_t* disp, const int* parent, const int* size, int i, int maxSpeckleSize) {
	if (disp[i] != SGMP_NO_DISP && size[parent[i]] <= maxSpeckleSize) disp[i] = SGMP_NO_DISP;
}
