// sgm_post_oracle.cpp -- sequential CPU restatement of the tSGM steps around SemiGlobalMatcher::Match (libs/MVS/SemiGlobalMatcher.cpp in
// /root/reference): ConsistencyCrossCheck :1449-1489, FilterByCost :1491-1514, ExtractMask :1516-1573, FlipDirection :1628-1655,
// UpscaleMask :1657-1690, RefineDisparityMap :1693-1811.  *** TEST INFRASTRUCTURE ONLY ***  PARITY UNPINNED (no reference goldens; the
// reference does not build here).  Written as the reference writes them: raster loops with in-place overwrites; cos/sin through
// csrc/pm_math.h like the product (documented deviation from libm).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../openmvs_amd/csrc/pm_math.h"

namespace {
const int16_t NO_DISP = 32767;
const uint8_t INVALID = 0, VALID = 255;
const int HW = 3;
typedef float real;
real o_cos(real x) { float s, c; pm_sincosf(x, &s, &c); return c; }
real o_sin(real x) { float s, c; pm_sincosf(x, &s, &c); return s; }
struct Fit {
	static real linear(real x) { return x / real(2); }
	static real poly4(real x) { return (x * x * x * x + x) / real(4); }
	static real parabola(real x) { return x / (x + real(1)); }
	static real sine(real x) { return real(0.5) * (o_sin((x - real(1)) * real(1.5707963267948966192313216916398)) + real(1)); }
	static real cosine(real x) { return (real(1) - o_cos(x * (real)(3.1415926535897932384626433832795 / 3.0))); }
	static real lcBlend(real x) { const real factor(real(1.195) - o_cos(x * (real)(3.1415926535897932384626433832795 / 2.3))); return cosine(x) * factor + linear(x) * (real(1) - factor); }
	static real semisubpixel(uint16_t primary, uint16_t other) { return real(0.5) * (static_cast<real>(primary) / static_cast<real>(other)); }
	static real subpixelMode(uint16_t prev, uint16_t center, uint16_t next, int mode) {
		if (prev == center) return center == next ? real(0) : semisubpixel(center, next);
		if (center == next) return prev == center ? real(0) : -semisubpixel(center, prev);
		const uint16_t ld(prev - center), rd(next - center);
		real x, mult;
		if (ld < rd) { x = static_cast<real>(ld) / static_cast<real>(rd); mult = real(1); }
		else { x = static_cast<real>(rd) / static_cast<real>(ld); mult = real(-1); }
		real value(0);
		switch (mode) { case 1: value = linear(x); break; case 2: value = poly4(x); break; case 3: value = parabola(x); break;
			case 4: value = sine(x); break; case 5: value = cosine(x); break; case 6: value = lcBlend(x); break; }
		return (value - real(0.5)) * mult;
	}
};
}

extern "C" {
struct OrcSgmPixel { unsigned long long idx; short minDisp, maxDisp; int pad; };

void orc_sgm_cross_check(int16_t* l2r, const int16_t* r2l, int wl, int h, int wr, int thCross) {
	for (int r = 0; r < h; ++r) for (int c = 0; c < wl; ++c) {
		int16_t& ld = l2r[(size_t)r * wl + c];
		if (ld == NO_DISP) continue;
		const int vx = c + ld;
		if (vx < 0 || vx >= wr) { ld = NO_DISP; continue; }
		const int16_t rd = r2l[(size_t)r * wr + vx];
		if (rd == NO_DISP) { ld = NO_DISP; continue; }
		if (abs(ld + rd) > thCross) ld = NO_DISP;
	}
}
void orc_sgm_filter_by_cost(int16_t* disp, const uint16_t* cost, int w, int h, uint16_t th) {
	for (size_t i = 0; i < (size_t)w * h; ++i) { if (disp[i] == NO_DISP) continue; if (cost[i] > th) disp[i] = NO_DISP; }
}
void orc_sgm_extract_mask(const int16_t* disp, uint8_t* mask, int w, int h, int thValid, int initValid) {
	if (initValid) memset(mask, VALID, (size_t)w * h);
	for (int r = 0; r < h; ++r) { int numValid = 0;
		for (int c = 0; c < w; ++c) { uint8_t& m = mask[(size_t)r * w + c]; if (m == INVALID) continue; m = INVALID; if (disp[(size_t)r * w + c] == NO_DISP) continue; if (++numValid >= thValid) break; } }
	for (int r = 0; r < h; ++r) { int numValid = 0;
		for (int c = w; --c >= 0; ) { uint8_t& m = mask[(size_t)r * w + c]; if (m == INVALID) continue; m = INVALID; if (disp[(size_t)r * w + c] == NO_DISP) continue; if (++numValid >= thValid) break; } }
}
void orc_sgm_upscale_mask(const uint8_t* mask, int w, int h, uint8_t* mask2x, int w2, int h2) {
	memset(mask2x, INVALID, (size_t)w2 * h2);
	for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) {
		const int r2 = r * 2 + HW, c2 = c * 2 + HW; const uint8_t m = mask[(size_t)r * w + c];
		for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { const int ux = c2 + j, uy = r2 + i; if (ux >= 0 && uy >= 0 && ux < w2 && uy < h2) mask2x[(size_t)uy * w2 + ux] = m; }
	}
}
void orc_sgm_flip_direction(const int16_t* l2r, int w, int h, int16_t* r2l) {
	for (size_t i = 0; i < (size_t)w * h; ++i) r2l[i] = NO_DISP;
	for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) {
		const int16_t d = l2r[(size_t)r * w + c];
		if (d == NO_DISP) continue;
		for (int x = (c + d - 1 > 0 ? c + d - 1 : 0), xe = (c + d + 2 < w ? c + d + 2 : w); x < xe; ++x) r2l[(size_t)r * w + x] = (int16_t)-d;
	}
}
void orc_sgm_refine(int16_t* disp, const OrcSgmPixel* pixels, const uint16_t* accums, long nPix, int mode, int steps) {
	if (steps <= 1) return;
	if (mode == 0) { for (long i = 0; i < nPix; ++i) if (disp[i] != NO_DISP) disp[i] = (int16_t)(disp[i] * steps); return; }
	for (long idx = 0; idx < nPix; ++idx) {
		const OrcSgmPixel& pixel = pixels[idx];
		if (pixel.maxDisp - pixel.minDisp < 2) continue;
		int16_t& d = disp[idx];
		if (d == NO_DISP) continue;
		const uint16_t* acc = accums + pixel.idx;
		const int idxDisp = d - pixel.minDisp;
		real disparity((real)d);
		if (d == pixel.minDisp) disparity += Fit::semisubpixel(acc[idxDisp], acc[idxDisp + 1]);
		else if (d + 1 == pixel.maxDisp) disparity -= Fit::semisubpixel(acc[idxDisp], acc[idxDisp - 1]);
		else disparity += Fit::subpixelMode(acc[idxDisp - 1], acc[idxDisp], acc[idxDisp + 1], mode);
		d = (int16_t)(int)floorf(disparity * steps + .5f);
	}
}
} // extern "C"
