// sgm_post_emul.cpp -- host emulation of the tSGM post-processing kernels: the host+device functions of openmvs_amd/csrc/sgm_post.h are
// driven by loops that visit the "threads" in a scrambled order, the way sgm_post.hip's kernels would on the GPU.  tests/test_sgm_post.py
// compares the results with the sequential oracle.  Test code only.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../openmvs_amd/csrc/sgm_post.h"

namespace {
std::vector<size_t> order(size_t n, uint64_t seed) {
	std::vector<size_t> v(n); for (size_t i = 0; i < n; ++i) v[i] = i;
	uint64_t st = seed * 2654435761u + 12345;
	for (size_t i = n; i > 1; --i) { st = st * 6364136223846793005ull + 1442695040888963407ull; std::swap(v[i-1], v[(st >> 33) % i]); }
	return v;
}
}
extern "C" {
struct EmuSgmPixel { unsigned long long idx; short minDisp, maxDisp; int pad; };
void emu_sgm_cross_check(int16_t* l2r, const int16_t* r2l, int wl, int h, int wr, int thCross) {
	for (size_t i : order((size_t)wl * h, 1)) sgmp_cross_check(l2r, r2l, wl, wr, (int)(i / wl), (int)(i % wl), thCross);
}
void emu_sgm_filter_by_cost(int16_t* disp, const uint16_t* cost, int w, int h, uint16_t th) {
	for (size_t i : order((size_t)w * h, 2)) sgmp_filter_by_cost(disp, cost, i, th);
}
void emu_sgm_extract_mask(const int16_t* disp, uint8_t* mask, int w, int h, int thValid, int initValid) {
	if (initValid) memset(mask, 0xFF, (size_t)w * h);
	for (size_t r : order((size_t)h, 3)) sgmp_extract_mask_row(disp, mask, w, (int)r, thValid);
}
void emu_sgm_upscale_mask(const uint8_t* mask, int w, int h, uint8_t* mask2x, int w2, int h2) {
	for (size_t i : order((size_t)w2 * h2, 4)) mask2x[i] = sgmp_upscale_mask(mask, w, h, (int)(i / w2), (int)(i % w2));
}
void emu_sgm_flip_direction(const int16_t* l2r, int w, int h, int16_t* r2l) {
	std::vector<uint32_t> keys((size_t)w * h, 0u);
	for (size_t i : order((size_t)w * h, 5)) sgmp_flip_scatter(l2r, keys.data(), w, (int)(i / w), (int)(i % w));
	for (size_t i : order((size_t)w * h, 6)) r2l[i] = sgmp_flip_decode(keys[i]);
}
void emu_sgm_refine(int16_t* disp, const EmuSgmPixel* pixels, const uint16_t* accums, long nPix, int mode, int steps) {
	if (steps <= 1) return;
	for (size_t i : order((size_t)nPix, 7)) disp[i] = sgmp_refine(disp[i], pixels[i].minDisp, pixels[i].maxDisp, accums + pixels[i].idx, mode, steps);
}
}

extern "C" {
unsigned long long emu_sgm_disparity2range_map(const int16_t* disp, int w, int h, const uint8_t* mask2x, int w2, int h2, int minNumDisp, int minNumDispInvalid,
		EmuSgmPixel* pixels, int* maxNumDisp) {
	std::vector<int16_t> rg((size_t)w * h * 2);
	for (size_t i : order((size_t)w * h, 8)) sgmp_range_of(disp, w, h, mask2x, w2, (int)(i / w), (int)(i % w), minNumDisp, minNumDispInvalid, &rg[i * 2], &rg[i * 2 + 1]);
	unsigned long long total = 0; int mx = 0;      // the host-side expansion of sgmhip_disparity2range_map
	for (int R = 0; R < h2; ++R) { const int rr = R < SGMP_HW + 2 ? 0 : std::min((R - SGMP_HW) / 2, h - 1);
		for (int Cc = 0; Cc < w2; ++Cc) { const int cc = Cc < SGMP_HW + 2 ? 0 : std::min((Cc - SGMP_HW) / 2, w - 1);
			const int16_t lo = rg[((size_t)rr * w + cc) * 2], hi = rg[((size_t)rr * w + cc) * 2 + 1];
			EmuSgmPixel& px = pixels[(size_t)R * w2 + Cc]; px.idx = total; px.minDisp = lo; px.maxDisp = hi; px.pad = 0;
			const int nd = (int16_t)(hi - lo); total += (unsigned long long)(long long)nd; if (nd > mx) mx = nd; } }
	if (maxNumDisp) *maxNumDisp = mx;
	return total;
}
void emu_sgm_depth2disparity_map(const float* depth, int dw, int dh, const double* invH, const double* invQ, int steps, int16_t* disp, int w, int h) {
	for (size_t i : order((size_t)w * h, 9)) disp[i] = sgmp_depth2disparity_px(depth, dw, dh, invH, invQ, steps, (int)(i / w), (int)(i % w));
}
void emu_sgm_disparity2depth_map(const int16_t* disp, const uint16_t* cost, int w, int h, const double* H, const double* Q, int steps, float* depth, float* conf, int dw, int dh) {
	for (size_t i : order((size_t)dw * dh, 10)) { float cf = 0.f; sgmp_disparity2depth_px(disp, cost, w, h, H, Q, steps, (int)(i / dw), (int)(i % dw), depth + i, &cf); if (cost) conf[i] = cf; }
}
}

extern "C" {
int emu_sgm_project_disparity2depth_map(const int16_t* disp, const uint16_t* cost, int w, int h, const double* Q, int steps, float* depth, float* range2, float* conf, int dw, int dh) {
	std::vector<unsigned long long> keys((size_t)dw * dh * 4, SGMP_KEY_NONE);
	for (size_t i : order((size_t)w * h, 11)) sgmp_proj_splat(disp, cost, w, Q, steps, (int)(i / w), (int)(i % w), keys.data(), dw, dh);
	int any = 0;
	memset(range2, 0, (size_t)dw * dh * 8);
	for (size_t i : order((size_t)dw * dh, 12)) { float cf = 0.f; any |= sgmp_proj_resolve(disp, cost, w, Q, steps, keys.data(), dw, dh, (int)(i / dw), (int)(i % dw), depth + i, range2 + i * 2, &cf); if (conf) conf[i] = cf; }
	return any;
}
void emu_sgm_fuse_pairs(const float* const* depthMaps, const float* const* rangeMaps, const float* const* confMaps, int nPairs, int dw, int dh, unsigned minViews, float* depth, float* conf) {
	for (size_t i : order((size_t)dw * dh, 13)) sgmp_fuse_pairs_px(depthMaps, rangeMaps, confMaps, nPairs, i, minViews, depth + i, conf + i);
}
}

extern "C" void emu_sgm_filter_speckles(int16_t* img, int w, int h, int16_t newVal, int maxSpeckleSize, int maxDiff) {
	(void)newVal;                                             // the kernels are specialised for newVal = NO_DISP, as the tSGM loop uses them
	const size_t n = (size_t)w * h;
	std::vector<int> parent(n), size(n, 0);
	for (size_t i = 0; i < n; ++i) parent[i] = (int)i;
	for (size_t i : order(n, 14)) sgmp_speckle_hook(img, parent.data(), w, h, (int)i, maxDiff);
	for (size_t i : order(n, 15)) sgmp_speckle_flatten(parent.data(), size.data(), (int)i);
	for (size_t i : order(n, 16)) sgmp_speckle_apply(img, parent.data(), size.data(), (int)i, maxSpeckleSize);
}
