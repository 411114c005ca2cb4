// sgm_post.hip -- kernels for the tSGM steps around Match (see sgm_post.h).  All HBM-streaming, one thread per pixel (or per row for
// ExtractMask, whose scan is sequential inside a row and stops after thValid hits).
#pragma once
#include <hip/hip_runtime.h>
#include "sgm_post.h"

__global__ void sgmp_cross_check_kernel(int16_t* l2r, const int16_t* r2l, int wl, int wr, int h, int thCross) {
	const size_t n = (size_t)wl * h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		sgmp_cross_check(l2r, r2l, wl, wr, (int)(i / wl), (int)(i % wl), thCross);
}
__global__ void sgmp_filter_by_cost_kernel(int16_t* disp, const uint16_t* cost, size_t n, uint16_t th) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) sgmp_filter_by_cost(disp, cost, i, th);
}
__global__ void sgmp_extract_mask_kernel(const int16_t* disp, uint8_t* mask, int w, int h, int thValid) {
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < h) sgmp_extract_mask_row(disp, mask, w, r, thValid);
}
__global__ void sgmp_upscale_mask_kernel(const uint8_t* mask, int w, int h, uint8_t* mask2x, int w2, int h2) {
	const size_t n = (size_t)w2 * h2;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		mask2x[i] = sgmp_upscale_mask(mask, w, h, (int)(i / w2), (int)(i % w2));
}
__global__ void sgmp_flip_scatter_kernel(const int16_t* l2r, uint32_t* keys, int w, int h) {
	const size_t n = (size_t)w * h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		sgmp_flip_scatter(l2r, keys, w, (int)(i / w), (int)(i % w));
}
__global__ void sgmp_flip_decode_kernel(const uint32_t* keys, int16_t* r2l, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) r2l[i] = sgmp_flip_decode(keys[i]);
}
// RefineDisparityMap on the resident problem: pixel table + 8-path sums of the last Match
__global__ void sgmp_refine_kernel(const SGMPixel* __restrict__ pixels, const unsigned short* __restrict__ accums, long nPix, short* __restrict__ disp, int mode, int steps) {
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nPix; i += (long)gridDim.x * blockDim.x) {
		const SGMPixel px = pixels[i];
		disp[i] = sgmp_refine(disp[i], px.minDisp, px.maxDisp, accums + px.idx, mode, steps);
	}
}

// Disparity2RangeMap, device part: the range of every low-resolution pixel (the host expands it to the 2x pixel table and its running index)
__global__ void sgmp_range_kernel(const int16_t* disp, int w, int h, const uint8_t* mask2x, int w2, int minNumDisp, int minNumDispInvalid, short2* ranges) {
	const size_t n = (size_t)w * h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		int16_t a, b;
		sgmp_range_of(disp, w, h, mask2x, w2, (int)(i / w), (int)(i % w), minNumDisp, minNumDispInvalid, &a, &b);
		ranges[i] = make_short2(a, b);
	}
}
struct SGMPMat { double m[16]; };
__global__ void sgmp_depth2disparity_kernel(const float* depth, int dw, int dh, SGMPMat invH, SGMPMat invQ, int steps, int16_t* disp, int w, int h) {
	const size_t n = (size_t)w * h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		disp[i] = sgmp_depth2disparity_px(depth, dw, dh, invH.m, invQ.m, steps, (int)(i / w), (int)(i % w));
}
__global__ void sgmp_disparity2depth_kernel(const int16_t* disp, const uint16_t* cost, int w, int h, SGMPMat H, SGMPMat Q, int steps, float* depth, float* conf, int dw, int dh) {
	const size_t n = (size_t)dw * dh;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		float cf = 0.f;
		sgmp_disparity2depth_px(disp, cost, w, h, H.m, Q.m, steps, (int)(i / dw), (int)(i % dw), depth + i, &cf);
		if (cost) conf[i] = cf;
	}
}

__global__ void sgmp_fill_u64(unsigned long long* p, size_t n, unsigned long long v) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void sgmp_proj_splat_kernel(const int16_t* disp, const uint16_t* cost, int w, int h, SGMPMat Q, int steps, unsigned long long* keys, int dw, int dh) {
	const size_t n = (size_t)w * h;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		sgmp_proj_splat(disp, cost, w, Q.m, steps, (int)(i / w), (int)(i % w), keys, dw, dh);
}
__global__ void sgmp_proj_resolve_kernel(const int16_t* disp, const uint16_t* cost, int w, SGMPMat Q, int steps, const unsigned long long* keys, int dw, int dh,
		float* depth, float* range2, float* conf, unsigned* numDepths) {
	const size_t n = (size_t)dw * dh;
	unsigned cnt = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		float cf = 0.f;
		cnt += (unsigned)sgmp_proj_resolve(disp, cost, w, Q.m, steps, keys, dw, dh, (int)(i / dw), (int)(i % dw), depth + i, range2 + i * 2, &cf);
		if (conf) conf[i] = cf;
	}
	if (cnt) atomicAdd(numDepths, cnt);
}
struct SGMPPairs { const float* depth[SGMP_MAX_PAIRS]; const float* range[SGMP_MAX_PAIRS]; const float* conf[SGMP_MAX_PAIRS]; };
__global__ void sgmp_fuse_pairs_kernel(SGMPPairs pr, int nPairs, size_t n, unsigned minViews, float* depth, float* conf) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		sgmp_fuse_pairs_px(pr.depth, pr.range, pr.conf, nPairs, i, minViews, depth + i, conf + i);
}

__global__ void sgmp_speckle_init_kernel(int* parent, int* size, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { parent[i] = i; size[i] = 0; }
}
__global__ void sgmp_speckle_hook_kernel(const int16_t* disp, int* parent, int w, int h, int maxDiff) {
	const int n = w * h;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) sgmp_speckle_hook(disp, parent, w, h, i, maxDiff);
}
__global__ void sgmp_speckle_flatten_kernel(int* parent, int* size, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) sgmp_speckle_flatten(parent, size, i);
}
__global__ void sgmp_speckle_apply_kernel(int16_t* disp, const int* parent, const int* size, int n, int maxSpeckleSize) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) sgmp_speckle_apply(disp, parent, size, i, maxSpeckleSize);
}
