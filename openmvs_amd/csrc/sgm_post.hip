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
