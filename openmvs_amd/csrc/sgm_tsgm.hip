// sgm_tsgm.hip -- the tSGM coarse-to-fine loop of SemiGlobalMatcher::Match(scene, ...) for one rectified pair, resident in HBM
// (reference libs/MVS/SemiGlobalMatcher.cpp:577-706; the same loop as openmvs_amd/tsgm.py, which drives it step by step through host buffers).
// Included by sgm_engine.hip.  Per level: both image pyramids are resampled from the full-resolution images (ViewData::GetImage,
// SemiGlobalMatcher.h:132-141), the disparity maps of the previous level become per-pixel search ranges (FlipDirection, Disparity2RangeMap), the
// pair is matched right->left and left->right, and the results are cross-checked; the first level also removes speckles and derives the masks.
// Only two scalars per Match travel to the host (the size of the cost volume and the widest range), to size the volume.
#pragma once
#include <hip/hip_runtime.h>
#include "sgm_post.h"

// cv::resize(..., INTER_AREA) of an 8-bit BGR image by an integer factor: f = 2 is ResizeAreaFastVec's (a+b+c+d+2)>>2, otherwise
// saturate_cast<uchar>(sum * (1/f^2)) (round half to even)
__global__ void sgmt_area_u8x3_kernel(const unsigned char* __restrict__ src, int sw, unsigned char* __restrict__ dst, int dw, int dh, int f) {
	const size_t n = (size_t)dw * dh * 3;
	const float scale = 1.f / (float)(f * f);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int ch = (int)(i % 3), x = (int)((i / 3) % dw), y = (int)(i / ((size_t)3 * dw));
		int s = 0;
		for (int j = 0; j < f; ++j) for (int k = 0; k < f; ++k) s += src[((size_t)(y * f + j) * sw + (x * f + k)) * 3 + ch];
		int o;
		if (f == 2) o = (s + 2) >> 2;
		else { o = (int)rintf((float)s * scale); o = o < 0 ? 0 : (o > 255 ? 255 : o); }
		dst[i] = (unsigned char)o;
	}
}
// the same for a float image whose size is a multiple of f (OpenCV's ResizeAreaFast: f = 2 pairs the sums, otherwise a running sum, times 1/f^2)
__global__ void sgmt_area_f32_kernel(const float* __restrict__ src, int sw, float* __restrict__ dst, int dw, int dh, int f) {
	const size_t n = (size_t)dw * dh;
	const float scale = 1.f / (float)(f * f);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % dw), y = (int)(i / dw);
		const float* p = src + (size_t)(y * f) * sw + x * f;
		float o;
		if (f == 2) o = ((p[0] + p[1]) + (p[sw] + p[sw + 1])) * 0.25f;
		else { float sum = 0.f; for (int j = 0; j < f; ++j) for (int k = 0; k < f; ++k) sum += p[(size_t)j * sw + k]; o = sum * scale; }
		dst[i] = o;
	}
}
// first level: cv::resize(mask, size, INTER_NEAREST) of the full-resolution mask (size * f == full size) and the crop to the valid grid (:627-631)
__global__ void sgmt_mask_first_kernel(const unsigned char* __restrict__ mask, int W, unsigned char* __restrict__ out, int vw, int vh, int f) {
	const size_t n = (size_t)vw * vh;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int c = (int)(i % vw), r = (int)(i / vw);
		out[i] = mask[(size_t)((r + SGMP_HW) * f) * W + (size_t)(c + SGMP_HW) * f];
	}
}
__global__ void sgmt_fill_i16_kernel(short* p, size_t n, short v) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// Disparity2RangeMap's expansion to the pixel table of the 2x level (:1409-1441), on the device: 2x pixel (R, C) takes the range of low-resolution
// pixel (R < HW+2 ? 0 : min((R-HW)/2, h-1), likewise for C); idx is the running sum of numDisp in raster order = an exclusive scan in three passes.
#define SGMT_TILE 1024
__device__ __forceinline__ int sgmt_nd(const short2* ranges, int w, int h, int w2, size_t i, short* lo, short* hi) {
	const int R = (int)(i / w2), Cc = (int)(i % w2);
	const int rr = R < SGMP_HW + 2 ? 0 : min((R - SGMP_HW) / 2, h - 1), cc = Cc < SGMP_HW + 2 ? 0 : min((Cc - SGMP_HW) / 2, w - 1);
	const short2 rg = ranges[(size_t)rr * w + cc];
	*lo = rg.x; *hi = rg.y;
	return (int)(short)(rg.y - rg.x);
}
__global__ __launch_bounds__(256) void sgmt_tile_sums_kernel(const short2* __restrict__ ranges, int w, int h, int w2, size_t n2, unsigned long long* __restrict__ tileSums, int* __restrict__ maxNd) {
	__shared__ unsigned long long s_sum[256];
	__shared__ int s_max[256];
	const size_t base = (size_t)blockIdx.x * SGMT_TILE;
	unsigned long long sum = 0; int mx = 0;
	for (int k = 0; k < SGMT_TILE / 256; ++k) {
		const size_t i = base + (size_t)threadIdx.x * (SGMT_TILE / 256) + k;
		if (i < n2) { short lo, hi; const int nd = sgmt_nd(ranges, w, h, w2, i, &lo, &hi); sum += (unsigned long long)(long long)nd; mx = max(mx, nd); }
	}
	s_sum[threadIdx.x] = sum; s_max[threadIdx.x] = mx;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if ((int)threadIdx.x < st) { s_sum[threadIdx.x] += s_sum[threadIdx.x + st]; s_max[threadIdx.x] = max(s_max[threadIdx.x], s_max[threadIdx.x + st]); }
		__syncthreads();
	}
	if (threadIdx.x == 0) { tileSums[blockIdx.x] = s_sum[0]; if (s_max[0] > 0) atomicMax(maxNd, s_max[0]); }
}
// exclusive scan of the tile sums by one workgroup (a few thousand entries); total[0] receives the grand total
__global__ __launch_bounds__(256) void sgmt_scan_tiles_kernel(unsigned long long* __restrict__ tileSums, int nTiles, unsigned long long* __restrict__ total) {
	__shared__ unsigned long long s_v[256];
	__shared__ unsigned long long s_carry;
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (int base = 0; base < nTiles; base += 256) {
		const int i = base + (int)threadIdx.x;
		const unsigned long long v = i < nTiles ? tileSums[i] : 0ull;
		s_v[threadIdx.x] = v;
		__syncthreads();
		for (int st = 1; st < 256; st <<= 1) {                      // Hillis-Steele inclusive scan
			const unsigned long long a = (int)threadIdx.x >= st ? s_v[threadIdx.x - st] : 0ull;
			__syncthreads();
			s_v[threadIdx.x] += a;
			__syncthreads();
		}
		const unsigned long long carry = s_carry;
		if (i < nTiles) tileSums[i] = carry + s_v[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 255) s_carry = carry + s_v[255];
		__syncthreads();
	}
	if (threadIdx.x == 0) total[0] = s_carry;
}
__global__ __launch_bounds__(256) void sgmt_expand_kernel(const short2* __restrict__ ranges, int w, int h, int w2, size_t n2, const unsigned long long* __restrict__ tileSums, SGMPixel* __restrict__ pixels) {
	__shared__ unsigned long long s_v[256];
	const size_t base = (size_t)blockIdx.x * SGMT_TILE;
	constexpr int PER = SGMT_TILE / 256;
	short lo[PER], hi[PER]; int nd[PER];
	unsigned long long sum = 0;
	for (int k = 0; k < PER; ++k) {
		const size_t i = base + (size_t)threadIdx.x * PER + k;
		nd[k] = 0; lo[k] = hi[k] = 0;
		if (i < n2) nd[k] = sgmt_nd(ranges, w, h, w2, i, &lo[k], &hi[k]);
		sum += (unsigned long long)(long long)nd[k];
	}
	s_v[threadIdx.x] = sum;
	__syncthreads();
	for (int st = 1; st < 256; st <<= 1) {
		const unsigned long long a = (int)threadIdx.x >= st ? s_v[threadIdx.x - st] : 0ull;
		__syncthreads();
		s_v[threadIdx.x] += a;
		__syncthreads();
	}
	unsigned long long run = tileSums[blockIdx.x] + s_v[threadIdx.x] - sum;
	for (int k = 0; k < PER; ++k) {
		const size_t i = base + (size_t)threadIdx.x * PER + k;
		if (i < n2) { SGMPixel px; px.idx = run; px.minDisp = lo[k]; px.maxDisp = hi[k]; px.pad = 0; pixels[i] = px; }
		run += (unsigned long long)(long long)nd[k];
	}
}
