// pm_fuse.hip -- kernels of the depth-map fusion (see pm_fuse.h for the algorithm and its reference, FuseDepthMaps,
// libs/MVS/SceneDensify.cpp:1372-1650).  Per image, best connected first:
//   pmfu_seed_kernel      depth != 0 and unclaimed pixels become pending seeds (and are counted: the reference's nDepths)
//   repeat until no seed is pending:
//     pmfu_reserve_kernel every pending seed atomicMin's its raster index into the neighbour cells its point projects to
//     pmfu_commit_kernel  seeds holding all their cells run the reference's per-seed body; the others go to the next round
//   pmfu_tile_sums / pmfu_scan_tiles / pmfu_scatter_kernel   compaction of the kept points in raster order (= reference numbering)
// All of it is HBM-bound integer/pointer chasing: ~17 projections and a handful of scattered 4-byte accesses per seed per round.
#pragma once
#include "pm_fuse.h"

#define PMFU_TILE 1024          // pixels per scan tile: 256 threads x 4 consecutive pixels
#define PMFU_TB 256

struct PMFuseOut {
	float* points; uint32_t* viewStart; uint32_t* views; float* weights; uint16_t* projs; uint8_t* colors; float* normals;
};

__global__ __launch_bounds__(256) void pmfu_seed_kernel(PMFuseCtx c, uint32_t* pending, uint32_t* nPending, unsigned long long* nDepths) {
	const uint32_t P = (uint32_t)pmfu_w(c, c.A) * (uint32_t)pmfu_h(c, c.A);   // image A's own pixels
	const size_t base = (size_t)c.A * pmfu_slab(c);
	unsigned cnt = 0;
	for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
		c.recN[p] = 0;
		if (c.depth[base + p] == 0.f) continue;
		++cnt;
		if (c.claimed[base + p] != PMFU_NO_ID) continue;
		pending[atomicAdd(nPending, 1u)] = p;
	}
	// wave-level sum before the one atomic per wave
	for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
	if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(nDepths, (unsigned long long)cnt);
}

__global__ __launch_bounds__(256) void pmfu_reserve_kernel(PMFuseCtx c, const uint32_t* pending, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) pmfu_reserve(c, pending[i]);
}

__global__ __launch_bounds__(256) void pmfu_commit_kernel(PMFuseCtx c, const uint32_t* pending, uint32_t n, uint32_t* next, uint32_t* nNext) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t p = pending[i];
	if (pmfu_owns(c, p)) pmfu_commit(c, p);
	else next[atomicAdd(nNext, 1u)] = p;
}

__global__ __launch_bounds__(256) void pmfu_merge_kernel(PMFuseCtx c, unsigned long long* nDepths) {
	const uint32_t P = (uint32_t)pmfu_w(c, c.A) * (uint32_t)pmfu_h(c, c.A);
	unsigned cnt = 0;
	for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) { pmfu_merge(c, p); cnt += c.recN[p]; }
	for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
	if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(nDepths, (unsigned long long)cnt);
}

// ---- compaction ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PMFU_TB) void pmfu_tile_sums(const uint8_t* recN, uint32_t P, uint2* tileSums) {
	__shared__ uint32_t sc[PMFU_TB], sv[PMFU_TB];
	const uint32_t t = threadIdx.x, b0 = blockIdx.x * PMFU_TILE + t * 4;
	uint32_t cc = 0, vv = 0;
	for (uint32_t k = 0; k < 4; ++k) if (b0 + k < P) { const uint32_t n = recN[b0 + k]; cc += n ? 1u : 0u; vv += n; }
	sc[t] = cc; sv[t] = vv;
	__syncthreads();
	for (uint32_t s = PMFU_TB / 2; s > 0; s >>= 1) {
		if (t < s) { sc[t] += sc[t + s]; sv[t] += sv[t + s]; }
		__syncthreads();
	}
	if (t == 0) tileSums[blockIdx.x] = make_uint2(sc[0], sv[0]);
}

// one block: exclusive scan of the tile sums, offset by the running totals of the images fused so far; totals are advanced
__global__ __launch_bounds__(1024) void pmfu_scan_tiles(const uint2* tileSums, uint32_t nTiles, uint32_t* totals, uint2* tileOff) {
	__shared__ uint32_t sc[1024], sv[1024];
	const uint32_t t = threadIdx.x;
	uint32_t carryC = totals[0], carryV = totals[1];
	for (uint32_t start = 0; start < nTiles; start += 1024) {
		const uint32_t i = start + t;
		const uint2 val = i < nTiles ? tileSums[i] : make_uint2(0u, 0u);
		sc[t] = val.x; sv[t] = val.y;
		__syncthreads();
		for (uint32_t off = 1; off < 1024; off <<= 1) {
			const uint32_t ac = t >= off ? sc[t - off] : 0u, av = t >= off ? sv[t - off] : 0u;
			__syncthreads();
			sc[t] += ac; sv[t] += av;
			__syncthreads();
		}
		if (i < nTiles) tileOff[i] = make_uint2(carryC + sc[t] - val.x, carryV + sv[t] - val.y);
		const uint32_t totC = sc[1023], totV = sv[1023];
		__syncthreads();
		carryC += totC; carryV += totV;
	}
	if (t == 0) { totals[0] = carryC; totals[1] = carryV; }
}

__global__ __launch_bounds__(PMFU_TB) void pmfu_scatter_kernel(PMFuseCtx c, const uint2* tileOff, PMFuseOut o) {
	__shared__ uint32_t sc[PMFU_TB], sv[PMFU_TB];
	const uint32_t P = (uint32_t)pmfu_w(c, c.A) * (uint32_t)pmfu_h(c, c.A);
	const size_t S = pmfu_slab(c);                                              // stride of the record arrays
	const uint32_t t = threadIdx.x, b0 = blockIdx.x * PMFU_TILE + t * 4;
	uint32_t n4[4], cc = 0, vv = 0;
	for (uint32_t k = 0; k < 4; ++k) { n4[k] = b0 + k < P ? c.recN[b0 + k] : 0u; cc += n4[k] ? 1u : 0u; vv += n4[k]; }
	sc[t] = cc; sv[t] = vv;
	__syncthreads();
	for (uint32_t off = 1; off < PMFU_TB; off <<= 1) {
		const uint32_t ac = t >= off ? sc[t - off] : 0u, av = t >= off ? sv[t - off] : 0u;
		__syncthreads();
		sc[t] += ac; sv[t] += av;
		__syncthreads();
	}
	const uint2 off0 = tileOff[blockIdx.x];
	uint32_t idx = off0.x + sc[t] - cc, vs = off0.y + sv[t] - vv;
	for (uint32_t k = 0; k < 4; ++k) {
		const uint32_t nv = n4[k];
		if (!nv) continue;
		const uint32_t p = b0 + k;
		o.viewStart[idx] = vs;
		for (int m = 0; m < 3; ++m) o.points[(size_t)idx * 3 + m] = c.recX[(size_t)m * S + p];
		if (o.colors) for (int m = 0; m < 3; ++m) o.colors[(size_t)idx * 3 + m] = c.recColor[(size_t)m * S + p];
		if (o.normals) for (int m = 0; m < 3; ++m) o.normals[(size_t)idx * 3 + m] = c.recNormal[(size_t)m * S + p];
		for (uint32_t v = 0; v < nv; ++v) {
			o.views[vs + v] = c.recView[(size_t)v * S + p];
			o.weights[vs + v] = c.recWeight[(size_t)v * S + p];
			const uint32_t pr = c.recProj[(size_t)v * S + p];
			o.projs[(size_t)(vs + v) * 2] = (uint16_t)(pr & 0xFFFFu); o.projs[(size_t)(vs + v) * 2 + 1] = (uint16_t)(pr >> 16);
		}
		++idx; vs += nv;
	}
}

__global__ void pmfu_fill_u32(uint32_t* p, size_t n, uint32_t v) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ __launch_bounds__(256) void pmfu_count_valid(const float* depth, size_t n, unsigned long long* out) {
	unsigned cnt = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) cnt += depth[i] != 0.f ? 1u : 0u;
	for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
	if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);
}
