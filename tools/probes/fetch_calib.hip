// FETCH_SIZE calibration on the access patterns of the sweep kernels (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read;
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels over a 2 GiB buffer of 16-byte entries (well past the
// 256 MiB Infinity Cache), each reading a KNOWN number of distinct entries exactly once:
//   calib_stream   every lane one entry, consecutive lanes consecutive entries (the guide's calibration case)
//   calib_gather   every lane one entry at a pseudo-random place (a bijection of the index: each entry read once, no two lanes of a wave in the same 128-byte line)
//   calib_runs     groups of 16 lanes read 16 consecutive entries (256 B), the groups at pseudo-random places -- what a wave of pm_sweep2_kernel<4,2> requests from one run of
//                  an anti-diagonal-major quad image (16 pixels along the diagonal, one tap)
// Run under `rocprofv3 --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum): bytes per dispatch printed here / FETCH_SIZE per dispatch = the correction factor.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o tools/probes/_build/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_stream(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 v = a[i];
	if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = v.x;
}
__global__ void calib_gather(const float4* __restrict__ a, float* __restrict__ out, size_t n, size_t mask) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const size_t j = (i * 0x9E3779B97F4A7C15ull + 0x1234567ull) & mask;   // odd multiplier: a bijection of [0, 2^k)
	const float4 v = a[j];
	if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = v.x;
}
__global__ void calib_runs(const float4* __restrict__ a, float* __restrict__ out, size_t n, size_t mask) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const size_t run = i >> 4, k = i & 15;
	const size_t j = (((run * 0x9E3779B97F4A7C15ull + 0x1234567ull) & (mask >> 4)) << 4) | k;
	const float4 v = a[j];
	if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = v.x;
}

int main(int argc, char** argv) {
	const int logN = argc > 1 ? atoi(argv[1]) : 27;          // 2^27 entries x 16 B = 2 GiB
	const size_t N = (size_t)1 << logN, mask = N - 1;
	float4* a; float* out;
	CK(hipMalloc(&a, N * sizeof(float4))); CK(hipMalloc(&out, 4));
	CK(hipMemset(a, 0, N * sizeof(float4)));
	const size_t n = N / 4;                                  // every kernel reads a quarter of the entries, once each: 512 MiB algorithmic
	const dim3 blk(256), grd((unsigned)((n + 255) / 256));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int rep = 0; rep < 3; ++rep) {
		float ms[3];
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_stream, grd, blk, 0, 0, a, out, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_gather, grd, blk, 0, 0, a, out, n, mask); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_runs, grd, blk, 0, 0, a, out, n, mask); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[2], e0, e1));
		printf("{\"rep\": %d, \"entries_read_per_kernel\": %zu, \"bytes_per_kernel\": %zu, \"stream_ms\": %.3f, \"gather_ms\": %.3f, \"runs_ms\": %.3f, \"stream_GBs\": %.0f, \"gather_GBs\": %.0f, \"runs_GBs\": %.0f}\n",
			rep, n, n * 16, ms[0], ms[1], ms[2], n * 16 / ms[0] * 1e-6, n * 16 / ms[1] * 1e-6, n * 16 / ms[2] * 1e-6);
	}
	CK(hipFree(a)); CK(hipFree(out));
	return 0;
}
