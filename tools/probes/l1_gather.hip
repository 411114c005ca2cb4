// Vector-L1 (TCP) throughput of 16-byte gathers as a function of the number of distinct 128-byte lines a wave-load touches (gfx950).  Every wave reads, over and over, from the same
// 16 KB window of a buffer (L1-resident after the first pass): lane l loads the 16-byte entry at line (l % NL) * stride, offset (l / NL) * 16 inside that line, so one
// global_load_dwordx4 of the wave touches exactly NL lines.  Reported: cycles per wave-load per CU at 4 / 8 / 16 waves per CU, and lines per cycle.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l1_gather.hip -o tools/probes/_build/l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void gather_kernel(const f4* __restrict__ buf, float* out, int iters, int NL, int rot) {
	const int lane = threadIdx.x;
	const f4* win = buf + (size_t)(blockIdx.x & 1) * 1024;             // every wave reads the same two 16 KB windows (1024 entries of 16 B = 128 lines each): L1-resident on every CU
	unsigned idx = (unsigned)(lane % NL) * 8u * (unsigned)(128 / NL > 0 ? 128 / NL : 1) + (unsigned)(lane / NL) % 8u;   // line stride spreads the NL lines over the window
	f4 acc = {0.f, 0.f, 0.f, 0.f};
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const f4 v = win[(idx + (unsigned)((it * 8 + u) * rot) * 8u) & 1023u];      // (the line set moves with the iteration: nothing to hoist)
			acc += v;
		}
	}
	if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
int main() {
	hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
	const int cus = pr.multiProcessorCount; const double hz = (double)pr.clockRate * 1e3;
	f4* d; (void)hipMalloc(&d, (size_t)4096 * 1024 * 16 + 65536); (void)hipMemset(d, 0, (size_t)4096 * 1024 * 16 + 65536);
	float* o; (void)hipMalloc(&o, 64);
	printf("%s, %d CUs, clockRate %d kHz\n", pr.name, cus, pr.clockRate);
	const int iters = 2000;
	for (int NL : {1, 2, 4, 8, 16, 32, 64}) for (int wpc : {4, 8, 16}) {
		const int waves = cus * wpc;
		hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
		hipLaunchKernelGGL(gather_kernel, dim3(waves), dim3(64), 0, 0, d, o, 20, NL, 1);
		(void)hipEventRecord(a, 0);
		hipLaunchKernelGGL(gather_kernel, dim3(waves), dim3(64), 0, 0, d, o, iters, NL, 1);
		(void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
		float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
		const double loadsPerCu = (double)iters * 8 * wpc;
		const double cyc = ms * 1e-3 * hz / loadsPerCu;
		printf("lines per wave-load %2d, %2d waves/CU: %.3f ms, %.1f cycles per wave-load per CU (at %.0f MHz) = %.2f lines/cycle, %.1f B/cycle/CU\n", NL, wpc, ms, cyc, hz / 1e6, NL / cyc, 1024.0 / cyc);
	}
	return 0;
}
