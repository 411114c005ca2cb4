// Vector-L1 / texture-address throughput of gathers (gfx950): cycles per wave-load per CU as a function of (a) how many distinct 128-byte lines a wave-load touches and how
// the lanes are laid over them, (b) the width of the load (4 / 8 / 16 bytes per lane), (c) the share of active lanes.  Every wave reads, over and over, from two 16 KB windows
// (L1-resident on every CU).  Lane l loads the entry at line (l % NL) * stride, slot (l / NL) % 8 of that line, so a wave-load touches exactly NL lines and the four lanes of
// a quad touch min(NL, 4) of them.  What the sweep kernels' tap rows do is the last row of each table: 64 lines, 16 bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l1_gather.hip -o tools/probes/_build/l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T>
__global__ __launch_bounds__(64) void gather_kernel(const float* __restrict__ buf, float* out, int iters, int NL, int activeLanes) {
	const int lane = threadIdx.x;
	if (lane >= activeLanes) return;
	const char* win = (const char*)buf + (size_t)(blockIdx.x & 1) * 16384;
	const unsigned idx = (unsigned)(lane % NL) * 8u * (unsigned)(128 / NL > 0 ? 128 / NL : 1) + (unsigned)(lane / NL) % 8u;   // in 16-byte entries
	T acc = {};
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const T v = *(const T*)(win + (size_t)((idx + (unsigned)(it * 8 + u) * 8u) & 1023u) * 16u);      // (the line set moves with the iteration: nothing to hoist)
			acc += v;
		}
	}
	float s = 0.f;
	for (unsigned i = 0; i < sizeof(T) / 4; ++i) s += ((const float*)&acc)[i];
	if (s == 123.456f) out[0] = s;
}
typedef float f1 __attribute__((ext_vector_type(1)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <class T> static void run(const char* what, const float* d, float* o, int cus, double hz, int NL, int wpc, int active) {
	const int iters = 1000, waves = cus * wpc;
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	hipLaunchKernelGGL(gather_kernel<T>, dim3(waves), dim3(64), 0, 0, d, o, 20, NL, active);
	(void)hipEventRecord(a, 0);
	hipLaunchKernelGGL(gather_kernel<T>, dim3(waves), dim3(64), 0, 0, d, o, iters, NL, active);
	(void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
	float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
	const double cyc = ms * 1e-3 * hz / ((double)iters * 8 * wpc);
	printf("%-6s lines per wave-load %2d, active lanes %2d, %2d waves/CU: %.3f ms, %5.1f cycles per wave-load per CU (at %.0f MHz), %.2f lanes/cycle, %.1f B/cycle/CU\n", what, NL, active, wpc, ms, cyc, hz / 1e6,
	       active / cyc, active * sizeof(T) / cyc);
}
int main() {
	hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
	const int cus = pr.multiProcessorCount; const double hz = (double)pr.clockRate * 1e3;
	float* d; (void)hipMalloc(&d, 65536); (void)hipMemset(d, 0, 65536);
	float* o; (void)hipMalloc(&o, 64);
	printf("%s, %d CUs, clockRate %d kHz\n", pr.name, cus, pr.clockRate);
	for (int NL : {1, 2, 4, 8, 64}) for (int wpc : {4, 16}) run<f4>("16 B", d, o, cus, hz, NL, wpc, 64);
	for (int NL : {1, 2, 4, 64}) run<f2>("8 B", d, o, cus, hz, NL, 16, 64);
	for (int NL : {1, 2, 4, 64}) run<f1>("4 B", d, o, cus, hz, NL, 16, 64);
	for (int act : {48, 32, 16}) run<f4>("16 B", d, o, cus, hz, 64, 16, act);
	return 0;
}
