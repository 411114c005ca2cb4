// LDS gather throughput (gfx950): cycles per wave-instruction per CU for ds_read_b32 / ds_read2_b32 / ds_read_b64 / ds_read_b128 whose 64 lanes read pseudo-random places of a
// 12 KB window -- what a tap row would cost if the source window of a wave were staged in LDS ("what would move the bound") -- next to the 64.5 cycles a
// scattered wave-load costs the texture-address unit (tools/probes/l1_gather.hip).  Addresses change every trip (a cheap LCG per lane), so nothing is hoisted.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_gather.hip -o tools/probes/_build/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(64) void lds_kernel(float* out, int iters) {
	__shared__ float win[3072];                       // 12 KB
	for (int i = threadIdx.x; i < 3072; i += 64) win[i] = (float)i;
	__syncthreads();
	unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	float acc = 0.f;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			s = s * 1664525u + 1013904223u;
			const unsigned a = (s >> 8) % 3000u;      // dword index
			if (KIND == 0) acc += win[a];
			if (KIND == 1) { acc += win[a] + win[a + 61]; }                                   // two dwords 61 apart: one ds_read2_b32
			if (KIND == 2) { const float2 v = *(const float2*)&win[a & ~1u]; acc += v.x + v.y; }
			if (KIND == 3) { const float4 v = *(const float4*)&win[a & ~3u]; acc += v.x + v.y + v.z + v.w; }
			if (KIND == 4) { acc += win[a] + win[a + 1] + win[a + 61] + win[a + 62]; }       // a bilinear sample from a plain window: two ds_read2_b32 (or four reads)
		}
	}
	if (acc == 123.456f) out[0] = acc;
}
template <int KIND> static void run(const char* what, float* o, int cus, double hz) {
	for (int wpc : {4, 8, 16}) {
		const int iters = 2000, waves = cus * wpc;
		hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
		hipLaunchKernelGGL(lds_kernel<KIND>, dim3(waves), dim3(64), 0, 0, o, 20);
		(void)hipEventRecord(a, 0);
		hipLaunchKernelGGL(lds_kernel<KIND>, dim3(waves), dim3(64), 0, 0, o, iters);
		(void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
		float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
		printf("%-44s %2d waves/CU: %.3f ms, %5.1f cycles per gather per CU (at %.0f MHz)\n", what, wpc, ms, ms * 1e-3 * hz / ((double)iters * 8 * wpc), hz / 1e6);
	}
}
int main() {
	hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
	const int cus = pr.multiProcessorCount; const double hz = (double)pr.clockRate * 1e3;
	float* o; (void)hipMalloc(&o, 64);
	printf("%s, %d CUs, clockRate %d kHz\n", pr.name, cus, pr.clockRate);
	run<0>("ds_read_b32, random dwords", o, cus, hz);
	run<1>("ds_read2_b32, two dwords 61 apart", o, cus, hz);
	run<2>("ds_read_b64, random aligned pairs", o, cus, hz);
	run<3>("ds_read_b128, random aligned quads", o, cus, hz);
	run<4>("bilinear sample: 4 dwords (2 x ds_read2_b32)", o, cus, hz);
	return 0;
}
