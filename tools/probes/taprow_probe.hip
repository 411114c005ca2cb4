// Calibration probe (measurement tool, not part of the product): how fast does one SIMD execute the sweep kernel's fast tap row
// (pm_tap_row_lds, the code the product runs) as a function of the number of resident waves?  Every wave runs `rows` tap rows back to back on a
// synthetic window; the dynamic LDS allocation sets how many one-wave workgroups fit a CU.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -shared -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/probes/taprow_probe.hip -o tools/probes/libtaprow_probe.so
#include "../../openmvs_amd/csrc/pm_kernels.hip"

template <int MODE>
__global__ __launch_bounds__(64, 3) void taprow_probe_kernel(float* out, int rows, float jitter) {
	constexpr int TC = 8 + PM_TCX;
	constexpr int TSTRIDE = PM_TR * TC + PM_TILE_PAD;
	extern __shared__ float dyn[];                       // occupancy control only
	__shared__ float s_tile[8 * TSTRIDE];
	__shared__ float2 s_w[8][PM_NT + 1];
	const int lane = threadIdx.x, g = lane / 8, v = lane % 8;
	for (int i = lane; i < 8 * TSTRIDE; i += 64) s_tile[i] = 0.25f + 0.001f * (float)(i % 97);
	for (int i = lane; i < 8 * (PM_NT + 1); i += 64) s_w[i / (PM_NT + 1)][i % (PM_NT + 1)] = make_float2(0.04f, 0.01f * (float)(i % 7));
	if (rows < 0) dyn[lane] = 1.f;
	__syncthreads();
	// a fronto-parallel mapping with unit scale: the 5 x 5 taps (step 2) of pixel g land inside the window of view v
	const float h0 = 2.f, h3 = 0.f, h6 = 0.f;
	const int ts0 = 100, tt0 = 50;
	float sum = 0.f, sumSq = 0.f, num = 0.f; bool oob = false;
	unsigned ok = 0;
	for (int r = 0; r < rows; ++r) {
		const int i = r % 5;
		// pixel g of the wave sits at (46 + g, 62 - g): first tap of row i at (42 + g, 58 - g + 2 i); skew row = x + y - ts0 in [0, 16], column = y - tt0 in [1, 16]
		// jitter: per-view offset of the footprint inside its window (real views differ in their sub-window position: LDS bank alignment between the views)
		const float X0 = 42.37f + (float)g + jitter * (float)(v & 1), X1 = 58.61f - (float)g + (float)(2 * i) - jitter * (float)(v & 1), X2 = 1.f;
		if (MODE == 0) ok += pm_tap_row_lds<TC>(s_tile + v * TSTRIDE, ts0, tt0, 4096, 4096, true, h0, h3, h6, X0, X1, X2, s_w[g] + i * 5, sum, sumSq, num, oob) ? 1u : 0u;
		else { sum += X0 * X1; sumSq += sum * 1.0001f; num += sumSq; }     // MODE 1: empty loop (overhead reference)
	}
	out[blockIdx.x * 64 + lane] = sum + sumSq + num + (float)ok;
}

extern "C" int taprow_probe(int blocks, int rows, int dynLdsBytes, int mode, float jitter, float* msOut, float* okFrac) {
	float* d = nullptr;
	if (hipMalloc(&d, sizeof(float) * 64 * (size_t)blocks) != hipSuccess) return -1;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int rep = 0; rep < 2; ++rep) {
		hipEventRecord(a, 0);
		if (mode == 0) hipLaunchKernelGGL(taprow_probe_kernel<0>, dim3(blocks), dim3(64), dynLdsBytes, 0, d, rows, jitter);
		else hipLaunchKernelGGL(taprow_probe_kernel<1>, dim3(blocks), dim3(64), dynLdsBytes, 0, d, rows, jitter);
		hipEventRecord(b, 0);
		if (hipEventSynchronize(b) != hipSuccess) return -2;
	}
	hipEventElapsedTime(msOut, a, b);
	float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	*okFrac = h[0];
	hipFree(d); hipEventDestroy(a); hipEventDestroy(b);
	return 0;
}
