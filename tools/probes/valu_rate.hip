// VALU issue-rate probe (gfx950): wave-instructions per cycle per SIMD for plain and packed fp32 ops, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Settles whether v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 move twice the flops of their scalar forms per issue cycle on this part
// (decides the instruction selection of the SGM cost kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(64) void rate_kernel(float* out, int iters, float seed) {
	float a[8]; v2f p[8];
	for (int i = 0; i < 8; ++i) { a[i] = seed + (float)i + (float)threadIdx.x * 1e-3f; p[i] = {a[i], a[i] + 0.5f}; }
	const float m = 1.0000001f; const v2f pm = {m, m};
	unsigned u[8]; unsigned long long q[8]; double dd[8]; const unsigned um = 0xD2511F53u; const double dm = 1.0000001;
	for (int i = 0; i < 8; ++i) { u[i] = (unsigned)(seed * 77.f) + i + threadIdx.x; q[i] = u[i]; dd[i] = a[i]; }
	for (int it = 0; it < iters; ++it) {
		if (KIND == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
		if (KIND == 1) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pm));) }
		if (KIND == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
		if (KIND == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pm));) }
		if (KIND == 4) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
		if (KIND == 5) { REP16(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));) }
		if (KIND == 6) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(um));) }
		if (KIND == 7) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(um));) }
		if (KIND == 8) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\n v_mad_u64_u32 %1, vcc, %8, %8, %1\n v_mad_u64_u32 %2, vcc, %8, %8, %2\n v_mad_u64_u32 %3, vcc, %8, %8, %3\n v_mad_u64_u32 %4, vcc, %8, %8, %4\n v_mad_u64_u32 %5, vcc, %8, %8, %5\n v_mad_u64_u32 %6, vcc, %8, %8, %6\n v_mad_u64_u32 %7, vcc, %8, %8, %7" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(um) : "vcc");) }
		if (KIND == 9) { REP16(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]), "+v"(dd[4]), "+v"(dd[5]), "+v"(dd[6]), "+v"(dd[7]) : "v"(dm));) }
		if (KIND == 10) { REP16(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]), "+v"(dd[4]), "+v"(dd[5]), "+v"(dd[6]), "+v"(dd[7]) : "v"(dm));) }
		if (KIND == 11) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
		if (KIND == 12) { REP16(asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
		if (KIND == 13) { REP16(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
		if (KIND == 14) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]), "+v"(dd[4]), "+v"(dd[5]), "+v"(dd[6]), "+v"(dd[7]) : "v"(dm));) }
		if (KIND == 15) { REP16(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
		if (KIND == 16) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));) }
	}
	float s = 0.f;
	for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i] + (float)q[i] + (float)dd[i];
	if (s == 123.456f) out[0] = s;
}
template <int KIND> static void run(const char* name, float* d, double clockHz, int cus) {
	const int iters = 2000;
	for (int wps = 1; wps <= 4; wps *= 2) {
		const int waves = cus * 4 * wps;
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		hipLaunchKernelGGL(rate_kernel<KIND>, dim3(waves), dim3(64), 0, 0, d, 10, 1.f);
		hipEventRecord(a, 0);
		hipLaunchKernelGGL(rate_kernel<KIND>, dim3(waves), dim3(64), 0, 0, d, iters, 1.f);
		hipEventRecord(b, 0); hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		const double instrPerSimd = (double)iters * 128.0 * wps;          // 16 x 8 instructions per iteration, wps waves share a SIMD
		printf("%-14s %d wave(s)/SIMD: %.3f ms, %.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", name, wps, ms, ms * 1e-3 * clockHz / instrPerSimd, clockHz / 1e6);
	}
}
int main() {
	hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
	float* d; hipMalloc(&d, 64);
	const double hz = (double)pr.clockRate * 1e3;
	printf("%s, %d CUs, clockRate %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
	run<0>("v_mul_f32", d, hz, pr.multiProcessorCount);
	run<1>("v_pk_mul_f32", d, hz, pr.multiProcessorCount);
	run<2>("v_fma_f32", d, hz, pr.multiProcessorCount);
	run<3>("v_pk_fma_f32", d, hz, pr.multiProcessorCount);
	run<4>("v_add_u32", d, hz, pr.multiProcessorCount);
	run<5>("v_mov_b32_dpp", d, hz, pr.multiProcessorCount);
	run<6>("v_mul_lo_u32", d, hz, pr.multiProcessorCount);
	run<7>("v_mul_hi_u32", d, hz, pr.multiProcessorCount);
	run<8>("v_mad_u64_u32", d, hz, pr.multiProcessorCount);
	run<9>("v_mul_f64", d, hz, pr.multiProcessorCount);
	run<10>("v_fma_f64", d, hz, pr.multiProcessorCount);
	run<11>("v_rcp_f32", d, hz, pr.multiProcessorCount);
	run<12>("v_cvt_i32_f32", d, hz, pr.multiProcessorCount);
	run<13>("v_fract_f32", d, hz, pr.multiProcessorCount);
	run<14>("v_rcp_f64", d, hz, pr.multiProcessorCount);
	run<15>("v_sub_f32", d, hz, pr.multiProcessorCount);
	run<16>("v_cndmask_b32", d, hz, pr.multiProcessorCount);
	return 0;
}
