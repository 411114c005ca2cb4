// Second issue-rate probe (gfx950): the non-arithmetic VALU instructions of the sweep kernels' visit -- selects, compares, min / max, conversions, the pieces of the IEEE
// division expansion, LDS reads -- in wave-instructions per SIMD cycle at 1 / 2 / 4 waves per SIMD, eight independent chains each (tools/probes/valu_rate.hip: the arithmetic ones).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate2.hip -o tools/probes/_build/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define CHAIN8(fmt) fmt(0) "\n" fmt(1) "\n" fmt(2) "\n" fmt(3) "\n" fmt(4) "\n" fmt(5) "\n" fmt(6) "\n" fmt(7)
#define OPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define F_CNDVCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define F_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9"
#define F_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8"
#define F_CMPS(i) "v_cmp_lt_f32_e64 %8, %" #i ", %9"
#define F_MIN(i) "v_min_f32 %" #i ", %" #i ", %8"
#define F_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8"
#define F_MIN3(i) "v_min3_i32 %" #i ", %" #i ", %8, %8"
#define F_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8"
#define F_CVTFI(i) "v_cvt_f32_i32 %" #i ", %" #i
#define F_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8"
#define F_DIVFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %8"
#define F_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %8"
#define F_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %8"
#define F_MOV(i) "v_mov_b32 %" #i ", %8"
#define F_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8"
#define F_SQRT(i) "v_sqrt_f32 %" #i ", %" #i
#define F_EXP(i) "v_exp_f32 %" #i ", %" #i
#define F_FLOOR(i) "v_floor_f32 %" #i ", %" #i
#define F_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8"
#define F_FMAC(i) "v_fmac_f32 %" #i ", %8, %8"
#define F_PAIRV(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define F_PAIRS(i) "v_cmp_lt_f32_e64 %9, %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9"
#define F_CND64VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc"
#define F_PAIRV64(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc"
#define F_MULK(i) "v_mul_f32 %" #i ", 0x3f490000, %" #i
template <int KIND>
__global__ __launch_bounds__(64) void rate_kernel(float* out, int iters, float seed) {
	float a[8];
	for (int i = 0; i < 8; ++i) a[i] = seed + (float)i + (float)threadIdx.x * 1e-3f;
	const float m = 1.0000001f;
	unsigned long long sm = __ballot(threadIdx.x & 1);
	for (int it = 0; it < iters; ++it) {
		if (KIND == 0) { REP16(asm volatile(CHAIN8(F_CNDVCC) : OPS : "v"(m) : "vcc");) }
		if (KIND == 1) { REP16(asm volatile(CHAIN8(F_CNDS) : OPS : "v"(m), "s"(sm));) }
		if (KIND == 2) { REP16(asm volatile(CHAIN8(F_CMP) : OPS : "v"(m) : "vcc");) }
		if (KIND == 3) { REP16(asm volatile(CHAIN8(F_CMPS) : OPS, "+s"(sm) : "v"(m));) }
		if (KIND == 4) { REP16(asm volatile(CHAIN8(F_MIN) : OPS : "v"(m));) }
		if (KIND == 5) { REP16(asm volatile(CHAIN8(F_MAXI) : OPS : "v"(m));) }
		if (KIND == 6) { REP16(asm volatile(CHAIN8(F_MIN3) : OPS : "v"(m));) }
		if (KIND == 7) { REP16(asm volatile(CHAIN8(F_XOR) : OPS : "v"(m));) }
		if (KIND == 8) { REP16(asm volatile(CHAIN8(F_CVTFI) : OPS : "v"(m));) }
		if (KIND == 9) { REP16(asm volatile(CHAIN8(F_CVTFU) : OPS : "v"(m));) }
		if (KIND == 10) { REP16(asm volatile(CHAIN8(F_MAD24) : OPS : "v"(m));) }
		if (KIND == 11) { REP16(asm volatile(CHAIN8(F_DIVFIX) : OPS : "v"(m));) }
		if (KIND == 12) { REP16(asm volatile(CHAIN8(F_DIVFMAS) : OPS : "v"(m) : "vcc");) }
		if (KIND == 13) { REP16(asm volatile(CHAIN8(F_DIVSCALE) : OPS : "v"(m) : "vcc");) }
		if (KIND == 14) { REP16(asm volatile(CHAIN8(F_MOV) : OPS : "v"(m));) }
		if (KIND == 15) { REP16(asm volatile(CHAIN8(F_ADD3) : OPS : "v"(m));) }
		if (KIND == 16) { REP16(asm volatile(CHAIN8(F_SQRT) : OPS : "v"(m));) }
		if (KIND == 17) { REP16(asm volatile(CHAIN8(F_EXP) : OPS : "v"(m));) }
		if (KIND == 18) { REP16(asm volatile(CHAIN8(F_FLOOR) : OPS : "v"(m));) }
		if (KIND == 19) { REP16(asm volatile(CHAIN8(F_ADDF) : OPS : "v"(m));) }
		if (KIND == 20) { REP16(asm volatile(CHAIN8(F_FMAC) : OPS : "v"(m));) }
		if (KIND == 21) { REP16(asm volatile(CHAIN8(F_MULK) : OPS : "v"(m));) }
		if (KIND == 22) { REP16(asm volatile(CHAIN8(F_PAIRV) : OPS : "v"(m) : "vcc");) }
		if (KIND == 23) { unsigned long long t; REP16(asm volatile(CHAIN8(F_PAIRS) : OPS : "v"(m), "s"(sm));) }
		if (KIND == 24) { REP16(asm volatile(CHAIN8(F_CND64VCC) : OPS : "v"(m) : "vcc");) }
		if (KIND == 25) { REP16(asm volatile(CHAIN8(F_PAIRV64) : OPS : "v"(m) : "vcc");) }
	}
	float s = (float)(sm & 3);
	for (int i = 0; i < 8; ++i) s += a[i];
	if (s == 123.456f) out[0] = s;
}
template <int KIND> static void run(const char* name, float* d, double clockHz, int cus) {
	const int iters = 2000;
	for (int wps = 1; wps <= 4; wps *= 2) {
		const int waves = cus * 4 * wps;
		hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
		hipLaunchKernelGGL(rate_kernel<KIND>, dim3(waves), dim3(64), 0, 0, d, 10, 1.f);
		(void)hipEventRecord(a, 0);
		hipLaunchKernelGGL(rate_kernel<KIND>, dim3(waves), dim3(64), 0, 0, d, iters, 1.f);
		(void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
		float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
		const double instrPerSimd = (double)iters * 128.0 * wps;
		printf("%-22s %d wave(s)/SIMD: %.3f ms, %.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", name, wps, ms, ms * 1e-3 * clockHz / instrPerSimd, clockHz / 1e6);
	}
}
int main() {
	hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
	float* d; (void)hipMalloc(&d, 64);
	const double hz = (double)pr.clockRate * 1e3;
	printf("%s, %d CUs, clockRate %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
	run<0>("v_cndmask_b32 vcc", d, hz, pr.multiProcessorCount);
	run<1>("v_cndmask_b32_e64 s[]", d, hz, pr.multiProcessorCount);
	run<2>("v_cmp_lt_f32 vcc", d, hz, pr.multiProcessorCount);
	run<3>("v_cmp_lt_f32_e64 s[]", d, hz, pr.multiProcessorCount);
	run<4>("v_min_f32", d, hz, pr.multiProcessorCount);
	run<5>("v_max_i32", d, hz, pr.multiProcessorCount);
	run<6>("v_min3_i32", d, hz, pr.multiProcessorCount);
	run<7>("v_xor_b32", d, hz, pr.multiProcessorCount);
	run<8>("v_cvt_f32_i32", d, hz, pr.multiProcessorCount);
	run<9>("v_cvt_f32_u32", d, hz, pr.multiProcessorCount);
	run<10>("v_mad_u32_u24", d, hz, pr.multiProcessorCount);
	run<11>("v_div_fixup_f32", d, hz, pr.multiProcessorCount);
	run<12>("v_div_fmas_f32", d, hz, pr.multiProcessorCount);
	run<13>("v_div_scale_f32", d, hz, pr.multiProcessorCount);
	run<14>("v_mov_b32", d, hz, pr.multiProcessorCount);
	run<15>("v_add3_u32", d, hz, pr.multiProcessorCount);
	run<16>("v_sqrt_f32", d, hz, pr.multiProcessorCount);
	run<17>("v_exp_f32", d, hz, pr.multiProcessorCount);
	run<18>("v_floor_f32", d, hz, pr.multiProcessorCount);
	run<19>("v_add_f32", d, hz, pr.multiProcessorCount);
	run<20>("v_fmac_f32", d, hz, pr.multiProcessorCount);
	run<21>("v_mul_f32 literal", d, hz, pr.multiProcessorCount);
	run<22>("cmp vcc + cndmask vcc (2 instr)", d, hz, pr.multiProcessorCount);
	run<23>("cmp s[] + cndmask_e64 s[] (2)", d, hz, pr.multiProcessorCount);
	run<24>("v_cndmask_b32_e64 .. vcc", d, hz, pr.multiProcessorCount);
	run<25>("cmp vcc + cndmask_e64 vcc (2)", d, hz, pr.multiProcessorCount);
	return 0;
}
