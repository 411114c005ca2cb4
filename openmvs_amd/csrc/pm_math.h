// pm_math.h -- arithmetic primitives shared by the HIP kernels and the CPU oracle.
//
// Why this header exists: PatchMatch's accept-if-better comparisons amplify a
// 1-ulp difference in a score into a different plane, so the GPU path and the
// CPU oracle must evaluate every transcendental with the *same* sequence of
// IEEE-754 binary32/binary64 operations.  libm (glibc) and ROCm's ocml differ in
// the last ulp, therefore exp/acos/atan2/sin/cos are restated here with fixed
// polynomial kernels (Cephes single-precision coefficients) built only from
// + - * / sqrt and bit moves.  Both sides are compiled with -ffp-contract=off so
// no fused multiply-add is formed behind our back; division and sqrt are the
// correctly rounded forms on both (x86 SSE, and hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Only leaf arithmetic lives here (plus the counter-based RNG).  The algorithm
// itself is written twice, independently: oracle/pm_oracle.cpp (sequential CPU
// restatement of the reference) and csrc/pm_kernels.hip (GPU).
//
// Reference call sites these stand in for (all in /root/reference):
//   DENSE_EXP = EXP           libs/MVS/DepthMap.h:67-70 (used DepthMap.h:410, DepthMap.cpp:527,531,558)
//   ACOS                      libs/MVS/DepthMap.cpp:531, DepthMap.h:451
//   atan2/acos/sin/cos        libs/Common/Util.inl:754-766 (Normal2Dir / Dir2Normal)
//   Random::random<float>()   libs/Common/Random.h:113-115
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PM_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define PM_HD inline
#endif

#define PM_PI_F 3.14159265358979323846f
#define PM_HALF_PI_F 1.57079632679489661923f

PM_HD float pm_u2f(uint32_t u) {
	float f;
#if defined(__HIP_DEVICE_COMPILE__)
	f = __uint_as_float(u);
#else
	memcpy(&f, &u, 4);
#endif
	return f;
}
PM_HD uint32_t pm_f2u(float f) {
	uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
	u = __float_as_uint(f);
#else
	memcpy(&u, &f, 4);
#endif
	return u;
}

// correctly rounded sqrt on both sides.  NB: on gfx950/ROCm 7.2 __fsqrt_rn() lowers to a bare
// v_sqrt_f32 (1 ulp); the builtin below gets the correctly rounded expansion under
// -fhip-fp32-correctly-rounded-divide-sqrt (checked in tests/test_gpu_patchmatch.py).
PM_HD float pm_sqrtf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_sqrtf(x);
#else
	return sqrtf(x);
#endif
}
// (float)sqrt((double)a*a + (double)b*b): cv::norm(Point2f) evaluates in double (DepthMap.cpp:545)
PM_HD float pm_hypot_d(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
	return (float)__builtin_sqrt((double)a * (double)a + (double)b * (double)b);
#else
	return (float)sqrt((double)a * (double)a + (double)b * (double)b);
#endif
}
// x/z and y/z, each correctly rounded (== the IEEE quotient the oracle computes with '/'), sharing one
// reciprocal.  On the device this is the compiler's own division expansion (v_rcp, two reciprocal-refining
// FMAs, quotient, two residual-correcting FMA steps) minus v_div_scale / v_div_fixup, which only matter when
// an operand or the quotient sits near the ends of the exponent range; outside the guarded range we fall back
// to '/'.  13 instructions for the pair instead of 22.  Checked bit-for-bit against '/' in the gpu tests.
PM_HD void pm_div2(float x, float y, float z, float* qx, float* qy) {
#if defined(__HIP_DEVICE_COMPILE__)
	const float az = __builtin_fabsf(z);
	if (az > 9.094947e-13f && az < 1.0995116e12f && __builtin_fabsf(x) < 1.0e18f && __builtin_fabsf(y) < 1.0e18f) { // 2^-40 < |z| < 2^40
		float r = __builtin_amdgcn_rcpf(z);
		const float e = __builtin_fmaf(-z, r, 1.0f);
		r = __builtin_fmaf(e, r, r);
		float q = x * r;
		float t = __builtin_fmaf(-z, q, x); q = __builtin_fmaf(t, r, q);
		t = __builtin_fmaf(-z, q, x); q = __builtin_fmaf(t, r, q);
		*qx = q;
		q = y * r;
		t = __builtin_fmaf(-z, q, y); q = __builtin_fmaf(t, r, q);
		t = __builtin_fmaf(-z, q, y); q = __builtin_fmaf(t, r, q);
		*qy = q;
	} else { *qx = x / z; *qy = y / z; }
#else
	*qx = x / z; *qy = y / z;
#endif
}
// The fast branch of pm_div2 without its range test, for callers that establish the range themselves (the sweep kernel's tap rows check
// 2^-40 <= z <= 2^40 and |x|, |y| < 1e18 once per row, after the fact, and redo the row through pm_div2 when that fails).
PM_HD void pm_div2_inrange(float x, float y, float z, float* qx, float* qy) {
#if defined(__HIP_DEVICE_COMPILE__)
	// the two quotients as one packed chain (v_pk_mul_f32, 4 x v_pk_fma_f32): a packed FMA issues in about the time of a plain one on this part (4.8 vs 4.2
	// cycles per wave-instruction, tools/probes/valu_rate.hip), the same IEEE operations lane by lane
	typedef float pm_f2v __attribute__((ext_vector_type(2)));
	float r = __builtin_amdgcn_rcpf(z);
	const float e = __builtin_fmaf(-z, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	const pm_f2v xy = {x, y}, rr = {r, r}, nz = {-z, -z};
	pm_f2v q = xy * rr;
	pm_f2v t = __builtin_elementwise_fma(nz, q, xy); q = __builtin_elementwise_fma(t, rr, q);
	t = __builtin_elementwise_fma(nz, q, xy); q = __builtin_elementwise_fma(t, rr, q);
	*qx = q.x; *qy = q.y;
#else
	*qx = x / z; *qy = y / z;
#endif
}
// x - (float)(int)x for x >= 0 (TImage::sample's interpolation weight): the subtraction is exact, so v_fract_f32 returns the same bits in one
// instruction instead of a conversion back to float and a subtraction
PM_HD float pm_fract_pos(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_fractf(x);
#else
	return x - floorf(x);
#endif
}
// min / max of values known not to be NaN (v_min_f32 / v_max_f32, fused into v_min3 / v_max3 by the compiler)
PM_HD float pm_fminf(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_fminf(a, b);
#else
	return a < b ? a : b;
#endif
}
PM_HD float pm_fmaxf(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_fmaxf(a, b);
#else
	return a > b ? a : b;
#endif
}
PM_HD float pm_floorf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_floorf(x);
#else
	return floorf(x);
#endif
}
PM_HD float pm_fabsf(float x) { return pm_u2f(pm_f2u(x) & 0x7fffffffu); }
// bit pattern of a float as a signed integer: for non-negative floats the integer order is the float order, every negative float (and -0) is below every
// non-negative one and a NaN with a clear sign bit is above every number -- which is all a "is every value inside [lo, hi], lo >= 0" test needs, on the integer
// min / max instructions (v_min3_i32 / v_max3_i32: no NaN canonicalisation in front of them)
PM_HD int pm_f2i(float f) { return (int)pm_f2u(f); }
PM_HD float pm_minf(float a, float b) { return a < b ? a : b; } // MINF(a,b): libs/Common/Types.h
PM_HD float pm_maxf(float a, float b) { return a > b ? a : b; }
PM_HD float pm_clampf(float v, float lo, float hi) { return pm_minf(pm_maxf(v, lo), hi); } // CLAMP: Types.h:1196

// exp(x), |rel err| ~ 1 ulp.  Cephes expf scheme: n = round(x*log2e), r = x - n*ln2
// (two-part ln2), degree-5 polynomial, scale by 2^n through the exponent field.
PM_HD float pm_expf(float x) {
	if (x < -87.0f) return 0.0f;
	if (x > 88.0f) return pm_u2f(0x7f800000u);
	const float t = x * 1.44269504088896341f;
	const float nf = pm_floorf(t + 0.5f);
	float r = x - nf * 0.693359375f;
	r = r - nf * -2.12194440e-4f;
	const float z = r * r;
	float p = 1.9875691500E-4f;
	p = p * r + 1.3981999507E-3f;
	p = p * r + 8.3334519073E-3f;
	p = p * r + 4.1665795894E-2f;
	p = p * r + 1.6666665459E-1f;
	p = p * r + 5.0000001201E-1f;
	p = p * z + r;
	p = p + 1.0f;
	const int n = (int)nf;
	return p * pm_u2f((uint32_t)(n + 127) << 23);
}

// asin on |x| <= 0.5 (Cephes asinf kernel)
PM_HD float pm_asin_kernel(float x) {
	const float z = x * x;
	float p = 4.2163199048E-2f;
	p = p * z + 2.4181311049E-2f;
	p = p * z + 4.5470025998E-2f;
	p = p * z + 7.4953002686E-2f;
	p = p * z + 1.6666752422E-1f;
	p = p * z * x + x;
	return p;
}
// acos(x) for x in [-1,1] (caller clamps), ~2 ulp
PM_HD float pm_acosf(float x) {
	if (x > 0.5f) {
		const float s = pm_sqrtf(0.5f * (1.0f - x));
		return 2.0f * pm_asin_kernel(s);
	}
	if (x < -0.5f) {
		const float s = pm_sqrtf(0.5f * (1.0f + x));
		return PM_PI_F - 2.0f * pm_asin_kernel(s);
	}
	return PM_HALF_PI_F - pm_asin_kernel(x);
}

// atan(x), x >= 0 (Cephes atanf)
PM_HD float pm_atan_pos(float x) {
	float y;
	if (x > 2.414213562373095f) {
		y = PM_HALF_PI_F;
		x = -(1.0f / x);
	} else if (x > 0.4142135623730950f) {
		y = 0.7853981633974483f;
		x = (x - 1.0f) / (x + 1.0f);
	} else {
		y = 0.0f;
	}
	const float z = x * x;
	float p = 8.05374449538e-2f;
	p = p * z - 1.38776856032E-1f;
	p = p * z + 1.99777106478E-1f;
	p = p * z - 3.33329491539E-1f;
	p = p * z * x + x;
	return y + p;
}
// atan2(y,x) in (-pi, pi]
PM_HD float pm_atan2f(float y, float x) {
	if (x == 0.0f) {
		if (y > 0.0f) return PM_HALF_PI_F;
		if (y < 0.0f) return -PM_HALF_PI_F;
		return 0.0f;
	}
	const float q = y / x;
	const float a = pm_atan_pos(pm_fabsf(q));
	const float at = q < 0.0f ? -a : a; // atan(y/x)
	if (x > 0.0f) return at;
	return y >= 0.0f ? at + PM_PI_F : at - PM_PI_F;
}

// sin and cos for |x| < 8192 (here |x| < 8).  Cephes sinf/cosf: octant reduction
// with a three-part pi/4, then the sin or cos minimax kernel on [-pi/4, pi/4].
PM_HD void pm_sincosf(float xin, float* s, float* c) {
	float x = pm_fabsf(xin);
	int j = (int)(x * 1.27323954473516f); // x * 4/pi
	float y = (float)j;
	if (j & 1) { j += 1; y += 1.0f; }
	j &= 7;
	x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
	const float z = x * x;
	float ps = -1.9515295891E-4f;
	ps = ps * z + 8.3321608736E-3f;
	ps = ps * z - 1.6666654611E-1f;
	ps = ps * z * x + x;
	float pc = 2.443315711809948E-005f;
	pc = pc * z - 1.388731625493765E-003f;
	pc = pc * z + 4.166664568298827E-002f;
	pc = pc * z * z - 0.5f * z + 1.0f;
	// octant bookkeeping: j in {0,2,4,6}
	float sv, cv;
	if (j == 2 || j == 6) { sv = pc; cv = ps; } else { sv = ps; cv = pc; }
	// sign of sin: negative for j in {4,6}; sign of cos: negative for j in {2,4}
	if (j == 4 || j == 6) sv = -sv;
	if (j == 2) cv = -cv; // cos(x) with x = y*pi/4 + r, j==2: cos = -sin(r)
	if (j == 4) cv = -cv;
	if (xin < 0.0f) sv = -sv;
	*s = sv;
	*c = cv;
}

// ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) -------------------
// Replaces the reference's per-thread std::mt19937 stream (Random.h:102-137), which
// cannot be reproduced by a parallel schedule; see DESIGN.md "RNG".
struct PmPhilox4 { uint32_t v[4]; };
PM_HD uint32_t pm_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
PM_HD PmPhilox4 pm_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#ifdef PM_PROBE_CHEAP_DRAW   /* timing probe of the kernels only (see pm_kernels.hip); never defined for the oracle or the product */
	{ PmPhilox4 o; uint32_t h = (c0 * 0x9E3779B1u) ^ (c1 * 0x85EBCA77u) ^ (c2 * 0xC2B2AE3Du) ^ k0 ^ k1; o.v[0] = h * 0x27D4EB2Fu; o.v[1] = (h ^ (h >> 15)) * 0x165667B1u; o.v[2] = (h ^ (h >> 13)) * 0x9E3779B1u; o.v[3] = c3; return o; }
#endif
	for (int r = 0; r < 10; ++r) {
		const uint32_t hi0 = pm_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
		const uint32_t hi1 = pm_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
		const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	PmPhilox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
	return o;
}
// uniform in [0,1], exactly the reference mapping (float)u32/(float)0xFFFFFFFF
// (Random.h:113-115; (float)4294967295 == 2^32, so the quotient is an exact scaling)
PM_HD float pm_u32_to_unit(uint32_t u) { return (float)u * 2.3283064365386963e-10f; }
