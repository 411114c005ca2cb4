// hipemu -- a small CPU stand-in for <hip/hip_runtime.h>, TEST INFRASTRUCTURE ONLY (like oracle/: nothing in the product may include or load it).
//
// Purpose: compile the product's .hip sources (kernels AND host engine) unchanged for the host, so that every kernel, its launch geometry, its
// cross-lane traffic and its atomics execute on the CPU in the `-m "not gpu"` test-suite and are compared with the oracle there.  The GPU run then
// confirms instead of debugging.  This is not a performance model and not a memory model: threads of a workgroup run as cooperative fibers, one
// after the other, and switch only at the operations that exchange data between lanes.
//
// Execution model
//   * a launch runs its workgroups one after the other; the threads of a workgroup are fibers (own stacks, a 10-instruction context switch) on one OS thread;
//   * a fiber runs until it reaches a cross-lane operation (__shfl*, __ballot/__any/__all, DPP, readlane, wave barrier, __syncthreads), parks there
//     and the next fiber runs; when every live lane of a wavefront (64 consecutive threads) is parked, the lanes parked at the same operation and
//     call site exchange their values and are released -- lanes parked elsewhere (divergent code) simply do not take part, as with EXEC on the
//     hardware; reading a lane that does not take part returns 0 and is counted (hipemu::inactive_reads);
//   * __syncthreads releases when every live thread of the workgroup is parked at a barrier;
//   * atomics are plain read-modify-writes (fibers never run concurrently); "device memory" is host memory, filled with 0xCD.. on allocation so that
//     reads of never-written memory show up as garbage rather than as zeros.
//   * HIPEMU_ORDER=reverse|stride changes the order in which lanes and workgroups run: results must not depend on it (necessary for freedom of races).
// What it cannot show: races between workgroups or waves that run concurrently on the hardware, memory-ordering bugs, and anything that depends on
// the hardware's wave lock-step outside the operations listed above (code relying on that without a barrier fails here -- deliberately).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <execinfo.h>
#include <chrono>
#include <functional>
#include <map>
#include <tuple>
#include <vector>

// ---- language ----------------------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) (&(x))
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define warpSize 64

struct dim3 { unsigned x, y, z; constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct float2 { float x, y; }; struct float4 { float x, y, z, w; }; struct uint2 { unsigned x, y; }; struct short2 { short x, y; }; struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline short2 make_short2(short x, short y) { return short2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

// cooperative context switch (x86-64 System V): save the callee-saved registers on the current stack, swap stack pointers, restore, return
#if !defined(__x86_64__)
#error "hipemu's fiber switch is written for x86-64"
#endif
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(".text\n.weak hipemu_switch\n.type hipemu_switch,@function\nhipemu_switch:\n"
    "pushq %rbp\npushq %rbx\npushq %r12\npushq %r13\npushq %r14\npushq %r15\n"
    "movq %rsp, (%rdi)\nmovq %rsi, %rsp\n"
    "popq %r15\npopq %r14\npopq %r13\npopq %r12\npopq %rbx\npopq %rbp\nret\n.size hipemu_switch, .-hipemu_switch\n");

namespace hipemu {
enum Op { OP_NONE = 0, OP_SHFL, OP_BALLOT, OP_DPP, OP_READLANE, OP_WAVE_BARRIER, OP_SYNC };
struct Fiber {
	void* sp = nullptr;            // saved stack pointer while the fiber is not running
	dim3 tid;
	int lane = 0, wave = 0;
	int state = 0;                 // 0 runnable, 1 parked, 2 done
	int op = 0, site = 0;
	uint64_t val = 0, old = 0;     // operand (bit pattern) and DPP 'old'
	int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	uint64_t result = 0, active = 0;
};
inline Fiber* cur = nullptr;
inline void* sched_sp = nullptr;
inline dim3 g_blockIdx, g_blockDim, g_gridDim;
inline const std::function<void()>* g_body = nullptr;
inline uint64_t inactive_reads = 0, launches = 0, fibers_run = 0, collectives = 0;
inline const char* g_kernel = "?";

[[noreturn]] inline void die(const char* what, int site = 0) {
	fprintf(stderr, "hipemu: %s (kernel %s, site line %d, block %u,%u,%u)\n", what, g_kernel, site, g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
	abort();
}

inline void park(int op, int site) {
	Fiber* f = cur;
	f->op = op; f->site = site; f->state = 1;
	hipemu_switch(&f->sp, sched_sp);
}
inline void trampoline() {
	(*g_body)();
	cur->state = 2;
	hipemu_switch(&cur->sp, sched_sp);
	abort();
}
inline void prepare(Fiber& f, char* stack, size_t size) {       // first switch into the fiber "returns" into trampoline()
	uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
	void** p = (void**)top;
	*--p = nullptr;                        // fake return address of trampoline (it never returns); keeps rsp = 8 mod 16 at its entry
	*--p = (void*)&trampoline;
	for (int i = 0; i < 6; ++i) *--p = nullptr;   // rbp rbx r12 r13 r14 r15
	f.sp = (void*)p;
}

// source lane of a DPP control for destination lane `l`, or -1 if there is none (row = 16 lanes)
inline int dpp_source(int ctrl, int l) {
	const int row = l >> 4, r = l & 15;
	if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = r + (ctrl & 15); return s > 15 ? -1 : (row << 4) + s; }     // row_shl:n  (lane reads lane+n)
	if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = r - (ctrl & 15); return s < 0 ? -1 : (row << 4) + s; }      // row_shr:n  (lane reads lane-n)
	if (ctrl >= 0x121 && ctrl <= 0x12f) return (row << 4) + ((r - (ctrl & 15)) & 15);                                  // row_ror:n
	if (ctrl == 0x130) return l < 63 ? l + 1 : -1;                                                                     // wave_shl:1 (lane reads lane+1 across rows)
	if (ctrl == 0x138) return l > 0 ? l - 1 : -1;                                                                      // wave_shr:1 (lane reads lane-1 across rows)
	if (ctrl == 0x140) return (row << 4) + (15 - r);                                                                   // row_mirror
	if (ctrl == 0x141) return (row << 4) + (r < 8 ? 7 - r : 23 - r);                                                    // row_half_mirror
	if (ctrl == 0x142) return row == 0 ? -1 : ((row - 1) << 4) + 15;                                                   // row_bcast:15
	if (ctrl == 0x143) return row < 2 ? -1 : 31;                                                                       // row_bcast:31
	if (ctrl <= 0xff) { const int q = l & ~3; return q + ((ctrl >> (2 * (l & 3))) & 3); }                              // quad_perm
	die("unsupported DPP control");
}

inline void resolve_group(std::vector<Fiber*>& g) {           // lanes of one wave parked at the same (op, site)
	Fiber* bylane[64] = {};
	uint64_t active = 0;
	for (Fiber* f : g) { bylane[f->lane] = f; active |= 1ull << f->lane; }
	const int op = g[0]->op;
	uint64_t ballot = 0;
	if (op == OP_BALLOT) for (Fiber* f : g) if (f->val) ballot |= 1ull << f->lane;
	for (Fiber* f : g) {
		f->active = active;
		switch (op) {
		case OP_SHFL: {
			const int self = f->lane, width = f->a2; int index;
			switch (f->a0) {
			case 0: index = (f->a1 & (width - 1)) + (self & ~(width - 1)); break;                                       // __shfl
			case 1: index = self ^ f->a1; index = index >= ((self + width) & ~(width - 1)) ? self : index; break;      // __shfl_xor
			case 2: index = self + f->a1; index = (int)((self & (width - 1)) + f->a1) >= width ? self : index; break;  // __shfl_down
			default: index = self - f->a1; index = (index < (self & ~(width - 1))) ? self : index; break;              // __shfl_up
			}
			index &= 63;
			if (bylane[index]) f->result = bylane[index]->val; else { f->result = 0; ++inactive_reads; }
			break; }
		case OP_BALLOT: f->result = ballot; break;
		case OP_DPP: {
			const int src = dpp_source(f->a0, f->lane);
			const bool rowOn = (f->a1 >> (f->lane >> 4)) & 1, bankOn = (f->a2 >> ((f->lane & 15) >> 2)) & 1;
			if (!rowOn || !bankOn) f->result = f->old;
			else if (src < 0 || !bylane[src]) f->result = f->a3 ? 0 : f->old;
			else f->result = bylane[src]->val;
			break; }
		case OP_READLANE:
			if (bylane[f->a0 & 63]) f->result = bylane[f->a0 & 63]->val; else { f->result = 0; ++inactive_reads; }
			break;
		default: break;
		}
	}
	for (Fiber* f : g) f->state = 0;
	++collectives;
}

// Execution order of the lanes of a workgroup and of the workgroups of a grid: HIPEMU_ORDER = forward (default) | reverse | stride (a fixed
// permutation i -> i * 7919 mod n, applied when n is not a multiple of 7919).  A kernel that is free of races gives the same result under every order.
inline int order_mode() {
	static int m = -1;
	if (m < 0) { const char* e = getenv("HIPEMU_ORDER"); m = !e ? 0 : (!strcmp(e, "reverse") ? 1 : (!strcmp(e, "stride") ? 2 : 0)); }
	return m;
}
inline uint64_t permute(uint64_t i, uint64_t n) {
	switch (order_mode()) {
	case 1: return n - 1 - i;
	case 2: return n % 7919 ? (i * 7919) % n : i;
	default: return i;
	}
}

struct StackPool {
	std::vector<char*> stacks; size_t size = 256 * 1024;
	char* get(size_t i) { while (stacks.size() <= i) stacks.push_back((char*)malloc(size)); return stacks[i]; }
	~StackPool() { for (char* s : stacks) free(s); }
};
inline StackPool g_stacks;

inline void run_block(const std::function<void()>& body) {
	const unsigned nT = g_blockDim.x * g_blockDim.y * g_blockDim.z;
	static std::vector<Fiber> fibers;
	fibers.assign(nT, Fiber());
	g_body = &body;
	for (unsigned t = 0; t < nT; ++t) {
		Fiber& f = fibers[t];
		f.tid = dim3(t % g_blockDim.x, (t / g_blockDim.x) % g_blockDim.y, t / (g_blockDim.x * g_blockDim.y));
		f.lane = (int)(t & 63); f.wave = (int)(t >> 6);
		prepare(f, g_stacks.get(t), g_stacks.size);
	}
	fibers_run += nT;
	const unsigned nW = (nT + 63) / 64;
	std::vector<Fiber*> group;
	for (;;) {
		bool progressed = false; unsigned alive = 0;
		for (unsigned i = 0; i < nT; ++i) {
			Fiber& f = fibers[permute(i, nT)];
			if (f.state != 0) continue;
			cur = &f; hipemu_switch(&sched_sp, f.sp); cur = nullptr;
			progressed = true;
		}
		unsigned atSync = 0;
		for (unsigned t = 0; t < nT; ++t) { if (fibers[t].state != 2) ++alive; if (fibers[t].state == 1 && fibers[t].op == OP_SYNC) ++atSync; }
		if (!alive) return;
		// every live fiber is parked now: resolve the wave-level operations, one (op, site) group at a time
		for (unsigned w = 0; w < nW; ++w) {
			const unsigned t0 = w * 64, t1 = t0 + 64 < nT ? t0 + 64 : nT;
			for (;;) {
				group.clear();
				for (unsigned t = t0; t < t1; ++t) {
					Fiber& f = fibers[t];
					if (f.state != 1 || f.op == OP_SYNC) continue;
					if (group.empty() || (group[0]->op == f.op && group[0]->site == f.site)) group.push_back(&f);
				}
				if (group.empty()) break;
				resolve_group(group); progressed = true;
			}
		}
		if (atSync == alive) { for (unsigned t = 0; t < nT; ++t) if (fibers[t].state == 1) fibers[t].state = 0; progressed = true; ++collectives; }
		if (!progressed) die("deadlock: no fiber can make progress");
	}
}

template <class F> inline void launch(const char* name, dim3 grid, dim3 block, F&& f) {
	g_kernel = name; g_gridDim = grid; g_blockDim = block; ++launches;
	if (!grid.x || !grid.y || !grid.z || !block.x || !block.y || !block.z) die("empty launch");
	if ((uint64_t)block.x * block.y * block.z > 1024) die("block larger than 1024 threads");
	const std::function<void()> body(f);
	const uint64_t nB = (uint64_t)grid.x * grid.y * grid.z;
	for (uint64_t i = 0; i < nB; ++i) {
		const uint64_t b = permute(i, nB);
		g_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y)));
		run_block(body);
	}
}

template <class T> inline uint64_t bits(T v) { static_assert(sizeof(T) <= 8, ""); uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> inline T unbits(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
template <class T> inline T shfl(int kind, T v, int arg, int width, int site) {
	Fiber* f = cur; f->val = bits(v); f->a0 = kind; f->a1 = arg; f->a2 = width;
	park(OP_SHFL, site); return unbits<T>(f->result);
}
inline uint64_t ballot(int pred, int site) { Fiber* f = cur; f->val = pred ? 1 : 0; park(OP_BALLOT, site); return f->result; }
inline int all_(int pred, int site) { Fiber* f = cur; f->val = pred ? 1 : 0; park(OP_BALLOT, site); return f->result == f->active; }
template <class T> inline T dpp(T old, T src, int ctrl, int rowMask, int bankMask, bool boundCtrl, int site) {
	Fiber* f = cur; f->val = bits(src); f->old = bits(old); f->a0 = ctrl; f->a1 = rowMask; f->a2 = bankMask; f->a3 = boundCtrl;
	park(OP_DPP, site); return unbits<T>(f->result);
}
template <class T> inline T readlane(T v, int lane, int site) { Fiber* f = cur; f->val = bits(v); f->a0 = lane; park(OP_READLANE, site); return unbits<T>(f->result); }
} // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)

// ---- cross-lane operations -----------------------------------------------------------------------------------------------------------------
#define HIPEMU_PICK(_1, _2, _3, N, ...) N
#define __shfl(...) HIPEMU_PICK(__VA_ARGS__, hipemu_shfl3, hipemu_shfl2)(0, __LINE__, __VA_ARGS__)
#define __shfl_xor(...) HIPEMU_PICK(__VA_ARGS__, hipemu_shfl3, hipemu_shfl2)(1, __LINE__, __VA_ARGS__)
#define __shfl_down(...) HIPEMU_PICK(__VA_ARGS__, hipemu_shfl3, hipemu_shfl2)(2, __LINE__, __VA_ARGS__)
#define __shfl_up(...) HIPEMU_PICK(__VA_ARGS__, hipemu_shfl3, hipemu_shfl2)(3, __LINE__, __VA_ARGS__)
#define hipemu_shfl3(kind, site, v, a, w) hipemu::shfl((kind), (v), (int)(a), (int)(w), (site))
#define hipemu_shfl2(kind, site, v, a) hipemu::shfl((kind), (v), (int)(a), 64, (site))
#define __ballot(p) hipemu::ballot((p), __LINE__)
#define __any(p) (hipemu::ballot((p), __LINE__) != 0)
#define __all(p) hipemu::all_((p), __LINE__)
#define __syncthreads() hipemu::park(hipemu::OP_SYNC, __LINE__)
#define __builtin_amdgcn_wave_barrier() hipemu::park(hipemu::OP_WAVE_BARRIER, __LINE__)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu::dpp((old), (src), (ctrl), (rm), (bm), (bc), __LINE__)
#define __builtin_amdgcn_readlane(v, l) hipemu::readlane((v), (l), __LINE__)
#define WAVE_LOCKSTEP_POINT() hipemu::park(hipemu::OP_WAVE_BARRIER, __LINE__)   // the product's marker for reliance on wave lock-step (see sgm_kernels.hip)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }

// ---- atomics (fibers never overlap: plain read-modify-write, returning the old value) ---------------------------------------------------
template <class T, class U> static inline T atomicAdd(T* p, U v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { const T o = *p; *p = (T)v; return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { const T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { const T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U, class V> static inline T atomicCAS(T* p, U c, V v) { const T o = *p; if (o == (T)c) *p = (T)v; return o; }

// HIP's device overloads of min / max
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }

// ---- runtime API (synchronous, single "device") ------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipemu_stream* hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point t; };
typedef hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// "device" allocations carry a 64-byte canary on both sides; every launch is followed by a check of all of them, so an out-of-bounds write is
// reported with the kernel that did it (on the hardware it would be a memory fault or silent corruption of a neighbouring buffer)
namespace hipemu {
enum { GUARD = 64 };
inline std::map<void*, size_t>& allocs() { static std::map<void*, size_t> m; return m; }
inline void check_guards(const char* when) {
	for (auto& a : allocs()) {
		const unsigned char* b = (const unsigned char*)a.first;
		for (int i = 0; i < GUARD; ++i) if (b[-1 - i] != 0xA5 || b[a.second + i] != 0xA5) {
			fprintf(stderr, "hipemu: write outside a device allocation of %zu bytes (%s the block, offset %d), detected %s kernel %s\n", a.second,
			        b[-1 - i] != 0xA5 ? "before" : "after", i, when, g_kernel);
			abort();
		}
	}
}
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) {
	char* q = (char*)malloc(n + 2 * hipemu::GUARD);
	if (!q) return hipErrorOutOfMemory;
	memset(q, 0xA5, hipemu::GUARD); memset(q + hipemu::GUARD, 0xCD, n); memset(q + hipemu::GUARD + n, 0xA5, hipemu::GUARD);
	hipemu::allocs()[q + hipemu::GUARD] = n;
	*p = (T*)(q + hipemu::GUARD); return hipSuccess;
}
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { *p = (T*)malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) {
	if (!p) return hipSuccess;
	auto it = hipemu::allocs().find(p);
	if (it == hipemu::allocs().end()) { fprintf(stderr, "hipemu: hipFree of a pointer that hipMalloc did not return (double free?)\n"); void* bt[32]; backtrace_symbols_fd(bt, backtrace(bt, 32), 2); abort(); }
	hipemu::check_guards("at hipFree after");
	hipemu::allocs().erase(it);
	free((char*)p - hipemu::GUARD); return hipSuccess;
}
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) { memcpy(d, (const char*)sym + off, n); return hipSuccess; }
static inline hipError_t hipMemcpyToSymbol(void* sym, const void* s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) { memcpy((char*)sym + off, s, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

// Arguments are converted to the kernel's parameter types at the call site, as a real launch does (and not inside the deferred body).
namespace hipemu {
template <class... P, class... A> inline void launch_args(const char* name, dim3 grid, dim3 block, void (*k)(P...), A&&... a) {
	std::tuple<std::decay_t<P>...> args(std::forward<A>(a)...);
	launch(name, grid, block, [&]() { std::apply(k, args); });
	check_guards("after");
}
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch_args(#kernel, dim3(grid), dim3(block), kernel, ##__VA_ARGS__)

// counters for the tests (ctypes): launches, fibers run, cross-lane exchanges resolved, reads of lanes that did not take part in an exchange
extern "C" __attribute__((weak, visibility("default"))) void hipemu_counters(uint64_t out[4]) {
	out[0] = hipemu::launches; out[1] = hipemu::fibers_run; out[2] = hipemu::collectives; out[3] = hipemu::inactive_reads;
}
