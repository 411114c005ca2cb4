// Host launch-rate probe: how many microseconds of host time does one kernel launch cost when one thread feeds S streams round-robin, and what does the GPU
// make of N dependent launches per stream whose kernels last `spin` microseconds?  Decides whether a batch of few views (tens of thousands of ~30 us diagonal launches
// on two streams) is bound by the host's enqueue rate or by the chain of kernel latencies.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/launch_rate.hip -o /tmp/launch_rate && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Params { float f[14]; int spinCycles; int pad; };
__global__ void spin_kernel(const void* t, Params p, int a, int b, int c, int d, unsigned pass) {
	const unsigned long long t0 = wall_clock64();   // 100 MHz constant clock
	while ((long long)(wall_clock64() - t0) < (long long)p.spinCycles) { __builtin_amdgcn_s_sleep(2); }
	if (t == (const void*)1 && a + b + c + d + (int)pass == 12345) printf("never\n");
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	hipStream_t st[8];
	for (int i = 0; i < 8; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
	const int N = 20000;
	printf("%-8s %-8s %-8s %-10s %-14s %-14s\n", "streams", "threads", "spin_us", "blocks", "enqueue us/l", "total us/l");
	for (int threads = 1; threads <= 2; ++threads)
	for (int S : {1, 2, 4}) {
		if (threads == 2 && S == 1) continue;
		for (int spinUs : {0, 30, 100}) for (int blocks : {1, 2048}) {
			Params p = {}; p.spinCycles = spinUs * 100;
			for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, st[w % S], nullptr, p, 1, 2, 3, 4, 5u);
			hipDeviceSynchronize();
			const double t0 = now();
			double tEnq = 0;
			if (threads == 1) {
				for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, st[i % S], nullptr, p, 1, 2, 3, 4, 5u);
				tEnq = now() - t0;
			} else {
				std::vector<std::thread> th;
				for (int s = 0; s < S; ++s) th.emplace_back([&, s] { for (int i = s; i < N; i += S) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, st[s], nullptr, p, 1, 2, 3, 4, 5u); });
				for (auto& t : th) t.join();
				tEnq = now() - t0;
			}
			hipDeviceSynchronize();
			const double tAll = now() - t0;
			printf("%-8d %-8s %-8d %-10d %-14.2f %-14.2f\n", S, threads == 1 ? "1" : "S", spinUs, blocks, tEnq / N * 1e6, tAll / N * 1e6);
			fflush(stdout);
		}
	}
	return 0;
}
