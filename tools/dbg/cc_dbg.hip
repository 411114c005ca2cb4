// debug harness: run the connected-component kernels on a depth map from stdin-like file and dump parent/size
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../openmvs_amd/csrc/pm_math.h"
#include "../../openmvs_amd/csrc/pm_filter.hip"
int main(int argc, char** argv) {
	int w = atoi(argv[1]), h = atoi(argv[2]); float th = atof(argv[3]);
	int n = w * h; std::vector<float> d(n);
	FILE* f = fopen(argv[4], "rb"); fread(d.data(), 4, n, f); fclose(f);
	float* dd; int *parent, *size;
	hipMalloc(&dd, n * 4); hipMalloc(&parent, n * 4); hipMalloc(&size, n * 4);
	hipMemcpy(dd, d.data(), n * 4, hipMemcpyHostToDevice);
	int gx = (n + 255) / 256;
	hipLaunchKernelGGL(pmf_cc_init_kernel, dim3(gx), dim3(256), 0, 0, parent, size, n);
	hipLaunchKernelGGL(pmf_cc_hook_kernel, dim3(gx), dim3(256), 0, 0, dd, parent, w, h, th);
	hipLaunchKernelGGL(pmf_cc_flatten_kernel, dim3(gx), dim3(256), 0, 0, parent, size, n);
	hipDeviceSynchronize();
	std::vector<int> hp(n), hs(n);
	hipMemcpy(hp.data(), parent, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), size, n * 4, hipMemcpyDeviceToHost);
	f = fopen(argv[5], "wb"); fwrite(hp.data(), 4, n, f); fwrite(hs.data(), 4, n, f); fclose(f);
	printf("err %s\n", hipGetErrorString(hipGetLastError()));
	return 0;
}
