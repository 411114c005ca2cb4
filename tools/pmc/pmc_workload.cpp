// Counter-collection workload with nothing but the C ABI in the profiled process (round-2's Python workloads hung or crashed rocprofv3 --pmc):
//   pmc_workload <scene.bin> [geo_iters] [levels]
// scene.bin: i32 n, w, h, nsrc | per view: f32 gray[w*h], f64 K[9] R[9] C[3], f32 dMin dMax, i32 neighbors[nsrc]   (tools/pmc/make_scene.py)
// Uploads the scene, runs one photometric schedule of all views in one batch (+ geo_iters geometric rounds) through the HBM-resident scene
// interface and prints the sweep statistics bench.py reports, so per-launch counters can be set against the algorithmic bytes per launch.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pmhip.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, pmhip_last_error(e)); return 10; } } while (0)

int main(int argc, char** argv) {
	if (argc < 2) return 2;
	FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
	const int geo = argc > 2 ? atoi(argv[2]) : 0;
	const int levels = argc > 3 ? atoi(argv[3]) : 2;
	int32_t hd[4]; if (fread(hd, 4, 4, f) != 4) return 4;
	const int n = hd[0], w = hd[1], h = hd[2], ns = hd[3];
	const size_t P = (size_t)w * h;
	pmhip_engine* e = nullptr;
	if (pmhip_create(0, &e)) { fprintf(stderr, "no device\n"); return 5; }
	CK(pmhip_init(e, 1));
	CK(pmhip_scene_create(e, n, w, h, levels));
	std::vector<float> gray(P);
	for (int i = 0; i < n; ++i) {
		double cam[21]; float rng[2]; std::vector<int32_t> nb((size_t)ns);
		if (fread(gray.data(), 4, P, f) != P || fread(cam, 8, 21, f) != 21 || fread(rng, 4, 2, f) != 2 || fread(nb.data(), 4, (size_t)ns, f) != (size_t)ns) return 4;
		CK(pmhip_scene_set_view(e, i, gray.data(), 0, cam, cam + 9, cam + 18, rng[0], rng[1], nb.data(), ns));
	}
	fclose(f);
	PMHipParams p; pmhip_default_params(&p); p.seed = 1; p.nSubResolutionLevels = (unsigned)levels; p.nEstimationGeometricIters = (unsigned)geo;
	std::vector<int32_t> ids((size_t)n); for (int i = 0; i < n; ++i) ids[(size_t)i] = i;
	CK(pmhip_stats_reset(e, 1));
	const auto t0 = std::chrono::steady_clock::now();
	CK(pmhip_scene_estimate(e, ids.data(), n, &p, -1, 0));
	for (int g = 0; g < geo; ++g) { CK(pmhip_scene_commit_round(e)); CK(pmhip_scene_estimate(e, ids.data(), n, &p, g, 0)); }
	CK(pmhip_sync(e));
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	PMHipKernelStats st; memset(&st, 0, sizeof(st)); CK(pmhip_stats_get(e, &st));
	printf("{\"views\": %d, \"w\": %d, \"h\": %d, \"geo_iters\": %d, \"seconds\": %.3f, \"sweep_launches\": %llu, \"avg_launch_us\": %.2f, \"algorithmic_bytes_per_launch\": %.1f, \"mpix_s\": %.3f, \"sweep_wall_s\": %.4f}\n",
		n, w, h, geo, dt, (unsigned long long)st.sweepLaunches, 1e3 * st.sweepMs / (double)(st.sweepLaunches ? st.sweepLaunches : 1),
		st.sweepBytes / (double)(st.sweepLaunches ? st.sweepLaunches : 1), (double)n * P / dt * 1e-6, st.sweepWallMs * 1e-3);
	pmhip_destroy(e);
	return 0;
}
