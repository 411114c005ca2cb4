// A SLICE of the benchmark for counter passes that finish (rocprofv3 --pmc over the whole 100-view schedule -- 43 126 sweep dispatches -- hung in 6 of 7 passes, profiles/r05_call6_pmc):
// the scene of the benchmark resident in HBM, the maps of the benchmark after its photometric pass loaded from a file, and then ONE level-0 sweep -- about 3 000 dispatches.
//   pmc_slice <scene.bin> prep  <maps.bin>                 full photometric pass (3 levels x 3 sweeps), maps of every view written to maps.bin      (not profiled)
//   pmc_slice <scene.bin> photo <maps.bin> [groups] [rep] [wide]  maps loaded; photometric pass with no sub-level and one sweep: init pass + one LT2RB sweep of the photometric kernels
//   pmc_slice <scene.bin> geo   <maps.bin> [groups] [rep]  maps loaded and committed as the previous round; geometric round 0: init pass + one sweep of the geometric kernels
// scene.bin as tools/pmc/make_scene.py writes it.  groups: view groups (streams) of the batch, default 1 (counter passes serialise dispatches anyway).
// rep (probe builds of the library only, -DPM_PROBES): every diagonal launch is issued `rep` times back to back -- the maps are garbage afterwards; what is measured is
// what the second launch of the SAME diagonal fetches (is L2 cold at every launch?).
// wide: PMHipTuning::wideMaxViews (-1 = no speculative kernels: pm_sweep2_kernel for every launch; 0 = the engine's default).
// Prints one JSON line with the sweep statistics of the slice.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>
#include "pmhip.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, pmhip_last_error(e)); return 10; } } while (0)

int main(int argc, char** argv) {
	if (argc < 4) { fprintf(stderr, "usage: pmc_slice scene.bin prep|photo|geo maps.bin [groups] [rep]\n"); return 2; }
	const std::string mode = argv[2];
	const int groups = argc > 4 ? atoi(argv[4]) : 1, rep = argc > 5 ? atoi(argv[5]) : 1, wide = argc > 6 ? atoi(argv[6]) : 0;
	FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
	int32_t hd[4]; if (fread(hd, 4, 4, f) != 4) return 4;
	const int n = hd[0], w = hd[1], h = hd[2], ns = hd[3];
	const size_t P = (size_t)w * h;
	pmhip_engine* e = nullptr;
	if (pmhip_create(0, &e)) { fprintf(stderr, "no device\n"); return 5; }
	CK(pmhip_init(e, 1));
	CK(pmhip_scene_create(e, n, w, h, 2));
	std::vector<float> gray(P);
	for (int i = 0; i < n; ++i) {
		double cam[21]; float rng[2]; std::vector<int32_t> nb((size_t)ns);
		if (fread(gray.data(), 4, P, f) != P || fread(cam, 8, 21, f) != 21 || fread(rng, 4, 2, f) != 2 || fread(nb.data(), 4, (size_t)ns, f) != (size_t)ns) return 4;
		CK(pmhip_scene_set_view(e, i, gray.data(), 0, cam, cam + 9, cam + 18, rng[0], rng[1], nb.data(), ns));
	}
	fclose(f);
	PMHipTuning tn; memset(&tn, 0, sizeof(tn)); tn.viewGroups = groups; tn.wideMaxViews = wide; CK(pmhip_set_tuning(e, &tn));
	if (rep > 1) {
		typedef int (*probe_fn)(int, int);
		probe_fn ps = (probe_fn)dlsym(RTLD_DEFAULT, "pmhip_probe_set");
		if (!ps) { fprintf(stderr, "rep > 1 needs a library built with -DPM_PROBES\n"); return 6; }
		ps(0, rep);
	}
	PMHipParams p; pmhip_default_params(&p); p.seed = 1; p.nSubResolutionLevels = 2; p.nEstimationGeometricIters = 2;
	std::vector<int32_t> ids((size_t)n); for (int i = 0; i < n; ++i) ids[(size_t)i] = i;
	std::vector<float> depth(P), normal(P * 3), conf(P);
	if (mode == "prep") {
		const auto t0 = std::chrono::steady_clock::now();
		CK(pmhip_scene_estimate(e, ids.data(), n, &p, -1, 1));
		const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		FILE* o = fopen(argv[3], "wb"); if (!o) return 7;
		for (int i = 0; i < n; ++i) {
			CK(pmhip_scene_get_maps(e, i, depth.data(), normal.data(), conf.data()));
			if (fwrite(depth.data(), 4, P, o) != P || fwrite(normal.data(), 4, P * 3, o) != P * 3) return 7;
		}
		fclose(o);
		printf("{\"mode\": \"prep\", \"views\": %d, \"seconds\": %.3f}\n", n, dt);
		pmhip_destroy(e);
		return 0;
	}
	FILE* m = fopen(argv[3], "rb"); if (!m) return 8;
	for (int i = 0; i < n; ++i) {
		if (fread(depth.data(), 4, P, m) != P || fread(normal.data(), 4, P * 3, m) != P * 3) return 8;
		CK(pmhip_scene_set_maps(e, i, depth.data(), normal.data()));
	}
	fclose(m);
	CK(pmhip_sync(e));
	CK(pmhip_stats_reset(e, 1));
	const auto t0 = std::chrono::steady_clock::now();
	if (mode == "photo") {
		p.nSubResolutionLevels = 0; p.nEstimationIters = 1;
		CK(pmhip_scene_estimate(e, ids.data(), n, &p, -1, 1));
	} else {
		CK(pmhip_scene_commit_round(e));
		CK(pmhip_scene_estimate(e, ids.data(), n, &p, 0, 1));
	}
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	PMHipKernelStats st; memset(&st, 0, sizeof(st)); CK(pmhip_stats_get(e, &st));
	printf("{\"mode\": \"%s\", \"views\": %d, \"groups\": %d, \"rep\": %d, \"seconds\": %.3f, \"sweep_launches\": %llu, \"avg_launch_us\": %.2f, \"sweep_stream_ms\": %.1f, \"init_ms\": %.1f, "
	       "\"algorithmic_bytes_per_launch\": %.1f, \"sweep_wall_ms\": %.1f}\n",
		mode.c_str(), n, groups, rep, dt, (unsigned long long)st.sweepLaunches, 1e3 * st.sweepMs / (double)(st.sweepLaunches ? st.sweepLaunches : 1), st.sweepMs, st.initMs,
		st.sweepBytes / (double)(st.sweepLaunches ? st.sweepLaunches : 1), st.sweepWallMs);
	pmhip_destroy(e);
	return 0;
}
