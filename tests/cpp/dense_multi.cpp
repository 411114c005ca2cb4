// The multi-device C++ host (include/DenseDepthMapsHIPMulti.hpp) against the single-engine driver on the same scene: maps of every view and the
// fused cloud must be the same bits whichever engine estimated a view.  Runs with several engines on ONE device through LocalCopyCollective (the
// single-GPU box; under the CPU emulator in the not-gpu suite); the RCCL policy is the same code path with ncclBroadcast in place of the copies.
// Built with -DPMHIP_WITH_RCCL and run as `dense_multi <scene.bin> 1 <seed> rccl` the multi-device object uses RcclCollective over device 0 alone: one rank is all a
// single-GPU box can give RCCL (two ranks may not share a device), and it runs every RCCL call of the N-device path (ncclCommInitAll, grouped ncclBroadcast).
// Usage: dense_multi <scene.bin> <engines> [seed] [serial]        scene.bin as tests/cpp/dense_driver.cpp; serial: no host threads (the CPU emulator is single-threaded)
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>
#include "DenseDepthMapsHIPMulti.hpp"

int main(int argc, char** argv) {
	if (argc < 3) return 2;
	FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
	const int nEng = atoi(argv[2]);
	int32_t hd[4]; if (fread(hd, 4, 4, f) != 4) return 4;
	// a negative view count: every view record starts with its own image size (a scene whose views differ in size, DepthMapsData::InitViews, SceneDensify.cpp:306-459)
	const bool mixed = hd[0] < 0;
	const int n = mixed ? -hd[0] : hd[0], w = hd[1], h = hd[2], ns = hd[3];
	std::vector<std::vector<float>> gray((size_t)n);
	std::vector<std::vector<unsigned char>> bgr((size_t)n);
	std::vector<MVS::DenseDepthMapsHIP::View> views((size_t)n);
	size_t Pmax = (size_t)w * h;
	for (int i = 0; i < n; ++i) {
		auto& v = views[(size_t)i];
		int32_t sz[2] = {w, h};
		if (mixed && fread(sz, 4, 2, f) != 2) return 4;
		const size_t P = (size_t)sz[0] * sz[1]; Pmax = std::max(Pmax, P);
		if (mixed) { v.w = sz[0]; v.h = sz[1]; }
		gray[(size_t)i].resize(P); bgr[(size_t)i].resize(P * 3);
		double cam[21]; float rng[2]; std::vector<int32_t> nb((size_t)ns);
		if (fread(gray[(size_t)i].data(), 4, P, f) != P || fread(bgr[(size_t)i].data(), 1, P * 3, f) != P * 3 || fread(cam, 8, 21, f) != 21 || fread(rng, 4, 2, f) != 2 ||
		    fread(nb.data(), 4, (size_t)ns, f) != (size_t)ns) return 4;
		v.gray = gray[(size_t)i].data(); v.bgr = bgr[(size_t)i].data();
		memcpy(v.K, cam, 72); memcpy(v.R, cam + 9, 72); memcpy(v.C, cam + 18, 24);
		v.dMin = rng[0]; v.dMax = rng[1]; v.neighbors = nb; v.ID = (uint32_t)i;
	}
	fclose(f);
	MVS::DenseDepthMapsHIP::Options opt;
	opt.seed = argc > 3 ? (uint32_t)atoi(argv[3]) : 31u; opt.nSpeckleSize = 30;
	try {
		MVS::DenseDepthMapsHIP one(0);
		if (!one.IsValid()) { fprintf(stderr, "no device\n"); return 5; }
		one.LoadScene(views, w, h, opt); one.ComputeDepthMaps();
		MVS::DenseDepthMapsHIP::PointCloud pc1; one.FuseDepthMaps(pc1);
#ifdef PMHIP_WITH_RCCL
		const bool rccl = argc > 4 && !strcmp(argv[4], "rccl");
		if (rccl && nEng != 1) { fprintf(stderr, "rccl mode: one engine per device, this box has one\n"); return 2; }
		MVS::DenseDepthMapsHIPMultiT<MVS::RcclCollective> multi(std::vector<int>((size_t)nEng, 0), true);
#else
		MVS::DenseDepthMapsHIPMultiT<MVS::LocalCopyCollective> multi(std::vector<int>((size_t)nEng, 0), argc <= 4);
#endif
		if (!multi.IsValid()) { fprintf(stderr, "no device\n"); return 5; }
		multi.LoadScene(views, w, h, opt); multi.ComputeDepthMaps();
		MVS::DenseDepthMapsHIP::PointCloud pc2; multi.FuseDepthMaps(pc2);
		std::vector<float> d1(Pmax), n1(Pmax * 3), c1(Pmax), d2(Pmax), n2(Pmax * 3), c2(Pmax);
		for (int i = 0; i < n; ++i) {
			if (multi.ViewWidth(i) != one.ViewWidth(i) || multi.ViewHeight(i) != one.ViewHeight(i)) { fprintf(stderr, "view %d: sizes differ\n", i); return 9; }
			const size_t P = (size_t)one.ViewWidth(i) * one.ViewHeight(i);
			one.GetMaps(i, d1.data(), n1.data(), c1.data()); multi.GetMaps(i, d2.data(), n2.data(), c2.data());
			if (memcmp(d1.data(), d2.data(), P * 4) || memcmp(n1.data(), n2.data(), P * 12) || memcmp(c1.data(), c2.data(), P * 4)) { fprintf(stderr, "view %d differs\n", i); return 10; }
		}
		if (pc1.size() != pc2.size() || pc1.points != pc2.points || pc1.viewStart != pc2.viewStart || pc1.views != pc2.views || pc1.weights != pc2.weights || pc1.colors != pc2.colors || pc1.normals != pc2.normals) {
			fprintf(stderr, "fused clouds differ: %zu vs %zu points\n", pc1.size(), pc2.size()); return 11; }
		size_t valid = 0; one.GetMaps(0, d1.data(), nullptr, nullptr); for (size_t q = 0; q < (size_t)one.ViewWidth(0) * one.ViewHeight(0); ++q) valid += d1[q] > 0;
		printf("%d engines == 1 engine: %d views%s, %zu fused points, view 0 valid %zu, %zu bytes exchanged\n", nEng, n, mixed ? " of different sizes" : "", pc1.size(), valid, multi.ExchangedBytes());
	} catch (const std::exception& ex) { fprintf(stderr, "%s\n", ex.what()); return 7; }
	return 0;
}
