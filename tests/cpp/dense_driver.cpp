// Drives include/DenseDepthMapsHIP.hpp the way a patched Scene::ComputeDepthMaps / DenseReconstruction would (SceneDensify.cpp:1754-1982,
// :1655-1750): whole scene in, all depth maps + fused cloud out, no Python anywhere.
// Usage: dense_driver <scene.bin> <out.bin> <dmap dir> [seed] [speckle]
//   scene.bin: i32 n, w, h, nsrc | per view: f32 gray[w*h], u8 bgr[w*h*3], f64 K[9] R[9] C[3], f32 dMin dMax, i32 neighbors[nsrc]
//   out.bin:   per view f32 depth[w*h] normal[w*h*3] conf[w*h] | u64 nPoints, nViews | f32 points[3*nPoints] | u32 viewStart[nPoints+1] | u32 views[nViews]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "DenseDepthMapsHIP.hpp"
#include "dmapio.h"

int main(int argc, char** argv) {
	if (argc < 4) return 2;
	FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
	int32_t hd[4]; if (fread(hd, 4, 4, f) != 4) return 4;
	const int n = hd[0], w = hd[1], h = hd[2], ns = hd[3];
	const size_t P = (size_t)w * h;
	std::vector<std::vector<float>> gray((size_t)n, std::vector<float>(P));
	std::vector<std::vector<unsigned char>> bgr((size_t)n, std::vector<unsigned char>(P * 3));
	std::vector<MVS::DenseDepthMapsHIP::View> views((size_t)n);
	for (int i = 0; i < n; ++i) {
		auto& v = views[(size_t)i];
		double cam[21]; float rng[2]; std::vector<int32_t> nb((size_t)ns);
		if (fread(gray[(size_t)i].data(), 4, P, f) != P || fread(bgr[(size_t)i].data(), 1, P * 3, f) != P * 3 || fread(cam, 8, 21, f) != 21 || fread(rng, 4, 2, f) != 2 ||
		    fread(nb.data(), 4, (size_t)ns, f) != (size_t)ns) return 4;
		v.gray = gray[(size_t)i].data(); v.bgr = bgr[(size_t)i].data();
		memcpy(v.K, cam, 72); memcpy(v.R, cam + 9, 72); memcpy(v.C, cam + 18, 24);
		v.dMin = rng[0]; v.dMax = rng[1]; v.neighbors = nb; v.ID = (uint32_t)i;
		char name[64]; snprintf(name, sizeof(name), "images/%05d.jpg", i); v.name = name;
	}
	fclose(f);
	MVS::DenseDepthMapsHIP dense(0);
	if (!dense.IsValid()) { fprintf(stderr, "no device\n"); return 5; }
	MVS::DenseDepthMapsHIP::Options opt;
	opt.seed = argc > 4 ? (uint32_t)atoi(argv[4]) : 31u;
	opt.nSpeckleSize = argc > 5 ? (unsigned)atoi(argv[5]) : 100u;
	try {
		dense.LoadScene(views, w, h, opt);
		dense.ComputeDepthMaps();                                                // SceneDensify.cpp:1754-1982
		dense.SaveDepthMaps<DMapHeader>(argv[3], dmap_write);                    // DepthData::Save per view
		MVS::DenseDepthMapsHIP::PointCloud pc;
		dense.FuseDepthMaps(pc);                                                 // :1372-1650
		f = fopen(argv[2], "wb"); if (!f) return 6;
		std::vector<float> d(P), nrm(P * 3), c(P);
		for (int i = 0; i < n; ++i) { dense.GetMaps(i, d.data(), nrm.data(), c.data()); fwrite(d.data(), 4, P, f); fwrite(nrm.data(), 4, P * 3, f); fwrite(c.data(), 4, P, f); }
		const uint64_t cnt[2] = {(uint64_t)pc.size(), (uint64_t)pc.views.size()};
		fwrite(cnt, 8, 2, f); fwrite(pc.points.data(), 4, pc.points.size(), f); fwrite(pc.viewStart.data(), 4, pc.viewStart.size(), f); fwrite(pc.views.data(), 4, pc.views.size(), f);
		fclose(f);
		printf("depth maps %d, fused points %zu\n", n, pc.size());
	} catch (const std::exception& ex) { fprintf(stderr, "%s\n", ex.what()); return 7; }
	return 0;
}
