// Compiles include/PatchMatchHIP.hpp against minimal stand-ins of the reference's types (cv::Mat1f-like images,
// MVS::Camera, MVS::DepthData, libs/MVS/DepthMap.h:157-271) and calls it the way SceneDensify.cpp:618-623 would.
// Usage: adapter_smoke <w> <h> <nviews> <in.bin> <out.bin>   (in: nviews*(w*h floats + 21 doubles), dMin, dMax as floats)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "PatchMatchHIP.hpp"

template <typename T, int CH = 1> struct Mat {
	int cols = 0, rows = 0; unsigned char* data = nullptr; std::vector<T> buf;
	bool empty() const { return buf.empty(); }
	void create(int r, int c) { rows = r; cols = c; buf.assign((size_t)r * c * CH, T()); data = reinterpret_cast<unsigned char*>(buf.data()); }
	void memset(int v) { std::memset(buf.data(), v, buf.size() * sizeof(T)); }
	template <typename U> U* ptr() { return reinterpret_cast<U*>(buf.data()); }
};
struct Mat33 { double val[9]; };
struct Pt3 { double v[3]; const double* ptr() const { return v; } };
struct Camera { Mat33 K, R; Pt3 C; };
struct ViewData { Mat<float> image, depthMap; Camera camera, cameraDepthMap; unsigned id = 0; unsigned GetID() const { return id; } };
struct BitMatrix { std::vector<bool> bits; int rows = 0, cols = 0; bool empty() const { return bits.empty(); } bool isSet(int r, int c) const { return bits[(size_t)r * cols + c]; } };
struct DepthData { std::vector<ViewData> images; Mat<float> depthMap, confMap; Mat<float, 3> normalMap; BitMatrix mask; float dMin = 0, dMax = 0; };

int main(int argc, char** argv) {
	if (argc < 6) return 2;
	const int w = atoi(argv[1]), h = atoi(argv[2]), n = atoi(argv[3]);
	FILE* f = fopen(argv[4], "rb"); if (!f) return 3;
	DepthData dd; dd.images.resize(n);
	for (int i = 0; i < n; ++i) {
		ViewData& v = dd.images[i]; v.image.create(h, w); v.id = (unsigned)i;
		if (fread(v.image.buf.data(), 4, (size_t)w * h, f) != (size_t)w * h) return 4;
		double cam[21]; if (fread(cam, 8, 21, f) != 21) return 4;
		memcpy(v.camera.K.val, cam, 72); memcpy(v.camera.R.val, cam + 9, 72); memcpy(v.camera.C.v, cam + 18, 24);
	}
	float rng[2]; if (fread(rng, 4, 2, f) != 2) return 4; fclose(f);
	dd.dMin = rng[0]; dd.dMax = rng[1];
	MVS::PatchMatchHIP pm(0);
	if (!pm.IsValid()) { fprintf(stderr, "no device\n"); return 5; }     // SceneDensify.cpp:1876-1877: caller falls back
	pm.Init(false);
	MVS::PatchMatchHIP::Options opt; opt.seed = 7;
	pm.SetOptions(opt);                                                 // once, next to Init (SceneDensify.cpp:1880)
	pm.EstimateDepthMap(dd);                                            // SceneDensify.cpp:620, the reference's one-argument call, unchanged
	f = fopen(argv[5], "wb");
	fwrite(dd.depthMap.buf.data(), 4, (size_t)w * h, f); fwrite(dd.normalMap.buf.data(), 4, (size_t)w * h * 3, f); fwrite(dd.confMap.buf.data(), 4, (size_t)w * h, f);
	fclose(f);
	pm.Release();
	return 0;
}
