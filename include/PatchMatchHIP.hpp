// PatchMatchHIP.hpp -- header-only C++ adapter with the class surface of the reference's GPU plug-in
// `class PatchMatchCUDA` (libs/MVS/PatchMatchCUDA.inl:76-139): ctor(device), Init(bool), Release(),
// EstimateDepthMap(DepthData&).  It only repacks MVS::DepthData into the PODs of pmhip.h, so the three
// call sites of the reference (libs/MVS/SceneDensify.cpp:618-623, :1872-1881, :1910-1916) compile
// against it unchanged once `pmCUDA`'s type is switched.
//
// Templated on the reference's types so this header itself needs no OpenMVS/OpenCV include:
//   DepthDataT must look like MVS::DepthData (libs/MVS/DepthMap.h:157-271): .images[i].{image,camera,depthMap,
//   cameraDepthMap,GetID()}, .depthMap, .normalMap, .confMap, .mask (BitMatrix: empty(), isSet(r, c)), .dMin, .dMax; images are cv::Mat1f-like
//   (.cols, .rows, .empty(), .ptr<float>()/data, isContinuous()).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "pmhip.h"

namespace MVS {

class PatchMatchHIP {
public:
	struct Options : PMHipParams { Options() { pmhip_default_params(this); } };

	explicit PatchMatchHIP(int device = 0) : engine_(nullptr), geom_(false), ignoreMaskOption_(false), round_(0) {
		// like PatchMatchCUDA::PatchMatchCUDA (PatchMatchCUDA.cpp:46-51); IsValid() == false plays the role of
		// "CUDA::devices.IsEmpty()" at SceneDensify.cpp:1876-1877 (the caller then releases the plug-in)
		if (pmhip_create(device, &engine_) != PMHIP_OK) engine_ = nullptr;
	}
	~PatchMatchHIP() { if (engine_) pmhip_destroy(engine_); }
	PatchMatchHIP(const PatchMatchHIP&) = delete;
	PatchMatchHIP& operator=(const PatchMatchHIP&) = delete;

	bool IsValid() const { return engine_ != nullptr; }
	void Init(bool bGeomConsistency) { geom_ = bGeomConsistency; check(pmhip_init(engine_, bGeomConsistency ? 1 : 0)); }
	void Release() { if (engine_) check(pmhip_release(engine_)); }
	// OPTDENSE::nIgnoreMaskLabel >= 0 (DepthData::mask may still be empty for a view without a mask file; the option alone changes the
	// level hand-off to INTER_NEAREST, SceneDensify.cpp:661)
	void SetIgnoreMaskOption(bool on) { ignoreMaskOption_ = on; }

	// The reference's call site is `pmCUDA->EstimateDepthMap(arrDepthData[idxImage]);` (SceneDensify.cpp:620) -- one argument, with the options
	// in the OPTDENSE globals and the pass implied by Init(bGeomConsistency).  Bind both beforehand and that line compiles unchanged:
	//   SetOptions(opt) once (next to `pmCUDA->Init(...)`, :1880 / :1915), SetRound(geoIter) at the top of each geometric round (:1909-1916).
	void SetOptions(const Options& opt) { opt_ = opt; }
	const Options& GetOptions() const { return opt_; }
	void SetRound(int nGeometricIter) { round_ = nGeometricIter < 0 ? 0 : nGeometricIter; }
	template <typename DepthDataT>
	void EstimateDepthMap(DepthDataT& depthData) { EstimateDepthMap(depthData, opt_, geom_ ? round_ : -1); }

	// nGeometricIter: the argument of DepthMapsData::EstimateDepthMap (SceneDensify.cpp:616), -1 for the photometric pass.
	// The reference's PatchMatchCUDA infers it from Init(true); pass it explicitly here so the round index reaches the RNG key.
	template <typename DepthDataT>
	void EstimateDepthMap(DepthDataT& depthData, const Options& opt, int nGeometricIter = -1) {
		const int n = (int)depthData.images.size();
		std::vector<PMHipView> views((size_t)n);
		for (int i = 0; i < n; ++i) {
			auto& v = depthData.images[i];
			PMHipView& o = views[(size_t)i];
			o.image = v.image.template ptr<float>(); o.w = v.image.cols; o.h = v.image.rows;
			copy9(v.camera.K.val, o.K); copy9(v.camera.R.val, o.R); copy3(v.camera.C.ptr(), o.C);
			o.depth = nullptr; o.id = (uint32_t)v.GetID();
			if (i > 0 && !v.depthMap.empty()) {
				o.depth = v.depthMap.template ptr<float>(); o.dw = v.depthMap.cols; o.dh = v.depthMap.rows;   // read through cameraDepthMap, own size
				copy9(v.cameraDepthMap.K.val, o.Kd); copy9(v.cameraDepthMap.R.val, o.Rd); copy3(v.cameraDepthMap.C.ptr(), o.Cd);
			}
		}
		const int w = views[0].w, h = views[0].h;
		if (depthData.depthMap.empty()) { depthData.depthMap.create(h, w); depthData.depthMap.memset(0); }
		if (depthData.normalMap.empty()) { depthData.normalMap.create(h, w); depthData.normalMap.memset(0); }
		depthData.confMap.create(h, w);
		PMHipDepthData dd;
		dd.views = views.data(); dd.nViews = n;
		dd.depthMap = depthData.depthMap.template ptr<float>();
		dd.normalMap = reinterpret_cast<float*>(depthData.normalMap.data);
		dd.confMap = depthData.confMap.template ptr<float>();
		dd.dMin = depthData.dMin; dd.dMax = depthData.dMax;
		// DepthData::mask (BitMatrix, one bit per pixel, set = keep) as the byte mask of the C ABI
		std::vector<unsigned char> mask;
		if (!depthData.mask.empty()) {
			mask.resize((size_t)w * h);
			for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) mask[(size_t)r * w + c] = depthData.mask.isSet(r, c) ? 255 : 0;
		}
		check(pmhip_estimate_depth_map_masked(engine_, &dd, mask.empty() ? nullptr : mask.data(), ignoreMaskOption_ ? 1 : 0, &opt,
		                                      geom_ ? (nGeometricIter < 0 ? 0 : nGeometricIter) : -1));
	}

private:
	static void copy9(const double* s, double* d) { for (int i = 0; i < 9; ++i) d[i] = s[i]; }
	static void copy3(const double* s, double* d) { for (int i = 0; i < 3; ++i) d[i] = s[i]; }
	void check(int rc) const {
		// the reference exits on a CUDA error (libs/Common/UtilCUDA.h:81-91); we throw and let the caller decide
		if (rc != PMHIP_OK) throw std::runtime_error(std::string("pmhip: ") + (engine_ ? pmhip_last_error(engine_) : "no device"));
	}
	pmhip_engine* engine_;
	bool geom_, ignoreMaskOption_;
	int round_;
	Options opt_;
};

} // namespace MVS
