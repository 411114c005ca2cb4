// DenseDepthMapsHIP.hpp -- header-only C++ driver of the HBM-resident scene interface (include/pmhip.h, section 2) with the shape of
// Scene::ComputeDepthMaps + Scene::DenseReconstruction (libs/MVS/SceneDensify.cpp:1754-1982, :1655-1750): upload the views once, estimate ALL
// depth maps of a round concurrently (that is what fills an MI355X: the per-view seam at SceneDensify.cpp:618-623 hands the GPU one depth
// map at a time), keep every map on the device between the photometric pass, the geometric-consistency rounds, the per-map post-filters,
// the cross-view filter and the fusion, and come back to the host only with what the caller asks for (maps, .dmap files, the fused cloud).
//
// The reference's loop and what replaces it (see INTEGRATION.md section 1b for the call-site patch):
//   :1884-1905  photometric pass, one EstimateDepthMap per image on a worker queue      -> Estimate(-1)            (one pmhip_scene_estimate)
//   :1906-1953  for each geometric iteration: re-load neighbours' .dmap, estimate, save  -> CommitRound(); Estimate(iter)
//   :1884-1886,1919-1920 + :2069-2093  nOptimize: REMOVE_SPECKLES, FILL_GAPS             -> PostFilter()
//   :1955-1980 / :2136-2222  ADJUST_FILTER: DenseReconstructionFilter                    -> FilterDepthMaps()
//   :1695-1712 / :1372-1650  FuseDepthMaps (or MergeDepthMaps when nMinViewsFuse < 2)    -> FuseDepthMaps()
//   DepthData::Save (DepthMap.cpp:234-252)                                                -> SaveDepthMaps() through include/dmapio.h
//
// Needs nothing of OpenMVS / OpenCV: views are described by plain pointers (an OpenMVS caller fills them from MVS::Image / DepthData,
// libs/MVS/DepthMap.h:157-271).  Throws std::runtime_error with the engine's message on any error, like PatchMatchHIP.hpp.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "pmhip.h"

namespace MVS {

class DenseDepthMapsHIP {
public:
	// OPTDENSE::DepthFlags, libs/MVS/DepthMap.h:87-92
	enum DepthFlags { REMOVE_SPECKLES = 1, FILL_GAPS = 2, ADJUST_FILTER = 4, OPTIMIZE = REMOVE_SPECKLES | FILL_GAPS | ADJUST_FILTER };

	struct Options : PMHipParams {
		// defaults of libs/MVS/DepthMap.cpp:69-114
		unsigned nOptimize = OPTIMIZE;
		unsigned nSpeckleSize = 100, nIpolGapSize = 7;
		float fDepthDiffThreshold = 0.01f, fNormalDiffThreshold = 25.f;
		unsigned nMinViewsFilter = 2, nMinViewsFilterAdjust = 1, nMinViewsFuse = 2;
		bool bFilterAdjust = true, bEstimateColor = true, bEstimateNormal = true;
		Options() { pmhip_default_params(this); }
	};

	// one calibrated image (MVS::Image + the DepthData the reference builds for it in InitViews, SceneDensify.cpp:273-460)
	struct View {
		int w = 0, h = 0;                       // this image's own size; 0 = the size given to LoadScene (the reference sizes every DepthData on its image, SceneDensify.cpp:306-459)
		const float* gray = nullptr;            // w*h, gray in [0,1] (Image::toGray), row-major
		const unsigned char* bgr = nullptr;     // w*h*3 or null (needed for FuseDepthMaps with bEstimateColor)
		const unsigned char* mask = nullptr;    // w*h or null: 0 = ignored pixel (--ignore-mask-label, DepthMap.cpp:296-323)
		double K[9], R[9], C[3];                // pixel camera of this image: x = K R (X - C)
		float dMin = 0, dMax = 0;               // depth range of the sparse points seen by the view (DepthData::dMin/dMax)
		std::vector<int32_t> neighbors;         // indices into the view array, best first (DepthData::neighbors after SelectViews)
		const float* initDepth = nullptr;       // optional initial estimate (InitDepthMap, SceneDensify.cpp:418-460), w*h
		const float* initNormal = nullptr;      // w*h*3
		uint32_t ID = 0;                        // Image::ID as stored in the .dmap (the engine numbers views by their index in the array)
		float connections = -1.f;               // Image::neighbors.size() (all scored neighbours): fusion order; < 0 = use neighbors.size()
		std::string name;                       // image file name as stored in the .dmap
	};

	struct PointCloud {                         // MVS::PointCloud (libs/MVS/PointCloud.h): CSR view lists
		std::vector<float> points, weights, normals;
		std::vector<uint32_t> viewStart, views;
		std::vector<uint16_t> projs;
		std::vector<unsigned char> colors;
		size_t size() const { return points.size() / 3; }
	};

	explicit DenseDepthMapsHIP(int device = 0) : e_(nullptr), w_(0), h_(0) {
		if (pmhip_create(device, &e_) != PMHIP_OK) e_ = nullptr;   // IsValid() == false -> the caller keeps the CPU path (SceneDensify.cpp:1876-1877)
	}
	~DenseDepthMapsHIP() { if (e_) pmhip_destroy(e_); }
	DenseDepthMapsHIP(const DenseDepthMapsHIP&) = delete;
	DenseDepthMapsHIP& operator=(const DenseDepthMapsHIP&) = delete;
	bool IsValid() const { return e_ != nullptr; }

	// Upload the scene: every image once, for all rounds (the reference re-reads images and neighbours' depth maps per depth map).
	void LoadScene(const std::vector<View>& views, int w, int h, const Options& opt) {
		views_ = views; w_ = w; h_ = h; opt_ = opt;
		const int n = (int)views.size();
		check(pmhip_init(e_, 0));
		check(pmhip_scene_create(e_, n, w, h, (int)opt.nSubResolutionLevels));
		bool anyMask = false;
		for (int i = 0; i < n; ++i) {
			const View& v = views[i];
			if (!v.gray) throw std::runtime_error("DenseDepthMapsHIP: view without an image");
			if (v.w > 0 && v.h > 0 && (v.w != w || v.h != h))
				check(pmhip_scene_set_view_sized(e_, i, v.gray, v.w, v.h, 0, v.K, v.R, v.C, v.dMin, v.dMax, v.neighbors.data(), (int)v.neighbors.size()));
			else
				check(pmhip_scene_set_view(e_, i, v.gray, 0, v.K, v.R, v.C, v.dMin, v.dMax, v.neighbors.data(), (int)v.neighbors.size()));
			if (v.mask) { check(pmhip_scene_set_mask(e_, i, v.mask)); anyMask = true; }
			if (v.bgr) check(pmhip_scene_set_color(e_, i, v.bgr));
		}
		(void)anyMask;
		ids_.resize((size_t)n);
		for (int i = 0; i < n; ++i) ids_[(size_t)i] = i;
	}

	// Scene::ComputeDepthMaps: all rounds for all views, then the post-filters the option bits ask for.  Returns the number of depth maps.
	size_t ComputeDepthMaps() {
		const int n = (int)ids_.size();
		const unsigned G = opt_.nEstimationGeometricIters;
		check(pmhip_init(e_, 0));
		for (int i = 0; i < n; ++i) {
			check(pmhip_scene_reset_view(e_, i));
			const View& v = views_[(size_t)i];
			if (v.initDepth) check(pmhip_scene_set_maps(e_, i, v.initDepth, v.initNormal));
		}
		check(pmhip_scene_estimate(e_, ids_.data(), n, &opt_, -1, 0));                // photometric pass, SceneDensify.cpp:1884-1905
		if (G == 0) PostFilter();
		for (unsigned g = 0; g < G; ++g) {                                             // :1906-1953
			check(pmhip_scene_commit_round(e_));                                       // "save depthNNNN.dmap, reload as neighbour"
			check(pmhip_init(e_, 1));                                                  // pmCUDA->Release(); Init(true), :1910-1916
			check(pmhip_scene_estimate(e_, ids_.data(), n, &opt_, (int)g, 0));
			if (g + 1 == G) PostFilter();
		}
		if (opt_.nOptimize & ADJUST_FILTER) FilterDepthMaps();                         // :1955-1980
		check(pmhip_sync(e_));
		return (size_t)n;
	}

	// nOptimize bits applied to every depth map after the last estimation round (:1884-1886, :1919-1920, :2069-2093)
	void PostFilter() {
		const int n = (int)ids_.size();
		if (opt_.nOptimize & REMOVE_SPECKLES) check(pmhip_scene_remove_small_segments(e_, ids_.data(), n, opt_.nSpeckleSize, opt_.fDepthDiffThreshold));
		if (opt_.nOptimize & FILL_GAPS) check(pmhip_scene_gap_interpolation(e_, ids_.data(), n, opt_.nIpolGapSize, opt_.fDepthDiffThreshold));
	}
	// Scene::DenseReconstructionFilter, :2136-2222: every map against the unfiltered maps of its neighbours, then all replaced at once
	void FilterDepthMaps() {
		check(pmhip_scene_filter(e_, ids_.data(), (int)ids_.size(), opt_.bFilterAdjust ? 1 : 0, opt_.nMinViewsFilter, opt_.nMinViewsFilterAdjust, opt_.fDepthDiffThreshold, 1));
		check(pmhip_scene_filter_commit(e_));
	}

	// DepthMapsData::FuseDepthMaps, :1372-1650 (MergeDepthMaps when nMinViewsFuse < 2): images best connected first (:1423-1450)
	void FuseDepthMaps(PointCloud& pc) { FuseOn(e_, views_, opt_, pc); }
	// the same on any engine that holds the final maps of all views (the multi-device host fuses on one device after a gather)
	static void FuseOn(pmhip_engine* e, const std::vector<View>& views, const Options& opt, PointCloud& pc) {
		auto chk = [e](int rc) { if (rc != PMHIP_OK) throw std::runtime_error(std::string("pmhip: ") + pmhip_last_error(e)); };
		auto score = [&views](int32_t i) { const View& v = views[(size_t)i]; return v.connections < 0 ? (float)v.neighbors.size() : v.connections; };
		std::vector<int32_t> order;
		for (int32_t i = 0; i < (int32_t)views.size(); ++i) if (opt.nMinViewsFuse < 2 || score(i) > 0) order.push_back(i);     // connections with score <= 0 are dropped (:1451-1452)
		if (opt.nMinViewsFuse >= 2)
			std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return score(a) > score(b); });   // ties in index order
		PMHipFuseParams fp;
		fp.nMinViewsFuse = opt.nMinViewsFuse; fp.fDepthDiffThreshold = opt.fDepthDiffThreshold; fp.fNormalDiffThreshold = opt.fNormalDiffThreshold;
		bool haveColor = opt.bEstimateColor;
		for (const View& v : views) haveColor = haveColor && v.bgr != nullptr;
		fp.bEstimateColor = haveColor ? 1 : 0; fp.bEstimateNormal = opt.bEstimateNormal ? 1 : 0;
		uint64_t nP = 0, nV = 0, nD = 0;
		chk(pmhip_scene_fuse(e, order.data(), (int)order.size(), &fp, &nP, &nV, &nD));
		pc.points.assign((size_t)nP * 3, 0.f); pc.viewStart.assign((size_t)nP + 1, 0u); pc.views.assign((size_t)nV, 0u); pc.weights.assign((size_t)nV, 0.f);
		pc.projs.assign((size_t)nV * 2, (uint16_t)0);
		pc.colors.assign(haveColor ? (size_t)nP * 3 : 0, (unsigned char)0); pc.normals.assign(opt.bEstimateNormal ? (size_t)nP * 3 : 0, 0.f);
		chk(pmhip_scene_fuse_get(e, pc.points.data(), pc.viewStart.data(), pc.views.data(), pc.weights.data(), pc.projs.data(),
		                         haveColor ? pc.colors.data() : nullptr, opt.bEstimateNormal ? pc.normals.data() : nullptr));
	}

	// One view's maps back to the host (any pointer may be null); ViewWidth(idx) x ViewHeight(idx) entries
	void GetMaps(int idx, float* depth, float* normal, float* conf) { check(pmhip_scene_get_maps(e_, idx, depth, normal, conf)); }
	int ViewWidth(int idx) const { return views_[(size_t)idx].w > 0 ? views_[(size_t)idx].w : w_; }
	int ViewHeight(int idx) const { return views_[(size_t)idx].h > 0 ? views_[(size_t)idx].h : h_; }

	// DepthData::Save for every view: "<dir>/depthNNNN.dmap" (ComposeDepthFilePath, DepthMap.h:72), written through dmap_write (include/dmapio.h);
	// declared as a template on the writer so that this header does not force libdmapio on callers that never save
	template <typename DMapHeaderT, typename WriteFn>
	void SaveDepthMaps(const std::string& dir, WriteFn dmapWrite) {
		std::vector<float> d, nrm, c;
		for (size_t i = 0; i < views_.size(); ++i) {
			const View& v = views_[i];
			const int vw = ViewWidth((int)i), vh = ViewHeight((int)i);
			d.resize((size_t)vw * vh); nrm.resize((size_t)vw * vh * 3); c.resize((size_t)vw * vh);
			GetMaps((int)i, d.data(), nrm.data(), c.data());
			DMapHeaderT hdr; std::memset(&hdr, 0, sizeof(hdr));
			hdr.imageWidth = hdr.depthWidth = (uint32_t)vw; hdr.imageHeight = hdr.depthHeight = (uint32_t)vh;
			hdr.dMin = v.dMin; hdr.dMax = v.dMax; hdr.type = 1u | 2u | 4u;
			hdr.nIDs = (uint32_t)std::min<size_t>(v.neighbors.size() + 1, 256);
			hdr.IDs[0] = v.ID;
			for (uint32_t k = 1; k < hdr.nIDs; ++k) hdr.IDs[k] = views_[(size_t)v.neighbors[k - 1]].ID;
			std::memcpy(hdr.K, v.K, 72); std::memcpy(hdr.R, v.R, 72); std::memcpy(hdr.C, v.C, 24);
			std::snprintf(hdr.imageFileName, sizeof(hdr.imageFileName), "%s", v.name.c_str());
			char name[32]; std::snprintf(name, sizeof(name), "depth%04u.dmap", v.ID);
			if (dmapWrite((dir + "/" + name).c_str(), &hdr, d.data(), nrm.data(), c.data(), nullptr) != 0) throw std::runtime_error("DenseDepthMapsHIP: cannot write " + dir + "/" + name);
		}
	}

	pmhip_engine* engine() { return e_; }

private:
	void check(int rc) const {
		if (rc != PMHIP_OK) throw std::runtime_error(std::string("pmhip: ") + (e_ ? pmhip_last_error(e_) : "no device"));
	}
	pmhip_engine* e_;
	int w_, h_;
	Options opt_;
	std::vector<View> views_;
	std::vector<int32_t> ids_;
};

} // namespace MVS
