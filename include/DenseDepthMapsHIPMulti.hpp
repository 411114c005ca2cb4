// DenseDepthMapsHIPMulti.hpp -- the C++ host of the multi-GPU split (BASELINE.json north_star: "independent reference views shard embarrassingly
// across the 8 GPUs of one node with a single RCCL broadcast of the image set over xGMI and no per-iteration collectives").
//
// One process, one pmhip engine and one host thread per device.  Every engine holds the whole scene (images of all views: any view can be a
// source view) and estimates a contiguous block of reference views; what crosses devices is
//   (1) ONE broadcast of the image set from the device the caller's images were uploaded to;
//   (2) at each round boundary the depth maps a device READS and does not own -- the source views of its reference views on other devices -- point to point from their
//       owners (the reference reloads the neighbours' depthNNNN.dmap there, SceneDensify.cpp:378-393); the same for depth + confidence before the cross-view filter
//       (:2136-2222); depth + normal + confidence of every block to the ONE fusing device before FuseDepthMaps (:1372-1650, sequential over the scene);
// nothing inside a sweep.  (The Python driver, openmvs_amd/distributed.py, goes one step further and holds only the needed views per rank: a compact scene of local slots.)  Same call shape as DenseDepthMapsHIP (LoadScene, ComputeDepthMaps, FuseDepthMaps, GetMaps), same results bit for bit:
// a view's maps do not depend on which device estimated them (tests/cpp/dense_multi.cpp).
//
// The collectives are a policy:
//   RcclCollective       (define PMHIP_WITH_RCCL, link -lrccl): ncclCommInitAll over the devices, grouped ncclBroadcast calls on the engines' streams --
//                        the product path; the round-boundary exchanges are grouped ncclSend / ncclRecv pairs of exactly the maps a device reads.
//   LocalCopyCollective  several engines on ONE device (debugging, and the single-GPU CI box): device-to-device copies between the engines' arrays.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>
#include "DenseDepthMapsHIP.hpp"
#ifdef PMHIP_WITH_RCCL
#include <rccl/rccl.h>
#endif

namespace MVS {

struct LocalCopyCollective {
	explicit LocalCopyCollective(const std::vector<int>&) {}
	void Broadcast(const std::vector<void*>& bufs, size_t bytes, int root, const std::vector<hipStream_t>& streams) {
		sync(streams);
		for (size_t d = 0; d < bufs.size(); ++d) if ((int)d != root && hipMemcpy(bufs[d], bufs[(size_t)root], bytes, hipMemcpyDeviceToDevice) != hipSuccess) throw std::runtime_error("LocalCopyCollective: copy failed");
	}
	// point-to-point copies: every transfer moves `bytes` from device src's buffer to device dst's
	struct Xfer { int src, dst; const void* from; void* to; size_t bytes; };
	void Exchange(const std::vector<Xfer>& xs, const std::vector<hipStream_t>& streams) {
		sync(streams);
		for (const Xfer& x : xs) if (x.bytes && hipMemcpy(x.to, x.from, x.bytes, hipMemcpyDeviceToDevice) != hipSuccess) throw std::runtime_error("LocalCopyCollective: copy failed");
	}
private:
	static void sync(const std::vector<hipStream_t>& streams) { for (hipStream_t s : streams) hipStreamSynchronize(s); }
};

#ifdef PMHIP_WITH_RCCL
class RcclCollective {
public:
	explicit RcclCollective(const std::vector<int>& devices) : devs_(devices), comms_(devices.size()) {
		if (ncclCommInitAll(comms_.data(), (int)devs_.size(), devs_.data()) != ncclSuccess) throw std::runtime_error("RcclCollective: ncclCommInitAll failed");
	}
	~RcclCollective() { for (ncclComm_t c : comms_) ncclCommDestroy(c); }
	RcclCollective(const RcclCollective&) = delete;
	RcclCollective& operator=(const RcclCollective&) = delete;
	// the single broadcast of the image set (xGMI); enqueued on each engine's stream, so it is ordered with that engine's kernels
	void Broadcast(const std::vector<void*>& bufs, size_t bytes, int root, const std::vector<hipStream_t>& streams) {
		ok(ncclGroupStart());
		for (size_t d = 0; d < devs_.size(); ++d) { hipSetDevice(devs_[d]); ok(ncclBroadcast(bufs[d], bufs[d], bytes, ncclChar, root, comms_[d], streams[d])); }
		ok(ncclGroupEnd());
	}
	typedef LocalCopyCollective::Xfer Xfer;
	// grouped ncclSend / ncclRecv pairs on the two engines' streams (xGMI is point to point: a transfer uses the link between its two devices and nothing else)
	void Exchange(const std::vector<Xfer>& xs, const std::vector<hipStream_t>& streams) {
		ok(ncclGroupStart());
		for (const Xfer& x : xs) if (x.bytes) {
			hipSetDevice(devs_[(size_t)x.src]); ok(ncclSend(x.from, x.bytes, ncclChar, x.dst, comms_[(size_t)x.src], streams[(size_t)x.src]));
			hipSetDevice(devs_[(size_t)x.dst]); ok(ncclRecv(x.to, x.bytes, ncclChar, x.src, comms_[(size_t)x.dst], streams[(size_t)x.dst]));
		}
		ok(ncclGroupEnd());
	}
private:
	static void ok(ncclResult_t r) { if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL: ") + ncclGetErrorString(r)); }
	std::vector<int> devs_;
	std::vector<ncclComm_t> comms_;
};
#endif

template <class Collective>
class DenseDepthMapsHIPMultiT {
public:
	typedef DenseDepthMapsHIP::Options Options;
	typedef DenseDepthMapsHIP::View View;
	typedef DenseDepthMapsHIP::PointCloud PointCloud;
	enum { REMOVE_SPECKLES = DenseDepthMapsHIP::REMOVE_SPECKLES, FILL_GAPS = DenseDepthMapsHIP::FILL_GAPS, ADJUST_FILTER = DenseDepthMapsHIP::ADJUST_FILTER };

	// devices: HIP device ordinals, one engine each (LocalCopyCollective: the same ordinal several times)
	// hostThreads = false drives the engines one after the other from the calling thread (same results; for single-threaded test environments)
	explicit DenseDepthMapsHIPMultiT(const std::vector<int>& devices, bool hostThreads = true) : devs_(devices), w_(0), h_(0), threads_(hostThreads) {
		for (int d : devs_) { pmhip_engine* e = nullptr; if (pmhip_create(d, &e) != PMHIP_OK) { release(); return; } eng_.push_back(e); }
		coll_.reset(new Collective(devs_));
	}
	~DenseDepthMapsHIPMultiT() { release(); }
	DenseDepthMapsHIPMultiT(const DenseDepthMapsHIPMultiT&) = delete;
	DenseDepthMapsHIPMultiT& operator=(const DenseDepthMapsHIPMultiT&) = delete;
	bool IsValid() const { return !eng_.empty() && eng_.size() == devs_.size(); }
	int NumDevices() const { return (int)eng_.size(); }
	// contiguous blocks that differ by at most one view (the split of openmvs_amd/distributed.py::shard_range)
	static void ShardRange(int nViews, int nDev, int r, int& first, int& count) { const int q = nViews / nDev, m = nViews % nDev; first = r * q + std::min(r, m); count = q + (r < m ? 1 : 0); }

	void LoadScene(const std::vector<View>& views, int w, int h, const Options& opt) {
		views_ = views; w_ = w; h_ = h; opt_ = opt;
		const int n = (int)views.size(), D = NumDevices();
		first_.resize((size_t)D); count_.resize((size_t)D);
		for (int d = 0; d < D; ++d) ShardRange(n, D, d, first_[(size_t)d], count_[(size_t)d]);
		// the foreign views every device reads: the source views of its block that another device owns (ascending)
		needs_.assign((size_t)D, std::vector<int>());
		for (int d = 0; d < D; ++d) {
			std::vector<char> mark((size_t)n, 0);
			for (int i = first_[(size_t)d]; i < first_[(size_t)d] + count_[(size_t)d]; ++i) for (int32_t nb : views[(size_t)i].neighbors) if (nb >= 0 && nb < n && !owns(d, nb)) mark[(size_t)nb] = 1;
			for (int i = 0; i < n; ++i) if (mark[(size_t)i]) needs_[(size_t)d].push_back(i);
		}
		for (int d = 0; d < D; ++d) {
			pmhip_engine* e = eng_[(size_t)d];
			check(d, pmhip_init(e, 0));
			check(d, pmhip_scene_create(e, n, w, h, (int)opt.nSubResolutionLevels));
			for (int i = 0; i < n; ++i) {
				const View& v = views[(size_t)i];
				if (!v.gray) throw std::runtime_error("DenseDepthMapsHIPMulti: view without an image");
				// images go up once, to the first device; the others get cameras and neighbour lists now and the pixels by the broadcast below.  A view with its own
				// size (DepthMapsData::InitViews sizes every depth map on its own image, SceneDensify.cpp:306-459) lives outside the scene's image array: every device
				// gets it from the host (such views are the exception -- a neighbour rescaled by ViewData::ScaleImage)
				if (sized(i)) check(d, pmhip_scene_set_view_sized(e, i, v.gray, v.w, v.h, 0, v.K, v.R, v.C, v.dMin, v.dMax, v.neighbors.data(), (int)v.neighbors.size()));
				else check(d, pmhip_scene_set_view(e, i, d == 0 ? v.gray : nullptr, 0, v.K, v.R, v.C, v.dMin, v.dMax, v.neighbors.data(), (int)v.neighbors.size()));
				if (v.mask) check(d, pmhip_scene_set_mask(e, i, v.mask));
				if (v.bgr && d == 0) check(d, pmhip_scene_set_color(e, i, v.bgr));    // colours are only read by the fusing device
			}
		}
		// (1) the one broadcast of the image set
		std::vector<void*> bufs; for (pmhip_engine* e : eng_) bufs.push_back(pmhip_scene_device_ptr(e, 0, 0));
		coll_->Broadcast(bufs, sizeof(float) * (size_t)w * h * n, 0, streams());
		for (int d = 0; d < D; ++d) check(d, pmhip_scene_images_updated(eng_[(size_t)d]));
	}

	size_t ComputeDepthMaps() {
		const int n = (int)views_.size(), D = NumDevices();
		const unsigned G = opt_.nEstimationGeometricIters;
		for (int d = 0; d < D; ++d) {
			check(d, pmhip_init(eng_[(size_t)d], 0));
			for (int i = 0; i < n; ++i) {
				check(d, pmhip_scene_reset_view(eng_[(size_t)d], i));
				const View& v = views_[(size_t)i];
				if (v.initDepth && owns(d, i)) check(d, pmhip_scene_set_maps(eng_[(size_t)d], i, v.initDepth, v.initNormal));
			}
		}
		estimateAll(-1);                                                               // photometric pass, SceneDensify.cpp:1884-1905
		if (G == 0) postFilterAll();
		for (unsigned g = 0; g < G; ++g) {                                             // :1906-1953
			gatherNeighbours(1);                                                       // (2) every device gets the depth maps of the round it reads ...
			for (int d = 0; d < D; ++d) { check(d, pmhip_scene_commit_round(eng_[(size_t)d])); check(d, pmhip_init(eng_[(size_t)d], 1)); }   // ... as the snapshot the geometric term reads
			estimateAll((int)g);
			if (g + 1 == G) postFilterAll();
		}
		if (opt_.nOptimize & ADJUST_FILTER) {                                          // :1955-1980, every map against the UNFILTERED maps of its neighbours
			gatherNeighbours(1); gatherNeighbours(3);
			perDevice([&](int d) {
				check(d, pmhip_scene_filter(eng_[(size_t)d], ids(d).data(), count_[(size_t)d], opt_.bFilterAdjust ? 1 : 0, opt_.nMinViewsFilter, opt_.nMinViewsFilterAdjust, opt_.fDepthDiffThreshold, 1));
				check(d, pmhip_scene_filter_commit(eng_[(size_t)d]));
			});
		}
		for (int d = 0; d < D; ++d) check(d, pmhip_sync(eng_[(size_t)d]));
		return (size_t)n;
	}

	// FuseDepthMaps is sequential over the scene (images best connected first, claimed pixels carried from image to image): one device, after a gather
	void FuseDepthMaps(PointCloud& pc) {
		gatherTo(0, 1); gatherTo(0, 2); gatherTo(0, 3);
		for (int d = 0; d < NumDevices(); ++d) check(d, pmhip_sync(eng_[(size_t)d]));
		DenseDepthMapsHIP::FuseOn(eng_[0], views_, opt_, pc);
	}
	// a view's maps from the device that owns it; ViewWidth(idx) x ViewHeight(idx) entries
	int ViewWidth(int idx) const { return sized(idx) ? views_[(size_t)idx].w : w_; }
	int ViewHeight(int idx) const { return sized(idx) ? views_[(size_t)idx].h : h_; }
	void GetMaps(int idx, float* depth, float* normal, float* conf) {
		for (int d = 0; d < NumDevices(); ++d) if (owns(d, idx)) { check(d, pmhip_scene_get_maps(eng_[(size_t)d], idx, depth, normal, conf)); return; }
		throw std::runtime_error("DenseDepthMapsHIPMulti: no such view");
	}
	pmhip_engine* engine(int d) { return eng_[(size_t)d]; }

private:
	bool owns(int d, int i) const { return i >= first_[(size_t)d] && i < first_[(size_t)d] + count_[(size_t)d]; }
	std::vector<int32_t> ids(int d) const { std::vector<int32_t> v((size_t)count_[(size_t)d]); for (int k = 0; k < count_[(size_t)d]; ++k) v[(size_t)k] = first_[(size_t)d] + k; return v; }
	std::vector<hipStream_t> streams() const { std::vector<hipStream_t> s; for (pmhip_engine* e : eng_) s.push_back((hipStream_t)pmhip_stream(e)); return s; }
	template <class F> void perDevice(F f) {
		// one host thread per device: the engines enqueue their kernels concurrently; an exception of any thread is rethrown here
		if (!threads_) { for (int d = 0; d < NumDevices(); ++d) f(d); return; }
		std::vector<std::thread> th; std::vector<std::string> errs((size_t)NumDevices());
		for (int d = 0; d < NumDevices(); ++d) th.emplace_back([&, d]() { try { f(d); } catch (const std::exception& ex) { errs[(size_t)d] = ex.what(); } });
		for (auto& t : th) t.join();
		for (const std::string& e : errs) if (!e.empty()) throw std::runtime_error(e);
	}
	void estimateAll(int geoIter) {
		perDevice([&](int d) { if (count_[(size_t)d]) check(d, pmhip_scene_estimate(eng_[(size_t)d], ids(d).data(), count_[(size_t)d], &opt_, geoIter, 0)); });
	}
	void postFilterAll() {
		perDevice([&](int d) {
			if (!count_[(size_t)d]) return;
			if (opt_.nOptimize & REMOVE_SPECKLES) check(d, pmhip_scene_remove_small_segments(eng_[(size_t)d], ids(d).data(), count_[(size_t)d], opt_.nSpeckleSize, opt_.fDepthDiffThreshold));
			if (opt_.nOptimize & FILL_GAPS) check(d, pmhip_scene_gap_interpolation(eng_[(size_t)d], ids(d).data(), count_[(size_t)d], opt_.nIpolGapSize, opt_.fDepthDiffThreshold));
		});
	}
	// Bytes that crossed devices so far (round boundaries, filter, fusion gather; not the image broadcast)
public:
	size_t ExchangedBytes() const { return exchanged_; }
	const std::vector<int>& ForeignViews(int d) const { return needs_[(size_t)d]; }
private:
	// a view that carries its own image size keeps its own maps on every device (pmhip_scene_device_ptr(e, what, idx) finds them); the others sit side by side in the scene arrays
	bool sized(int i) const { const View& v = views_[(size_t)i]; return v.w > 0 && v.h > 0 && (v.w != w_ || v.h != h_); }
	size_t mapBytes(int what, int i) const { return sizeof(float) * (size_t)ViewWidth(i) * ViewHeight(i) * (what == 2 ? 3 : 1); }
	// transfers of the ascending views `list` (all owned by device src) to device dst: runs of consecutive scene-size views as one transfer, a view of its own size on its own
	void addTransfers(std::vector<typename Collective::Xfer>& xs, int what, int src, int dst, const std::vector<int>& list) {
		for (size_t k = 0; k < list.size(); ) {
			size_t e = k + 1;
			if (!sized(list[k])) while (e < list.size() && list[e] == list[e - 1] + 1 && !sized(list[e])) ++e;
			typename Collective::Xfer x; x.src = src; x.dst = dst; x.bytes = 0;
			for (size_t q = k; q < e; ++q) x.bytes += mapBytes(what, list[q]);
			x.from = pmhip_scene_device_ptr(eng_[(size_t)src], what, list[k]);
			x.to = pmhip_scene_device_ptr(eng_[(size_t)dst], what, list[k]);
			xs.push_back(x); exchanged_ += x.bytes; k = e;
		}
	}
	// one per-view array (what: 1 depth, 2 normal, 3 conf): every device receives the views it reads from their owners
	void gatherNeighbours(int what) {
		std::vector<typename Collective::Xfer> xs;
		for (int d = 0; d < NumDevices(); ++d) {
			const std::vector<int>& nd = needs_[(size_t)d];
			for (size_t k = 0; k < nd.size(); ) {
				const int r = ownerOf(nd[k]); size_t e = k + 1;
				while (e < nd.size() && ownerOf(nd[e]) == r) ++e;
				addTransfers(xs, what, r, d, std::vector<int>(nd.begin() + (long)k, nd.begin() + (long)e));
				k = e;
			}
		}
		coll_->Exchange(xs, streams());
		if (what == 1) for (int d = 0; d < NumDevices(); ++d) check(d, pmhip_scene_maps_updated(eng_[(size_t)d], 0, (int)views_.size()));
	}
	// every block of one per-view array to ONE device (the fusing one)
	void gatherTo(int root, int what) {
		std::vector<typename Collective::Xfer> xs;
		for (int r = 0; r < NumDevices(); ++r) if (r != root && count_[(size_t)r]) {
			std::vector<int> block; for (int32_t i : ids(r)) block.push_back(i);
			addTransfers(xs, what, r, root, block);
		}
		coll_->Exchange(xs, streams());
		if (what == 1) check(root, pmhip_scene_maps_updated(eng_[(size_t)root], 0, (int)views_.size()));
	}
	int ownerOf(int i) const { for (int d = 0; d < NumDevices(); ++d) if (owns(d, i)) return d; return 0; }
	void check(int d, int rc) const { if (rc != PMHIP_OK) throw std::runtime_error("pmhip (device " + std::to_string(devs_[(size_t)d]) + "): " + pmhip_last_error(eng_[(size_t)d])); }
	void release() { for (pmhip_engine* e : eng_) pmhip_destroy(e); eng_.clear(); }

	std::vector<int> devs_;
	std::vector<pmhip_engine*> eng_;
	std::unique_ptr<Collective> coll_;
	int w_, h_;
	bool threads_;
	Options opt_;
	std::vector<View> views_;
	std::vector<int> first_, count_;
	std::vector<std::vector<int>> needs_;
	size_t exchanged_ = 0;
};

#ifdef PMHIP_WITH_RCCL
typedef DenseDepthMapsHIPMultiT<RcclCollective> DenseDepthMapsHIPMulti;
#endif

} // namespace MVS
