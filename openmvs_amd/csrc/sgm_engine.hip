// sgm_engine.hip -- host side of libsgmhip.so (include/sgmhip.h).
#include "../../include/sgmhip.h"
#include "sgm_kernels.hip"
#include "sgm_kernels_sub.hip"
#include "sgm_post.hip"
#include "sgm_tsgm.hip"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#define SGMCHK(e, call) do { hipError_t _r = (call); if (_r != hipSuccess) { (e)->err = std::string(#call) + ": " + hipGetErrorString(_r); return SGMHIP_E_HIP; } } while (0)

struct sgmhip_engine {
	int device = 0; hipStream_t stream = nullptr; std::string err;
	int w = 0, h = 0, vw = 0, vh = 0, maxNumDisp = 0; uint64_t numCosts = 0;
	size_t capImg = 0, capPix = 0, capCosts = 0;
	unsigned char* d_color = nullptr; float* d_grayL = nullptr; float* d_grayR = nullptr;
	SGMPixel* d_pixels = nullptr; unsigned char* d_costs = nullptr; unsigned short* d_accums = nullptr; float4* d_setup = nullptr;
	short* d_disp = nullptr; unsigned short* d_cost = nullptr; unsigned short* d_P2s = nullptr;
	bool statsOn = false; SGMHipStats stats{};
	unsigned char* d_deltas = nullptr; size_t capDeltas = 0; int maxP2 = 65535;   // DELTA aggregation (uniform ranges, max P2 <= 255): 8 byte volumes instead of atomic u16 sums
	bool uniform = false; int uniformMin = 0, uniformMax = 0; SGMUniform* d_uniform = nullptr;   // every pixel has [uniformMin, uniformMax) and idx = pixel * nD (checked on the device at set_problem): the register-resident path kernel applies
	int subGroups = 0;            // 0, or the lanes per sub-group (8, 16, 32): Match with the sub-group kernels of sgm_kernels_sub.hip (narrow, ragged ranges); see sgmhip_set_sub_group_kernels
	struct Ev { hipEvent_t a, b; int kind; }; std::vector<Ev> events;
};

static void sgmFree(sgmhip_engine* e) {
	hipSetDevice(e->device);
	void* ps[] = {e->d_color, e->d_grayL, e->d_grayR, e->d_pixels, e->d_costs, e->d_accums, e->d_disp, e->d_cost, e->d_setup};
	for (void* p : ps) if (p) hipFree(p);
	e->d_color = nullptr; e->d_grayL = e->d_grayR = nullptr; e->d_pixels = nullptr; e->d_costs = nullptr; e->d_accums = nullptr; e->d_disp = nullptr; e->d_cost = nullptr; e->d_setup = nullptr;
	e->capImg = e->capPix = e->capCosts = 0;
}
static void evB(sgmhip_engine* e, int kind) { if (!e->statsOn) return; sgmhip_engine::Ev ev; ev.kind = kind; hipEventCreate(&ev.a); hipEventCreate(&ev.b); hipEventRecord(ev.a, e->stream); e->events.push_back(ev); }
static void evE(sgmhip_engine* e) { if (!e->statsOn) return; hipEventRecord(e->events.back().b, e->stream); }
static int sgmCollect(sgmhip_engine* e) {
	if (e->events.empty()) return 0;
	SGMCHK(e, hipStreamSynchronize(e->stream));
	for (auto& ev : e->events) { float ms = 0; hipEventElapsedTime(&ms, ev.a, ev.b); (ev.kind == 0 ? e->stats.costMs : ev.kind == 1 ? e->stats.aggrMs : e->stats.wtaMs) += ms; hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
	e->events.clear();
	return 0;
}

// buffers of the resident problem (grown, never shrunk) and its dimensions
static int sgmReserve(sgmhip_engine* e, int w, int h, uint64_t numCosts, int maxNumDisp) {
	const size_t nImg = (size_t)w * h, nPix = (size_t)(w - 2 * SGM_HW) * (h - 2 * SGM_HW);
	if (nImg > e->capImg || nPix > e->capPix || numCosts > e->capCosts) {
		SGMCHK(e, hipStreamSynchronize(e->stream));
		const size_t cI = std::max(nImg, e->capImg), cP = std::max(nPix, e->capPix), cC = std::max<size_t>(numCosts, e->capCosts);
		sgmFree(e);
		SGMCHK(e, hipMalloc(&e->d_color, cI * 3)); SGMCHK(e, hipMalloc(&e->d_grayL, cI * 4)); SGMCHK(e, hipMalloc(&e->d_grayR, cI * 4));
		SGMCHK(e, hipMalloc(&e->d_pixels, cP * sizeof(SGMPixel))); SGMCHK(e, hipMalloc(&e->d_disp, cP * 2)); SGMCHK(e, hipMalloc(&e->d_cost, cP * 2)); SGMCHK(e, hipMalloc(&e->d_setup, cP * sizeof(float4)));
		SGMCHK(e, hipMalloc(&e->d_costs, cC + 256)); SGMCHK(e, hipMalloc(&e->d_accums, (cC + 3) / 4 * 8 + 8)); // u16 sums, addressed as 32-bit words by the path kernels
		e->capImg = cI; e->capPix = cP; e->capCosts = cC;
	}
	e->w = w; e->h = h; e->vw = w - 2 * SGM_HW; e->vh = h - 2 * SGM_HW; e->numCosts = numCosts; e->maxNumDisp = maxNumDisp;
	e->uniform = false;
	return 0;
}

extern "C" {

int sgmhip_create(int device, sgmhip_engine** out) {
	if (!out) return SGMHIP_E_ARG;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) return SGMHIP_E_NODEVICE;
	if (device < 0) device = 0;
	sgmhip_engine* e = new sgmhip_engine(); e->device = device;
	if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&e->d_P2s, 512) != hipSuccess) { delete e; return SGMHIP_E_HIP; }
	*out = e;
	return 0;
}
void sgmhip_destroy(sgmhip_engine* e) {
	if (!e) return;
	hipSetDevice(e->device); hipStreamSynchronize(e->stream);
	for (auto& ev : e->events) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
	sgmFree(e); if (e->d_P2s) hipFree(e->d_P2s); if (e->d_uniform) hipFree(e->d_uniform); if (e->d_deltas) hipFree(e->d_deltas);
	hipStreamDestroy(e->stream); delete e;
}
const char* sgmhip_last_error(sgmhip_engine* e) { return e ? e->err.c_str() : "null engine"; }

int sgmhip_generate_p2s(uint16_t P2, float alpha, float beta, uint16_t out256[256]) {
	if (!out256) return SGMHIP_E_ARG;
	for (int i = 0; i < 256; ++i) { const float fi = (float)i; out256[i] = (uint16_t)(int)floorf((float)P2 * (1.f + alpha * pm_expf(-(fi * fi) / (2.f * (beta * beta)))) + .5f); }
	return 0;
}

int sgmhip_set_problem(sgmhip_engine* e, const uint8_t* leftBGR, const float* leftGray, const float* rightGray, int w, int h,
		const SGMHipPixelData* pixels, uint64_t numCosts, int maxNumDisp) {
	if (!e || !leftBGR || !leftGray || !rightGray || !pixels || w <= 2 * SGM_HW || h <= 2 * SGM_HW || numCosts == 0 || maxNumDisp <= 0 || maxNumDisp > 256) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	const size_t nImg = (size_t)w * h, nPix = (size_t)(w - 2 * SGM_HW) * (h - 2 * SGM_HW);
	{ const int rc = sgmReserve(e, w, h, numCosts, maxNumDisp); if (rc) return rc; }
	SGMCHK(e, hipMemcpyAsync(e->d_color, leftBGR, nImg * 3, hipMemcpyHostToDevice, e->stream));
	SGMCHK(e, hipMemcpyAsync(e->d_grayL, leftGray, nImg * 4, hipMemcpyHostToDevice, e->stream));
	SGMCHK(e, hipMemcpyAsync(e->d_grayR, rightGray, nImg * 4, hipMemcpyHostToDevice, e->stream));
	SGMCHK(e, hipMemcpyAsync(e->d_pixels, pixels, nPix * sizeof(SGMPixel), hipMemcpyHostToDevice, e->stream));
	// one range for all pixels?  (then Match aggregates with sgm_path_uniform_kernel)
	static const bool allowUniform = [] { const char* v = getenv("SGMHIP_UNIFORM"); return !v || atoi(v) != 0; }();
	SGMUniform hu = {1, 0, 0, 0};
	if (allowUniform) {
		if (!e->d_uniform) SGMCHK(e, hipMalloc(&e->d_uniform, sizeof(SGMUniform)));
		SGMCHK(e, hipMemcpyAsync(e->d_uniform, &hu, sizeof(hu), hipMemcpyHostToDevice, e->stream));
		hipLaunchKernelGGL(sgm_uniform_check_kernel, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, e->stream, e->d_pixels, (long)nPix, e->d_uniform);
		SGMCHK(e, hipMemcpyAsync(&hu, e->d_uniform, sizeof(hu), hipMemcpyDeviceToHost, e->stream));
	} else hu.ok = 0;
	SGMCHK(e, hipStreamSynchronize(e->stream));
	e->uniform = hu.ok != 0 && hu.maxDisp - hu.minDisp == maxNumDisp && (uint64_t)nPix * (uint64_t)maxNumDisp == numCosts;
	e->uniformMin = hu.minDisp; e->uniformMax = hu.maxDisp;
	return 0;
}

static void launchPath(sgmhip_engine* e, hipStream_t st, int NK, int lines, int P1, const SGMDirs& dirs, bool delta) {
#define SGM_LAUNCH_PATH(NK_, DL_) hipLaunchKernelGGL((sgm_path_kernel<NK_, DL_>), dim3(lines), dim3(64), 0, st, e->d_grayL, e->w, e->vw, e->vh, e->d_pixels, e->d_costs, (unsigned*)e->d_accums, e->d_P2s, P1, dirs, e->d_deltas, (unsigned long long)e->numCosts)
	switch (NK) {
	case 1: if (delta) SGM_LAUNCH_PATH(1, true); else SGM_LAUNCH_PATH(1, false); break;
	case 2: if (delta) SGM_LAUNCH_PATH(2, true); else SGM_LAUNCH_PATH(2, false); break;
	default: if (delta) SGM_LAUNCH_PATH(4, true); else SGM_LAUNCH_PATH(4, false); break;
	}
#undef SGM_LAUNCH_PATH
}
// the 8 byte volumes of the DELTA aggregation (kept between calls, grown on demand); false: no room, the caller takes the atomic path
static bool ensureDeltas(sgmhip_engine* e) {
	if (e->capDeltas >= e->numCosts * 8) return true;
	if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
	if (e->d_deltas) { hipFree(e->d_deltas); e->d_deltas = nullptr; e->capDeltas = 0; }
	if (hipMalloc(&e->d_deltas, e->numCosts * 8 + 16) == hipSuccess) { e->capDeltas = e->numCosts * 8; return true; }
	(void)hipGetLastError();
	return false;
}

static int sgmMatch(sgmhip_engine* e, uint16_t P1);
int sgmhip_match(sgmhip_engine* e, uint16_t P1, const uint16_t P2s[256], int sync) {
	if (!e || !P2s || e->numCosts == 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	SGMCHK(e, hipMemcpyAsync(e->d_P2s, P2s, 512, hipMemcpyHostToDevice, e->stream));
	e->maxP2 = 0; for (int i = 0; i < 256; ++i) e->maxP2 = std::max(e->maxP2, (int)P2s[i]);
	{ const int rc = sgmMatch(e, P1); if (rc) return rc; }
	if (sync) SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}
// cost volume, 8-path aggregation and winner-take-all of the resident problem (P2s already on the device), asynchronous on the engine's stream
// sub-group variant: LP lanes per pixel / pair / line
extern "C++" {
template <int LP>
static int sgmMatchSubT(sgmhip_engine* e, uint16_t P1) {
	constexpr int PW = 64 / LP;
	const long nPix = (long)e->vw * e->vh;
	const int W = e->vw, H = e->vh;
	evB(e, 0);
	// cost volume: the lane-per-pixel kernel serves narrow ranges too (0.9 against 1.28 ms for 3-12 disparities per pixel at 2048x1536); SGMHIP_COST_PX=0: the sub-group kernel
	static const bool pxCost = [] { const char* v = getenv("SGMHIP_COST_PX"); return !v || atoi(v) != 0; }();
	if (pxCost) {
		const long nTiles = (long)((W + 63) / 64) * H;
		hipLaunchKernelGGL(sgm_cost_px_kernel, dim3((unsigned)((nTiles + 3) / 4)), dim3(256), 0, e->stream, e->d_color, e->d_grayL, e->d_grayR, e->w, e->h, W, H, e->d_pixels, e->d_costs);
	} else {
		hipLaunchKernelGGL(sgm_setup_kernel, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, e->stream, e->d_color, e->d_grayL, e->w, W, H, e->d_pixels, e->d_setup);
		const long nPairs = (long)((W + 1) / 2) * H;
		hipLaunchKernelGGL((sgm_cost_sub_kernel<LP>), dim3((unsigned)((nPairs + 4 * PW - 1) / (4 * PW))), dim3(256), 0, e->stream, e->d_color, e->d_grayL, e->d_grayR, e->w, e->h, W, H, e->d_pixels, e->d_setup, e->d_costs);
	}
	evE(e);
	static const int allowDeltaSub = [] { const char* v = getenv("SGMHIP_DELTA"); return v ? atoi(v) : 3; }();
	bool delta = e->maxP2 <= 255 && (allowDeltaSub & 2) != 0;
	if (delta && !ensureDeltas(e)) delta = false;
	if (!delta) SGMCHK(e, hipMemsetAsync(e->d_accums, 0, (e->numCosts + 1) / 2 * 4, e->stream));
	struct Dir { int dx, dy; SGMLines ln; } dirs[8] = {
		{0, 1,   {W, 0, 0, 1, 0,      0, 0, 0, 0, 0}},
		{1, 0,   {H, 0, 0, 0, 1,      0, 0, 0, 0, 0}},
		{0, -1,  {W, 0, H - 1, 1, 0,  0, 0, 0, 0, 0}},
		{-1, 0,  {H, W - 1, 0, 0, 1,  0, 0, 0, 0, 0}},
		{1, 1,   {W, 0, 0, 1, 0,      H - 1, 0, 1, 0, 1}},
		{-1, 1,  {W - 1, 0, 0, 1, 0,  H, W - 1, 0, 0, 1}},
		{1, -1,  {W - 1, 1, H - 1, 1, 0,  H, 0, 0, 0, 1}},
		{-1, -1, {W, 0, H - 1, 1, 0,  H - 1, W - 1, 0, 0, 1}},
	};
	const int horizFirst[8] = {1, 3, 0, 2, 4, 5, 6, 7}, vertFirst[8] = {0, 2, 1, 3, 4, 5, 6, 7};
	const int* ord = W >= H ? horizFirst : vertFirst;
	SGMDirs sd; memset(&sd, 0, sizeof(sd));
	int total = 0;                                                     // in workgroups: PW lines each
	for (int i = 0; i < 8; ++i) {
		const Dir& d = dirs[ord[i]];
		sd.dx[i] = d.dx; sd.dy[i] = d.dy; sd.ln[i] = d.ln; sd.first[i] = total;
		total += (d.ln.nA + d.ln.nB + PW - 1) / PW;
	}
	sd.first[8] = total;
	evB(e, 1);
	if (total > 0) {
#define SGM_LAUNCH_SUB(MD_, DL_) hipLaunchKernelGGL((sgm_path_sub_kernel<LP, MD_, DL_>), dim3(total), dim3(64), 0, e->stream, e->d_grayL, e->w, e->vw, e->vh, e->d_pixels, e->d_costs, (unsigned*)e->d_accums, e->d_P2s, (int)P1, sd, e->d_deltas, (unsigned long long)e->numCosts)
		if (e->maxNumDisp <= 64) { if (delta) SGM_LAUNCH_SUB(64, true); else SGM_LAUNCH_SUB(64, false); }
		else { if (delta) SGM_LAUNCH_SUB(256, true); else SGM_LAUNCH_SUB(256, false); }
#undef SGM_LAUNCH_SUB
	}
	if (e->statsOn) e->stats.aggrLaunches += 1;
	evE(e);
	evB(e, 2);
	if (delta) hipLaunchKernelGGL(sgm_sum_wta_kernel, dim3((unsigned)((nPix + 15) / 16)), dim3(256), 0, e->stream, e->d_pixels, e->d_costs, e->d_deltas, (unsigned long long)e->numCosts, e->d_accums, nPix, e->d_disp, e->d_cost);
	else hipLaunchKernelGGL((sgm_wta_sub_kernel<LP>), dim3((unsigned)((nPix + 4 * PW - 1) / (4 * PW))), dim3(256), 0, e->stream, e->d_pixels, e->d_accums, nPix, e->d_disp, e->d_cost);
	evE(e);
	SGMCHK(e, hipGetLastError());
	if (e->statsOn) e->stats.calls += 1;
	return 0;
}

} // extern "C++"

static int sgmMatchSub(sgmhip_engine* e, uint16_t P1) {
	switch (e->subGroups) {
	case 8: return sgmMatchSubT<8>(e, P1);
	case 32: return sgmMatchSubT<32>(e, P1);
	default: return sgmMatchSubT<16>(e, P1);
	}
}

static int sgmMatch(sgmhip_engine* e, uint16_t P1) {
	if (e->subGroups) return sgmMatchSub(e, P1);
	const long nPix = (long)e->vw * e->vh;
	const int W = e->vw, H = e->vh;
	evB(e, 0);
	static const bool pxCost = [] { const char* v = getenv("SGMHIP_COST_PX"); return !v || atoi(v) != 0; }();   // 0: the wave-per-pixel-pair cost kernel
	static const bool uniCost = [] { const char* v = getenv("SGMHIP_COST_UNI"); return !v || atoi(v) != 0; }();   // 0: the sliding-window kernel for uniform ranges too
	if (pxCost) {
		// one pixel per lane, 64-pixel tiles of a row per wave (does the left-window prologue itself); with one range for all pixels the right-image strip of a tile sits in LDS
		const long nTiles = (long)((W + 63) / 64) * H;
		const dim3 g((unsigned)((nTiles + 3) / 4));
		if (e->uniform && uniCost) {
#define SGM_LAUNCH_UNI(MD_) hipLaunchKernelGGL((sgm_cost_uni_kernel<MD_>), g, dim3(256), 0, e->stream, e->d_color, e->d_grayL, e->d_grayR, e->w, e->h, W, H, e->uniformMin, e->maxNumDisp, e->d_costs)
			if (e->maxNumDisp <= 64) SGM_LAUNCH_UNI(64); else if (e->maxNumDisp <= 128) SGM_LAUNCH_UNI(128); else SGM_LAUNCH_UNI(256);
#undef SGM_LAUNCH_UNI
		} else hipLaunchKernelGGL(sgm_cost_px_kernel, g, dim3(256), 0, e->stream, e->d_color, e->d_grayL, e->d_grayR, e->w, e->h, W, H, e->d_pixels, e->d_costs);
	} else {
		hipLaunchKernelGGL(sgm_setup_kernel, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, e->stream, e->d_color, e->d_grayL, e->w, W, H, e->d_pixels, e->d_setup);
		const long nPairs = (long)((W + 1) / 2) * H;
		hipLaunchKernelGGL(sgm_cost_kernel, dim3((unsigned)((nPairs + 3) / 4)), dim3(256), 0, e->stream, e->d_color, e->d_grayL, e->d_grayR, e->w, e->h, W, H, e->d_pixels, e->d_setup, e->d_costs);
	}
	evE(e);
	const int NK = e->maxNumDisp <= 64 ? 1 : (e->maxNumDisp <= 128 ? 2 : 4);
	// DELTA aggregation: penalties that fit a byte (L - C <= P2); 8 scratch bytes per entry.  Uniform ranges: the register-resident kernel (NK <= 2); ragged ranges
	// (round 4): sgm_path_kernel<NK, true> -- one coalesced byte store per lane and step instead of an atomic add into the shared u16 sums
	static const int allowDelta = [] { const char* v = getenv("SGMHIP_DELTA"); return v ? atoi(v) : 3; }();   // bit 0: uniform ranges, bit 1: ragged ranges; 0: atomic u16 sums
	bool delta = e->maxP2 <= 255 && ((e->uniform && NK <= 2) ? (allowDelta & 1) != 0 : (allowDelta & 2) != 0);
	if (delta && !ensureDeltas(e)) delta = false;   // no room: the atomic path
	if (!delta) SGMCHK(e, hipMemsetAsync(e->d_accums, 0, (e->numCosts + 1) / 2 * 4, e->stream)); // imageAccumCosts.Memset(0), :990
	// the eight paths with the threaded variant's start sets, SemiGlobalMatcher.cpp:1083-1200
	struct Dir { int dx, dy; SGMLines ln; } dirs[8] = {
		{0, 1,   {W, 0, 0, 1, 0,      0, 0, 0, 0, 0}},            // width-down
		{1, 0,   {H, 0, 0, 0, 1,      0, 0, 0, 0, 0}},            // height-right
		{0, -1,  {W, 0, H - 1, 1, 0,  0, 0, 0, 0, 0}},            // width-up
		{-1, 0,  {H, W - 1, 0, 0, 1,  0, 0, 0, 0, 0}},            // height-left
		{1, 1,   {W, 0, 0, 1, 0,      H - 1, 0, 1, 0, 1}},        // right-down: top row, then left column y >= 1
		{-1, 1,  {W - 1, 0, 0, 1, 0,  H, W - 1, 0, 0, 1}},        // left-down: top row x < W-1, then right column
		{1, -1,  {W - 1, 1, H - 1, 1, 0,  H, 0, 0, 0, 1}},        // right-up: bottom row x >= 1, then left column
		{-1, -1, {W, 0, H - 1, 1, 0,  H - 1, W - 1, 0, 0, 1}},    // left-up: bottom row, then right column y <= H-2
	};
	// one grid for all of them, the directions with the longest lines first (their chains bound the kernel's duration)
	const int horizFirst[8] = {1, 3, 0, 2, 4, 5, 6, 7}, vertFirst[8] = {0, 2, 1, 3, 4, 5, 6, 7};
	const int* ord = W >= H ? horizFirst : vertFirst;
	SGMDirs sd; memset(&sd, 0, sizeof(sd));
	int total = 0;
	for (int i = 0; i < 8; ++i) {
		const Dir& d = dirs[ord[i]];
		sd.dx[i] = d.dx; sd.dy[i] = d.dy; sd.ln[i] = d.ln; sd.first[i] = total;
		total += d.ln.nA + d.ln.nB;
	}
	sd.first[8] = total;
	evB(e, 1);
	if (total > 0) {
		if (e->uniform && NK <= 2) {
			const int align = e->maxNumDisp % 2 == 0 ? 2 : 1;
#define SGM_LAUNCH_UNIFORM(NK_, AL_, DL_) hipLaunchKernelGGL((sgm_path_uniform_kernel<NK_, AL_, DL_, false>), dim3(total), dim3(64), 0, e->stream, e->d_grayL, e->w, e->vw, e->vh, e->maxNumDisp, e->d_costs, (unsigned*)e->d_accums, e->d_P2s, (int)P1, sd, e->d_deltas, (unsigned long long)e->numCosts)
			static const int stageMaxNK = [] { const char* v = getenv("SGMHIP_STAGE"); return v ? atoi(v) : 2; }();   // stage the delta bytes for up to this many entries per lane (0: never)
			if (delta && e->maxNumDisp % 16 == 0 && NK <= stageMaxNK) {   // delta bytes staged through LDS, written out 16 bytes at a time
#define SGM_LAUNCH_STAGED(NK_) hipLaunchKernelGGL((sgm_path_uniform_kernel<NK_, 2, true, true>), dim3(total), dim3(64), 0, e->stream, e->d_grayL, e->w, e->vw, e->vh, e->maxNumDisp, e->d_costs, (unsigned*)e->d_accums, e->d_P2s, (int)P1, sd, e->d_deltas, (unsigned long long)e->numCosts)
				if (NK == 1) SGM_LAUNCH_STAGED(1); else SGM_LAUNCH_STAGED(2);
#undef SGM_LAUNCH_STAGED
			} else if (delta) {
				if (NK == 1) { if (align == 2) SGM_LAUNCH_UNIFORM(1, 2, true); else SGM_LAUNCH_UNIFORM(1, 1, true); }
				else { if (align == 2) SGM_LAUNCH_UNIFORM(2, 2, true); else SGM_LAUNCH_UNIFORM(2, 1, true); }
			} else {
				if (NK == 1) { if (align == 2) SGM_LAUNCH_UNIFORM(1, 2, false); else SGM_LAUNCH_UNIFORM(1, 1, false); }
				else { if (align == 2) SGM_LAUNCH_UNIFORM(2, 2, false); else SGM_LAUNCH_UNIFORM(2, 1, false); }
			}
#undef SGM_LAUNCH_UNIFORM
		}
		else launchPath(e, e->stream, NK, total, (int)P1, sd, delta);
	}
	if (e->statsOn) e->stats.aggrLaunches += 1;
	evE(e);
	evB(e, 2);
	if (delta) hipLaunchKernelGGL(sgm_sum_wta_kernel, dim3((unsigned)((nPix + 15) / 16)), dim3(256), 0, e->stream, e->d_pixels, e->d_costs, e->d_deltas, (unsigned long long)e->numCosts, e->d_accums, nPix, e->d_disp, e->d_cost);
	else {
		// winner-take-all: a wavefront per pixel spends most of its instructions on the 6-step reduction of 64 lanes; eight pixels per
		// wavefront with a strided loop over the range need a fraction of the wave-instructions per pixel (same first minimum; SGMHIP_WTA_LANES =
		// 64 selects the one-pixel kernel again)
		static const int wtaLanes = [] { const char* v = getenv("SGMHIP_WTA_LANES"); const int n = v ? atoi(v) : 8; return (n == 8 || n == 16 || n == 32 || n == 64) ? n : 8; }();   // measured: 0.72 / 0.33 / 0.20 / 0.14 ms at 64 / 32 / 16 / 8 lanes (profiles/r02_sgm_wta_lanes.log)
		if (wtaLanes == 8) hipLaunchKernelGGL((sgm_wta_sub_kernel<8>), dim3((unsigned)((nPix + 31) / 32)), dim3(256), 0, e->stream, e->d_pixels, e->d_accums, nPix, e->d_disp, e->d_cost);
		else if (wtaLanes == 16) hipLaunchKernelGGL((sgm_wta_sub_kernel<16>), dim3((unsigned)((nPix + 15) / 16)), dim3(256), 0, e->stream, e->d_pixels, e->d_accums, nPix, e->d_disp, e->d_cost);
		else if (wtaLanes == 32) hipLaunchKernelGGL((sgm_wta_sub_kernel<32>), dim3((unsigned)((nPix + 7) / 8)), dim3(256), 0, e->stream, e->d_pixels, e->d_accums, nPix, e->d_disp, e->d_cost);
		else hipLaunchKernelGGL(sgm_wta_kernel, dim3((unsigned)((nPix + 3) / 4)), dim3(256), 0, e->stream, e->d_pixels, e->d_accums, nPix, e->d_disp, e->d_cost);
	}
	evE(e);
	SGMCHK(e, hipGetLastError());
	if (e->statsOn) e->stats.calls += 1;
	return 0;
}

int sgmhip_get_results(sgmhip_engine* e, int16_t* disparity, uint16_t* cost, uint8_t* costs, uint16_t* accums) {
	if (!e || e->numCosts == 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	const size_t nPix = (size_t)e->vw * e->vh;
	if (disparity) SGMCHK(e, hipMemcpyAsync(disparity, e->d_disp, nPix * 2, hipMemcpyDeviceToHost, e->stream));
	if (cost) SGMCHK(e, hipMemcpyAsync(cost, e->d_cost, nPix * 2, hipMemcpyDeviceToHost, e->stream));
	if (costs) SGMCHK(e, hipMemcpyAsync(costs, e->d_costs, e->numCosts, hipMemcpyDeviceToHost, e->stream));
	if (accums) SGMCHK(e, hipMemcpyAsync(accums, e->d_accums, e->numCosts * 2, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}
int sgmhip_set_sub_group_kernels(sgmhip_engine* e, int lanes) {
	if (!e || (lanes != 0 && lanes != 1 && lanes != 8 && lanes != 16 && lanes != 32)) return SGMHIP_E_ARG;
	e->subGroups = lanes == 1 ? 16 : lanes;
	return 0;
}
int sgmhip_sync(sgmhip_engine* e) { if (!e) return SGMHIP_E_ARG; SGMCHK(e, hipSetDevice(e->device)); SGMCHK(e, hipStreamSynchronize(e->stream)); return 0; }
int sgmhip_stats_reset(sgmhip_engine* e, int enable) { if (!e) return SGMHIP_E_ARG; hipSetDevice(e->device); sgmCollect(e); memset(&e->stats, 0, sizeof(e->stats)); e->statsOn = enable != 0; return 0; }
int sgmhip_stats_get(sgmhip_engine* e, SGMHipStats* out) { if (!e || !out) return SGMHIP_E_ARG; hipSetDevice(e->device); int rc = sgmCollect(e); if (rc) return rc; *out = e->stats; return 0; }

// ---- tSGM steps around Match (SemiGlobalMatcher.cpp:1449-1811), see csrc/sgm_post.h ---------------------------------------------
namespace {
struct DevBuf {   // scoped device allocation for the stateless helpers
	void* p = nullptr;
	~DevBuf() { if (p) hipFree(p); }
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
inline unsigned gridFor(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 65535); }
}

int sgmhip_consistency_cross_check(sgmhip_engine* e, int16_t* l2r, const int16_t* r2l, int wl, int h, int wr, int thCross) {
	if (!e || !l2r || !r2l || wl <= 0 || wr <= 0 || h <= 0 || thCross < 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, b; const size_t nl = (size_t)wl * h, nr = (size_t)wr * h;
	SGMCHK(e, a.alloc(nl * 2)); SGMCHK(e, b.alloc(nr * 2));
	SGMCHK(e, hipMemcpyAsync(a.p, l2r, nl * 2, hipMemcpyHostToDevice, e->stream)); SGMCHK(e, hipMemcpyAsync(b.p, r2l, nr * 2, hipMemcpyHostToDevice, e->stream));
	hipLaunchKernelGGL(sgmp_cross_check_kernel, dim3(gridFor(nl)), dim3(256), 0, e->stream, (int16_t*)a.p, (const int16_t*)b.p, wl, wr, h, thCross);
	SGMCHK(e, hipMemcpyAsync(l2r, a.p, nl * 2, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_filter_by_cost(sgmhip_engine* e, int16_t* disparity, const uint16_t* cost, int w, int h, uint16_t th) {
	if (!e || !disparity || !cost || w <= 0 || h <= 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, b; const size_t n = (size_t)w * h;
	SGMCHK(e, a.alloc(n * 2)); SGMCHK(e, b.alloc(n * 2));
	SGMCHK(e, hipMemcpyAsync(a.p, disparity, n * 2, hipMemcpyHostToDevice, e->stream)); SGMCHK(e, hipMemcpyAsync(b.p, cost, n * 2, hipMemcpyHostToDevice, e->stream));
	hipLaunchKernelGGL(sgmp_filter_by_cost_kernel, dim3(gridFor(n)), dim3(256), 0, e->stream, (int16_t*)a.p, (const uint16_t*)b.p, n, th);
	SGMCHK(e, hipMemcpyAsync(disparity, a.p, n * 2, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_extract_mask(sgmhip_engine* e, const int16_t* disparity, uint8_t* mask, int w, int h, int thValid, int initValid) {
	if (!e || !disparity || !mask || w <= 0 || h <= 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, b; const size_t n = (size_t)w * h;
	SGMCHK(e, a.alloc(n * 2)); SGMCHK(e, b.alloc(n));
	SGMCHK(e, hipMemcpyAsync(a.p, disparity, n * 2, hipMemcpyHostToDevice, e->stream));
	if (initValid) SGMCHK(e, hipMemsetAsync(b.p, 0xFF, n, e->stream));      // maskMap.create(size); setTo(VALID), :1521-1524
	else SGMCHK(e, hipMemcpyAsync(b.p, mask, n, hipMemcpyHostToDevice, e->stream));
	hipLaunchKernelGGL(sgmp_extract_mask_kernel, dim3((h + 63) / 64), dim3(64), 0, e->stream, (const int16_t*)a.p, (uint8_t*)b.p, w, h, thValid);
	SGMCHK(e, hipMemcpyAsync(mask, b.p, n, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_upscale_mask(sgmhip_engine* e, const uint8_t* mask, int w, int h, uint8_t* mask2x, int w2, int h2) {
	if (!e || !mask || !mask2x || w <= 0 || h <= 0 || w2 <= 0 || h2 <= 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, b; const size_t n = (size_t)w * h, n2 = (size_t)w2 * h2;
	SGMCHK(e, a.alloc(n)); SGMCHK(e, b.alloc(n2));
	SGMCHK(e, hipMemcpyAsync(a.p, mask, n, hipMemcpyHostToDevice, e->stream));
	hipLaunchKernelGGL(sgmp_upscale_mask_kernel, dim3(gridFor(n2)), dim3(256), 0, e->stream, (const uint8_t*)a.p, w, h, (uint8_t*)b.p, w2, h2);
	SGMCHK(e, hipMemcpyAsync(mask2x, b.p, n2, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_flip_direction(sgmhip_engine* e, const int16_t* l2r, int w, int h, int16_t* r2l) {
	if (!e || !l2r || !r2l || w <= 0 || h <= 0 || w > 65534) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, k, b; const size_t n = (size_t)w * h;
	SGMCHK(e, a.alloc(n * 2)); SGMCHK(e, k.alloc(n * 4)); SGMCHK(e, b.alloc(n * 2));
	SGMCHK(e, hipMemcpyAsync(a.p, l2r, n * 2, hipMemcpyHostToDevice, e->stream));
	SGMCHK(e, hipMemsetAsync(k.p, 0, n * 4, e->stream));
	hipLaunchKernelGGL(sgmp_flip_scatter_kernel, dim3(gridFor(n)), dim3(256), 0, e->stream, (const int16_t*)a.p, (uint32_t*)k.p, w, h);
	hipLaunchKernelGGL(sgmp_flip_decode_kernel, dim3(gridFor(n)), dim3(256), 0, e->stream, (const uint32_t*)k.p, (int16_t*)b.p, n);
	SGMCHK(e, hipMemcpyAsync(r2l, b.p, n * 2, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_refine_disparity(sgmhip_engine* e, int subpixelMode, int subpixelSteps) {
	if (!e || e->numCosts == 0 || subpixelMode < 0 || subpixelMode > SGMP_SUBPIXEL_LC_BLEND || subpixelSteps < 0 || subpixelSteps > 64) return SGMHIP_E_ARG;
	if (subpixelSteps <= 1) return 0;                                     // :1696-1697
	SGMCHK(e, hipSetDevice(e->device));
	const long nPix = (long)e->vw * e->vh;
	hipLaunchKernelGGL(sgmp_refine_kernel, dim3(gridFor((size_t)nPix)), dim3(256), 0, e->stream, e->d_pixels, e->d_accums, nPix, e->d_disp, subpixelMode, subpixelSteps);
	SGMCHK(e, hipGetLastError());
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_disparity2range_map(sgmhip_engine* e, const int16_t* disparity, int w, int h, const uint8_t* mask2x, int w2, int h2,
		int minNumDisp, int minNumDispInvalid, SGMHipPixelData* pixels, uint64_t* numCosts, int* maxNumDisp) {
	if (!e || !disparity || !mask2x || !pixels || w <= 0 || h <= 0 || w2 <= SGM_HW + 2 * w || h2 < SGM_HW + 2 * h) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, m, r; const size_t n = (size_t)w * h, n2 = (size_t)w2 * h2;
	SGMCHK(e, a.alloc(n * 2)); SGMCHK(e, m.alloc(n2)); SGMCHK(e, r.alloc(n * 4));
	SGMCHK(e, hipMemcpyAsync(a.p, disparity, n * 2, hipMemcpyHostToDevice, e->stream)); SGMCHK(e, hipMemcpyAsync(m.p, mask2x, n2, hipMemcpyHostToDevice, e->stream));
	hipLaunchKernelGGL(sgmp_range_kernel, dim3(gridFor(n)), dim3(256), 0, e->stream, (const int16_t*)a.p, w, h, (const uint8_t*)m.p, w2, minNumDisp, minNumDispInvalid, (short2*)r.p);
	std::vector<int16_t> rg(n * 2);
	SGMCHK(e, hipMemcpyAsync(rg.data(), r.p, n * 4, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	// expansion to the 2x pixel table in raster order (:1409-1441): 2x pixel (R, C) takes the range of low-resolution pixel
	// (R < HW+2 ? 0 : min((R-HW)/2, h-1), likewise for C); idx is the running sum of numDisp
	uint64_t total = 0; int mx = 0;
	for (int R = 0; R < h2; ++R) {
		const int rr = R < SGM_HW + 2 ? 0 : std::min((R - SGM_HW) / 2, h - 1);
		for (int Cc = 0; Cc < w2; ++Cc) {
			const int cc = Cc < SGM_HW + 2 ? 0 : std::min((Cc - SGM_HW) / 2, w - 1);
			const int16_t lo = rg[((size_t)rr * w + cc) * 2], hi = rg[((size_t)rr * w + cc) * 2 + 1];
			SGMHipPixelData& px = pixels[(size_t)R * w2 + Cc];
			px.idx = total; px.minDisp = lo; px.maxDisp = hi;
			const int nd = (int16_t)(hi - lo);
			total += (uint64_t)(int64_t)nd;
			if (nd > mx) mx = nd;
		}
	}
	if (numCosts) *numCosts = total;
	if (maxNumDisp) *maxNumDisp = mx;
	return 0;
}

int sgmhip_depth2disparity_map(sgmhip_engine* e, const float* depthMap, int dw, int dh, const double invH[9], const double invQ[16], int subpixelSteps,
		int16_t* disparity, int w, int h) {
	if (!e || !depthMap || !invH || !invQ || !disparity || dw <= 0 || dh <= 0 || w <= 0 || h <= 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, b; const size_t nd = (size_t)dw * dh, n = (size_t)w * h;
	SGMCHK(e, a.alloc(nd * 4)); SGMCHK(e, b.alloc(n * 2));
	SGMCHK(e, hipMemcpyAsync(a.p, depthMap, nd * 4, hipMemcpyHostToDevice, e->stream));
	SGMPMat mh{}, mq{}; memcpy(mh.m, invH, 72); memcpy(mq.m, invQ, 128);
	hipLaunchKernelGGL(sgmp_depth2disparity_kernel, dim3(gridFor(n)), dim3(256), 0, e->stream, (const float*)a.p, dw, dh, mh, mq, subpixelSteps, (int16_t*)b.p, w, h);
	SGMCHK(e, hipMemcpyAsync(disparity, b.p, n * 2, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_disparity2depth_map(sgmhip_engine* e, const int16_t* disparity, const uint16_t* cost, int w, int h, const double H[9], const double Q[16],
		int subpixelSteps, float* depthMap, float* confMap, int dw, int dh) {
	if (!e || !disparity || !H || !Q || !depthMap || (cost && !confMap) || dw <= 0 || dh <= 0 || w <= 0 || h <= 0 || subpixelSteps <= 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, c, d, f; const size_t n = (size_t)w * h, nd = (size_t)dw * dh;
	SGMCHK(e, a.alloc(n * 2)); SGMCHK(e, c.alloc(n * 2)); SGMCHK(e, d.alloc(nd * 4)); SGMCHK(e, f.alloc(nd * 4));
	SGMCHK(e, hipMemcpyAsync(a.p, disparity, n * 2, hipMemcpyHostToDevice, e->stream));
	if (cost) SGMCHK(e, hipMemcpyAsync(c.p, cost, n * 2, hipMemcpyHostToDevice, e->stream));
	SGMPMat mh{}, mq{}; memcpy(mh.m, H, 72); memcpy(mq.m, Q, 128);
	hipLaunchKernelGGL(sgmp_disparity2depth_kernel, dim3(gridFor(nd)), dim3(256), 0, e->stream, (const int16_t*)a.p, cost ? (const uint16_t*)c.p : nullptr, w, h, mh, mq, subpixelSteps, (float*)d.p, (float*)f.p, dw, dh);
	SGMCHK(e, hipMemcpyAsync(depthMap, d.p, nd * 4, hipMemcpyDeviceToHost, e->stream));
	if (cost) SGMCHK(e, hipMemcpyAsync(confMap, f.p, nd * 4, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_project_disparity2depth_map(sgmhip_engine* e, const int16_t* disparity, const uint16_t* cost, int w, int h, const double Q[16], int subpixelSteps,
		float* depthMap, float* depthRangeMap, float* confMap, int dw, int dh, int* anyDepth) {
	if (!e || !disparity || !Q || !depthMap || !depthRangeMap || (cost && !confMap) || w <= 0 || h <= 0 || dw <= 0 || dh <= 0 || subpixelSteps <= 0 || (size_t)w * h > 0xFFFFFFFFull) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, c, k, d, rg, cf, cnt; const size_t n = (size_t)w * h, nd = (size_t)dw * dh;
	SGMCHK(e, a.alloc(n * 2)); SGMCHK(e, c.alloc(n * 2)); SGMCHK(e, k.alloc(nd * 4 * 8)); SGMCHK(e, d.alloc(nd * 4)); SGMCHK(e, rg.alloc(nd * 8)); SGMCHK(e, cf.alloc(nd * 4)); SGMCHK(e, cnt.alloc(4));
	SGMCHK(e, hipMemcpyAsync(a.p, disparity, n * 2, hipMemcpyHostToDevice, e->stream));
	if (cost) SGMCHK(e, hipMemcpyAsync(c.p, cost, n * 2, hipMemcpyHostToDevice, e->stream));
	SGMCHK(e, hipMemsetAsync(cnt.p, 0, 4, e->stream)); SGMCHK(e, hipMemsetAsync(rg.p, 0, nd * 8, e->stream));
	SGMPMat mq{}; memcpy(mq.m, Q, 128);
	const uint16_t* dc = cost ? (const uint16_t*)c.p : nullptr;
	hipLaunchKernelGGL(sgmp_fill_u64, dim3(gridFor(nd * 4)), dim3(256), 0, e->stream, (unsigned long long*)k.p, nd * 4, SGMP_KEY_NONE);
	hipLaunchKernelGGL(sgmp_proj_splat_kernel, dim3(gridFor(n)), dim3(256), 0, e->stream, (const int16_t*)a.p, dc, w, h, mq, subpixelSteps, (unsigned long long*)k.p, dw, dh);
	hipLaunchKernelGGL(sgmp_proj_resolve_kernel, dim3(gridFor(nd)), dim3(256), 0, e->stream, (const int16_t*)a.p, dc, w, mq, subpixelSteps, (const unsigned long long*)k.p, dw, dh,
		(float*)d.p, (float*)rg.p, cost ? (float*)cf.p : nullptr, (unsigned*)cnt.p);
	unsigned num = 0;
	SGMCHK(e, hipMemcpyAsync(depthMap, d.p, nd * 4, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipMemcpyAsync(depthRangeMap, rg.p, nd * 8, hipMemcpyDeviceToHost, e->stream));
	if (cost) SGMCHK(e, hipMemcpyAsync(confMap, cf.p, nd * 4, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipMemcpyAsync(&num, cnt.p, 4, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	if (anyDepth) *anyDepth = num > 0 ? 1 : 0;
	return 0;
}

int sgmhip_fuse_pairs(sgmhip_engine* e, const float* const* depthMaps, const float* const* depthRangeMaps, const float* const* confMaps, int nPairs, int dw, int dh,
		unsigned minViews, float* depthMap, float* confMap) {
	if (!e || !depthMaps || !depthRangeMaps || !confMaps || nPairs < 0 || nPairs > SGMP_MAX_PAIRS || dw <= 0 || dh <= 0 || !depthMap || !confMap) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	const size_t nd = (size_t)dw * dh;
	std::vector<DevBuf> bufs((size_t)nPairs * 3);
	SGMPPairs pr{};
	for (int p = 0; p < nPairs; ++p) {
		if (!depthMaps[p] || !depthRangeMaps[p] || !confMaps[p]) return SGMHIP_E_ARG;
		SGMCHK(e, bufs[p * 3].alloc(nd * 4)); SGMCHK(e, bufs[p * 3 + 1].alloc(nd * 8)); SGMCHK(e, bufs[p * 3 + 2].alloc(nd * 4));
		SGMCHK(e, hipMemcpyAsync(bufs[p * 3].p, depthMaps[p], nd * 4, hipMemcpyHostToDevice, e->stream));
		SGMCHK(e, hipMemcpyAsync(bufs[p * 3 + 1].p, depthRangeMaps[p], nd * 8, hipMemcpyHostToDevice, e->stream));
		SGMCHK(e, hipMemcpyAsync(bufs[p * 3 + 2].p, confMaps[p], nd * 4, hipMemcpyHostToDevice, e->stream));
		pr.depth[p] = (const float*)bufs[p * 3].p; pr.range[p] = (const float*)bufs[p * 3 + 1].p; pr.conf[p] = (const float*)bufs[p * 3 + 2].p;
	}
	DevBuf d, c; SGMCHK(e, d.alloc(nd * 4)); SGMCHK(e, c.alloc(nd * 4));
	hipLaunchKernelGGL(sgmp_fuse_pairs_kernel, dim3(gridFor(nd)), dim3(256), 0, e->stream, pr, nPairs, nd, minViews, (float*)d.p, (float*)c.p);
	SGMCHK(e, hipMemcpyAsync(depthMap, d.p, nd * 4, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipMemcpyAsync(confMap, c.p, nd * 4, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

// SemiGlobalMatcher::Fuse (:738-859) for pairs that have the reference image on the left, resident: ProjectDisparity2DepthMap of every pair, pairs that
// produce no depth are dropped (:779-782), then the per-pixel cluster fusion -- the per-pair maps never leave the device.
int sgmhip_fuse_disparities(sgmhip_engine* e, int nPairs, const int16_t* const* disparities, const uint16_t* const* costs, const int* widths, const int* heights,
		const double* Qs, const int* subpixelSteps, int dw, int dh, unsigned minViews, float* depthMap, float* confMap, int* nUsed) {
	if (!e || nPairs < 0 || nPairs > SGMP_MAX_PAIRS || (nPairs && (!disparities || !costs || !widths || !heights || !Qs || !subpixelSteps)) || dw <= 0 || dh <= 0 || !depthMap || !confMap) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	hipStream_t st = e->stream;
	const size_t nd = (size_t)dw * dh;
	std::vector<DevBuf> maps((size_t)nPairs * 3);
	DevBuf k, cnt, d, c;
	SGMCHK(e, k.alloc(nd * 4 * 8)); SGMCHK(e, cnt.alloc(4 * (size_t)std::max(nPairs, 1))); SGMCHK(e, d.alloc(nd * 4)); SGMCHK(e, c.alloc(nd * 4));
	SGMCHK(e, hipMemsetAsync(cnt.p, 0, 4 * (size_t)std::max(nPairs, 1), st));
	std::vector<DevBuf> in((size_t)nPairs * 2);
	for (int p = 0; p < nPairs; ++p) {
		const int w = widths[p], h = heights[p];
		if (!disparities[p] || !costs[p] || w <= 0 || h <= 0 || subpixelSteps[p] <= 0 || (size_t)w * h > 0xFFFFFFFFull) return SGMHIP_E_ARG;
		const size_t n = (size_t)w * h;
		SGMCHK(e, in[p * 2].alloc(n * 2)); SGMCHK(e, in[p * 2 + 1].alloc(n * 2));
		SGMCHK(e, maps[p * 3].alloc(nd * 4)); SGMCHK(e, maps[p * 3 + 1].alloc(nd * 8)); SGMCHK(e, maps[p * 3 + 2].alloc(nd * 4));
		SGMCHK(e, hipMemcpyAsync(in[p * 2].p, disparities[p], n * 2, hipMemcpyHostToDevice, st)); SGMCHK(e, hipMemcpyAsync(in[p * 2 + 1].p, costs[p], n * 2, hipMemcpyHostToDevice, st));
		SGMCHK(e, hipMemsetAsync(maps[p * 3 + 1].p, 0, nd * 8, st));
		SGMPMat mq{}; memcpy(mq.m, Qs + 16 * p, 128);
		hipLaunchKernelGGL(sgmp_fill_u64, dim3(gridFor(nd * 4)), dim3(256), 0, st, (unsigned long long*)k.p, nd * 4, SGMP_KEY_NONE);
		hipLaunchKernelGGL(sgmp_proj_splat_kernel, dim3(gridFor(n)), dim3(256), 0, st, (const int16_t*)in[p * 2].p, (const uint16_t*)in[p * 2 + 1].p, w, h, mq, subpixelSteps[p], (unsigned long long*)k.p, dw, dh);
		hipLaunchKernelGGL(sgmp_proj_resolve_kernel, dim3(gridFor(nd)), dim3(256), 0, st, (const int16_t*)in[p * 2].p, (const uint16_t*)in[p * 2 + 1].p, w, mq, subpixelSteps[p],
			(const unsigned long long*)k.p, dw, dh, (float*)maps[p * 3].p, (float*)maps[p * 3 + 1].p, (float*)maps[p * 3 + 2].p, (unsigned*)cnt.p + p);
	}
	std::vector<unsigned> num((size_t)std::max(nPairs, 1), 0u);
	if (nPairs) SGMCHK(e, hipMemcpyAsync(num.data(), cnt.p, 4 * (size_t)nPairs, hipMemcpyDeviceToHost, st));
	SGMCHK(e, hipStreamSynchronize(st));
	SGMPPairs pr{}; int used = 0;
	for (int p = 0; p < nPairs; ++p) if (num[p] > 0) { pr.depth[used] = (const float*)maps[p * 3].p; pr.range[used] = (const float*)maps[p * 3 + 1].p; pr.conf[used] = (const float*)maps[p * 3 + 2].p; ++used; }
	if (nUsed) *nUsed = used;
	if (used == 0) { memset(depthMap, 0, nd * 4); memset(confMap, 0, nd * 4); return 0; }
	hipLaunchKernelGGL(sgmp_fuse_pairs_kernel, dim3(gridFor(nd)), dim3(256), 0, st, pr, used, nd, minViews, (float*)d.p, (float*)c.p);
	SGMCHK(e, hipGetLastError());
	SGMCHK(e, hipMemcpyAsync(depthMap, d.p, nd * 4, hipMemcpyDeviceToHost, st));
	SGMCHK(e, hipMemcpyAsync(confMap, c.p, nd * 4, hipMemcpyDeviceToHost, st));
	SGMCHK(e, hipStreamSynchronize(st));
	return 0;
}

int sgmhip_filter_speckles(sgmhip_engine* e, int16_t* disparity, int w, int h, int maxSpeckleSize, int maxDiff) {
	if (!e || !disparity || w <= 0 || h <= 0 || maxSpeckleSize < 0 || maxDiff < 0 || (size_t)w * h > 0x7fffffffull) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	DevBuf a, p, z; const size_t n = (size_t)w * h;
	SGMCHK(e, a.alloc(n * 2)); SGMCHK(e, p.alloc(n * 4)); SGMCHK(e, z.alloc(n * 4));
	SGMCHK(e, hipMemcpyAsync(a.p, disparity, n * 2, hipMemcpyHostToDevice, e->stream));
	const unsigned g = gridFor(n);
	hipLaunchKernelGGL(sgmp_speckle_init_kernel, dim3(g), dim3(256), 0, e->stream, (int*)p.p, (int*)z.p, (int)n);
	hipLaunchKernelGGL(sgmp_speckle_hook_kernel, dim3(g), dim3(256), 0, e->stream, (const int16_t*)a.p, (int*)p.p, w, h, maxDiff);
	hipLaunchKernelGGL(sgmp_speckle_flatten_kernel, dim3(g), dim3(256), 0, e->stream, (int*)p.p, (int*)z.p, (int)n);
	hipLaunchKernelGGL(sgmp_speckle_apply_kernel, dim3(g), dim3(256), 0, e->stream, (int16_t*)a.p, (const int*)p.p, (const int*)z.p, (int)n, maxSpeckleSize);
	SGMCHK(e, hipMemcpyAsync(disparity, a.p, n * 2, hipMemcpyDeviceToHost, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

int sgmhip_set_disparity(sgmhip_engine* e, const int16_t* disparity) {
	if (!e || !disparity || e->numCosts == 0) return SGMHIP_E_ARG;
	SGMCHK(e, hipSetDevice(e->device));
	SGMCHK(e, hipMemcpyAsync(e->d_disp, disparity, (size_t)e->vw * e->vh * 2, hipMemcpyHostToDevice, e->stream));
	SGMCHK(e, hipStreamSynchronize(e->stream));
	return 0;
}

// ---- the whole coarse-to-fine loop for one rectified pair, resident (SemiGlobalMatcher.cpp:577-706; kernels in sgm_tsgm.hip) -----------------
namespace {
// computeMaxResolution(w, h, level = 8, minResolution) (libs/Common/Types.inl:2459-2477) -> k with scale = 1 / max(2, 2^k); 0 = plain SGM
int tsgmLevels(int w, int h, unsigned minResolution) {
	if (!minResolution) return 0;
	const unsigned size0 = (unsigned)std::max(w, h);
	unsigned level = 8;
	if ((size0 >> level) < minResolution) { level = 0; while ((size0 >> (level + 1)) >= minResolution) ++level; }
	return (int)std::max(1u, level);
}
inline int cvRound(double v) { return (int)nearbyint(v); }      // cv::saturate_cast<int>(double): round half to even
}

int sgmhip_tsgm_match(sgmhip_engine* e, const uint8_t* leftBGR, const uint8_t* rightBGR, const float* leftGray, const float* rightGray,
		const uint8_t* leftMask, const uint8_t* rightMask, int w, int h, unsigned minResolution, const int16_t* initLeftDisparity,
		int nSpeckleSize, int subpixelMode, int subpixelSteps, uint16_t P1, const uint16_t P2s[256], int16_t* disparity, uint16_t* cost, int* numLevels) {
	if (!e || !leftBGR || !rightBGR || !leftGray || !rightGray || !leftMask || !rightMask || !P2s || !disparity || !cost || w <= 0 || h <= 0 || nSpeckleSize < 0 ||
	    subpixelMode < 0 || subpixelMode > SGMP_SUBPIXEL_LC_BLEND || subpixelSteps < 0 || subpixelSteps > 64) return SGMHIP_E_ARG;
	const int k = tsgmLevels(w, h, minResolution);
	if (k == 0) { e->err = "plain SGM (minResolution = 0) is not driven here: only tSGM"; return SGMHIP_E_ARG; }
	if ((w % (1 << k)) || (h % (1 << k)) || (w >> k) <= 2 * SGM_HW + 2 || (h >> k) <= 2 * SGM_HW + 2) { e->err = "image size must be a multiple of 2^levels and its coarsest level larger than the window"; return SGMHIP_E_ARG; }
	SGMCHK(e, hipSetDevice(e->device));
	hipStream_t st = e->stream;
	const size_t nFull = (size_t)w * h, nValid = (size_t)(w - 2 * SGM_HW) * (h - 2 * SGM_HW);
	// full-resolution inputs and the per-level working set (sized for the finest level)
	DevBuf fLB, fRB, fLG, fRG, fLM, fRM, lB, rB, lG, rG, lM, rM, lM2, rM2, lD, rD, lDn, rDn, keys, ranges, tiles, scal, par, siz;
	SGMCHK(e, fLB.alloc(nFull * 3)); SGMCHK(e, fRB.alloc(nFull * 3)); SGMCHK(e, fLG.alloc(nFull * 4)); SGMCHK(e, fRG.alloc(nFull * 4)); SGMCHK(e, fLM.alloc(nFull)); SGMCHK(e, fRM.alloc(nFull));
	SGMCHK(e, lB.alloc(nFull * 3)); SGMCHK(e, rB.alloc(nFull * 3)); SGMCHK(e, lG.alloc(nFull * 4)); SGMCHK(e, rG.alloc(nFull * 4));
	SGMCHK(e, lM.alloc(nValid)); SGMCHK(e, rM.alloc(nValid)); SGMCHK(e, lM2.alloc(nValid)); SGMCHK(e, rM2.alloc(nValid));
	SGMCHK(e, lD.alloc(nValid * 2)); SGMCHK(e, rD.alloc(nValid * 2)); SGMCHK(e, lDn.alloc(nValid * 2)); SGMCHK(e, rDn.alloc(nValid * 2));
	SGMCHK(e, keys.alloc(nValid * 4)); SGMCHK(e, ranges.alloc(nValid * 4)); SGMCHK(e, par.alloc(nValid * 4)); SGMCHK(e, siz.alloc(nValid * 4));
	const int maxTiles = (int)((nValid + SGMT_TILE - 1) / SGMT_TILE);
	SGMCHK(e, tiles.alloc((size_t)maxTiles * 8)); SGMCHK(e, scal.alloc(16));
	SGMCHK(e, hipMemcpyAsync(fLB.p, leftBGR, nFull * 3, hipMemcpyHostToDevice, st)); SGMCHK(e, hipMemcpyAsync(fRB.p, rightBGR, nFull * 3, hipMemcpyHostToDevice, st));
	SGMCHK(e, hipMemcpyAsync(fLG.p, leftGray, nFull * 4, hipMemcpyHostToDevice, st)); SGMCHK(e, hipMemcpyAsync(fRG.p, rightGray, nFull * 4, hipMemcpyHostToDevice, st));
	SGMCHK(e, hipMemcpyAsync(fLM.p, leftMask, nFull, hipMemcpyHostToDevice, st)); SGMCHK(e, hipMemcpyAsync(fRM.p, rightMask, nFull, hipMemcpyHostToDevice, st));
	SGMCHK(e, hipMemcpyAsync(e->d_P2s, P2s, 512, hipMemcpyHostToDevice, st));
	e->maxP2 = 0; for (int i = 0; i < 256; ++i) e->maxP2 = std::max(e->maxP2, (int)P2s[i]);

	int16_t* leftDisp = (int16_t*)lD.p; int16_t* rightDisp = (int16_t*)rD.p; int16_t* leftNew = (int16_t*)lDn.p; int16_t* rightNew = (int16_t*)rDn.p;
	uint8_t* lm = (uint8_t*)lM.p; uint8_t* rm = (uint8_t*)rM.p; uint8_t* lmNext = (uint8_t*)lM2.p; uint8_t* rmNext = (uint8_t*)rM2.p;
	int dW = 0, dH = 0;            // size of leftDisp (the previous level's valid grid)
	int mW = 0, mH = 0;            // size of the masks
	int levels = 0;
	bool first = true;
	// Disparity2RangeMap on the device + Match of (bgr, grayA -> grayB) with those ranges; the result is copied to `out`
	auto rangeAndMatch = [&](const int16_t* disp, const uint8_t* mask2x, int vw, int vh, int lw, int lh, const void* bgr, const void* gA, const void* gB, int a, int b, int16_t* out) -> int {
		const size_t n = (size_t)dW * dH, n2 = (size_t)vw * vh;
		hipLaunchKernelGGL(sgmp_range_kernel, dim3(gridFor(n)), dim3(256), 0, st, disp, dW, dH, mask2x, vw, a, b, (short2*)ranges.p);
		const int nT = (int)((n2 + SGMT_TILE - 1) / SGMT_TILE);
		SGMCHK(e, hipMemsetAsync(scal.p, 0, 16, st));
		hipLaunchKernelGGL(sgmt_tile_sums_kernel, dim3(nT), dim3(256), 0, st, (const short2*)ranges.p, dW, dH, vw, n2, (unsigned long long*)tiles.p, (int*)((char*)scal.p + 8));
		hipLaunchKernelGGL(sgmt_scan_tiles_kernel, dim3(1), dim3(256), 0, st, (unsigned long long*)tiles.p, nT, (unsigned long long*)scal.p);
		unsigned long long hs[2] = {0, 0};
		SGMCHK(e, hipMemcpyAsync(hs, scal.p, 16, hipMemcpyDeviceToHost, st));
		SGMCHK(e, hipStreamSynchronize(st));
		const uint64_t numCosts = hs[0]; const int mx = (int)(hs[1] & 0xffffffffull);
		if (numCosts == 0 || mx <= 0 || mx > 256) { e->err = "tsgm: empty or too wide disparity ranges at a level"; return SGMHIP_E_ARG; }
		{ const int rc = sgmReserve(e, lw, lh, numCosts, mx); if (rc) return rc; }
		hipLaunchKernelGGL(sgmt_expand_kernel, dim3(nT), dim3(256), 0, st, (const short2*)ranges.p, dW, dH, vw, n2, (const unsigned long long*)tiles.p, e->d_pixels);
		const size_t nImg = (size_t)lw * lh;
		SGMCHK(e, hipMemcpyAsync(e->d_color, bgr, nImg * 3, hipMemcpyDeviceToDevice, st));
		SGMCHK(e, hipMemcpyAsync(e->d_grayL, gA, nImg * 4, hipMemcpyDeviceToDevice, st));
		SGMCHK(e, hipMemcpyAsync(e->d_grayR, gB, nImg * 4, hipMemcpyDeviceToDevice, st));
		// narrow ranges (every level but the first): 16-lane sub-groups; wide ranges: one wavefront per pixel / line
		const int saved = e->subGroups;
		if (!saved && numCosts <= (uint64_t)n2 * 24) e->subGroups = 16;
		const int rcm = sgmMatch(e, P1);
		e->subGroups = saved;
		if (rcm) return rcm;
		SGMCHK(e, hipMemcpyAsync(out, e->d_disp, n2 * 2, hipMemcpyDeviceToDevice, st));
		return 0;
	};
	auto speckles = [&](int16_t* d, int vw, int vh) {
		const int n = vw * vh; const unsigned g = gridFor((size_t)n);
		hipLaunchKernelGGL(sgmp_speckle_init_kernel, dim3(g), dim3(256), 0, st, (int*)par.p, (int*)siz.p, n);
		hipLaunchKernelGGL(sgmp_speckle_hook_kernel, dim3(g), dim3(256), 0, st, (const int16_t*)d, (int*)par.p, vw, vh, 5);
		hipLaunchKernelGGL(sgmp_speckle_flatten_kernel, dim3(g), dim3(256), 0, st, (int*)par.p, (int*)siz.p, n);
		hipLaunchKernelGGL(sgmp_speckle_apply_kernel, dim3(g), dim3(256), 0, st, d, (const int*)par.p, (const int*)siz.p, n, nSpeckleSize);
	};
	for (int lvl = k; lvl >= 0; --lvl) {
		const int f = 1 << lvl, lw = w / f, lh = h / f, vw = lw - 2 * SGM_HW, vh = lh - 2 * SGM_HW;
		const size_t nImg = (size_t)lw * lh, nV = (size_t)vw * vh;
		const void *pLB, *pRB, *pLG, *pRG;
		if (f == 1) { pLB = fLB.p; pRB = fRB.p; pLG = fLG.p; pRG = fRG.p; }
		else {
			hipLaunchKernelGGL(sgmt_area_u8x3_kernel, dim3(gridFor(nImg * 3)), dim3(256), 0, st, (const unsigned char*)fLB.p, w, (unsigned char*)lB.p, lw, lh, f);
			hipLaunchKernelGGL(sgmt_area_u8x3_kernel, dim3(gridFor(nImg * 3)), dim3(256), 0, st, (const unsigned char*)fRB.p, w, (unsigned char*)rB.p, lw, lh, f);
			hipLaunchKernelGGL(sgmt_area_f32_kernel, dim3(gridFor(nImg)), dim3(256), 0, st, (const float*)fLG.p, w, (float*)lG.p, lw, lh, f);
			hipLaunchKernelGGL(sgmt_area_f32_kernel, dim3(gridFor(nImg)), dim3(256), 0, st, (const float*)fRG.p, w, (float*)rG.p, lw, lh, f);
			pLB = lB.p; pRB = rB.p; pLG = lG.p; pRG = rG.p;
		}
		if (first) {
			const int hw2 = cvRound(lw * 0.5), hh2 = cvRound(lh * 0.5);                 // Image8U::computeResize(size, 0.5), :622
			dW = hw2 - 2 * SGM_HW; dH = hh2 - 2 * SGM_HW;
			if (dW <= 0 || dH <= 0) { e->err = "tsgm: coarsest level too small"; return SGMHIP_E_ARG; }
			if (initLeftDisparity) SGMCHK(e, hipMemcpyAsync(leftDisp, initLeftDisparity, (size_t)dW * dH * 2, hipMemcpyHostToDevice, st));
			else hipLaunchKernelGGL(sgmt_fill_i16_kernel, dim3(gridFor((size_t)dW * dH)), dim3(256), 0, st, leftDisp, (size_t)dW * dH, (short)SGMP_NO_DISP);
			hipLaunchKernelGGL(sgmt_mask_first_kernel, dim3(gridFor(nV)), dim3(256), 0, st, (const unsigned char*)fLM.p, w, lm, vw, vh, f);   // :627-631
			hipLaunchKernelGGL(sgmt_mask_first_kernel, dim3(gridFor(nV)), dim3(256), 0, st, (const unsigned char*)fRM.p, w, rm, vw, vh, f);
		} else {
			hipLaunchKernelGGL(sgmp_upscale_mask_kernel, dim3(gridFor(nV)), dim3(256), 0, st, (const uint8_t*)lm, mW, mH, lmNext, vw, vh);   // :634-635
			hipLaunchKernelGGL(sgmp_upscale_mask_kernel, dim3(gridFor(nV)), dim3(256), 0, st, (const uint8_t*)rm, mW, mH, rmNext, vw, vh);
			std::swap(lm, lmNext); std::swap(rm, rmNext);
		}
		mW = vw; mH = vh;
		const int a = first ? 11 : 5, b = first ? 33 : 7;
		// right -> left with ranges from the flipped previous disparities (:641-654)
		const size_t nD = (size_t)dW * dH;
		SGMCHK(e, hipMemsetAsync(keys.p, 0, nD * 4, st));
		hipLaunchKernelGGL(sgmp_flip_scatter_kernel, dim3(gridFor(nD)), dim3(256), 0, st, (const int16_t*)leftDisp, (uint32_t*)keys.p, dW, dH);
		hipLaunchKernelGGL(sgmp_flip_decode_kernel, dim3(gridFor(nD)), dim3(256), 0, st, (const uint32_t*)keys.p, rightDisp, nD);
		{ const int rc = rangeAndMatch(rightDisp, rm, vw, vh, lw, lh, pRB, pRG, pLG, a, b, rightNew); if (rc) return rc; }
		// left -> right (:657-667)
		{ const int rc = rangeAndMatch(leftDisp, lm, vw, vh, lw, lh, pLB, pLG, pRG, a, b, leftNew); if (rc) return rc; }
		std::swap(leftDisp, leftNew); std::swap(rightDisp, rightNew);
		dW = vw; dH = vh;
		hipLaunchKernelGGL(sgmp_cross_check_kernel, dim3(gridFor(nV)), dim3(256), 0, st, leftDisp, (const int16_t*)rightDisp, vw, vw, vh, 1);
		if (first) {                                                                   // :680-690
			hipLaunchKernelGGL(sgmp_cross_check_kernel, dim3(gridFor(nV)), dim3(256), 0, st, rightDisp, (const int16_t*)leftDisp, vw, vw, vh, 1);
			speckles(leftDisp, vw, vh); speckles(rightDisp, vw, vh);
			hipLaunchKernelGGL(sgmp_extract_mask_kernel, dim3((vh + 63) / 64), dim3(64), 0, st, (const int16_t*)leftDisp, lm, vw, vh, 3);
			hipLaunchKernelGGL(sgmp_extract_mask_kernel, dim3((vh + 63) / 64), dim3(64), 0, st, (const int16_t*)rightDisp, rm, vw, vh, 3);
		}
		first = false; ++levels;
	}
	// sub-pixel refinement on the resident sums of the last (left) Match with the cross-checked map (:699)
	const size_t nV = (size_t)dW * dH;
	SGMCHK(e, hipMemcpyAsync(e->d_disp, leftDisp, nV * 2, hipMemcpyDeviceToDevice, st));
	if (subpixelSteps > 1)
		hipLaunchKernelGGL(sgmp_refine_kernel, dim3(gridFor(nV)), dim3(256), 0, st, e->d_pixels, e->d_accums, (long)nV, e->d_disp, subpixelMode, subpixelSteps);
	SGMCHK(e, hipGetLastError());
	SGMCHK(e, hipMemcpyAsync(disparity, e->d_disp, nV * 2, hipMemcpyDeviceToHost, st));
	SGMCHK(e, hipMemcpyAsync(cost, e->d_cost, nV * 2, hipMemcpyDeviceToHost, st));
	SGMCHK(e, hipStreamSynchronize(st));
	if (numLevels) *numLevels = levels;
	return 0;
}

} // extern "C"
