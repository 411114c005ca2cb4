// pm_filter.hip -- DepthMapsData::FilterDepthMap (libs/MVS/SceneDensify.cpp:1050-1299) on the HBM-resident scene.
//
// Splat (:1084-1128): every neighbour pixel is projected into the reference view and z-tested onto the 4 surrounding
// pixels.  The reference does this sequentially in raster order with "if (depthRef != 0 && depthRef < z) continue;
// depthRef = z; conf = c", i.e. the final depth is the minimum z and the final confidence belongs to the LAST source pixel
// (raster order) that attains it.  That is one 64-bit atomicMin on the key (float bits of z << 32) | (0xFFFFFFFF - source
// index): positive floats order like their bit patterns, and the complemented index makes the later pixel win ties.
// Vote (:1141-1290): independent per reference pixel, neighbours visited in the reference's order (n = N-1 .. 0) so the
// float accumulations round identically.  Camera maths in double, cv::Matx accumulation order (Camera.h:338-399).
#pragma once
#include <hip/hip_runtime.h>

#define PMF_MAXN 8   // numMaxNeighbors, SceneDensify.cpp:2152
#define PMF_EMPTY 0xFFFFFFFFFFFFFFFFull

struct PMFCam { double K[9], R[9], C[3]; };
struct PMFTask {          // one reference view
	PMFCam ref; PMFCam nb[PMF_MAXN];
	const float* refDepth; const float* refConf;
	const float* nbDepth[PMF_MAXN]; const float* nbConf[PMF_MAXN];
	unsigned long long* splat;   // [N][h*w] packed keys
	float* outDepth; float* outConf;
	int N, w, h, filterable;
	float dMin, dMax;
	int nbw[PMF_MAXN], nbh[PMF_MAXN];   // every neighbour's maps have their own size (depthData.depthMap.size(), SceneDensify.cpp:1085)
};

__device__ __forceinline__ void pmf_mulMV(const double* M, double v0, double v1, double v2, double* o) {
#pragma unroll
	for (int i = 0; i < 3; ++i) o[i] = ((0.0 + M[i * 3] * v0) + M[i * 3 + 1] * v1) + M[i * 3 + 2] * v2;
}
__device__ __forceinline__ void pmf_mulMtV(const double* M, double v0, double v1, double v2, double* o) {
#pragma unroll
	for (int i = 0; i < 3; ++i) o[i] = ((0.0 + M[i] * v0) + M[3 + i] * v1) + M[6 + i] * v2;
}
__device__ __forceinline__ void pmf_I2W(const PMFCam& c, double x, double y, double z, double* X) { // Camera.h:338-356
	double t[3];
	pmf_mulMtV(c.R, (x - c.K[2]) * z / c.K[0], (y - c.K[5]) * z / c.K[4], z, t);
	X[0] = t[0] + c.C[0]; X[1] = t[1] + c.C[1]; X[2] = t[2] + c.C[2];
}
__device__ __forceinline__ void pmf_W2C(const PMFCam& c, const double* X, double* o) { pmf_mulMV(c.R, X[0] - c.C[0], X[1] - c.C[1], X[2] - c.C[2], o); }
__device__ __forceinline__ bool pmf_similar(float d0, float d1, float th) { return pm_fabsf(d0 - d1) / d0 < th; }

__global__ void pmf_clear_kernel(unsigned long long* p, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = PMF_EMPTY;
}

// grid.y = reference views of the batch, grid.z = neighbour slot
__global__ void pmf_splat_kernel(const PMFTask* __restrict__ tasks) {
	const PMFTask& t = tasks[blockIdx.y];
	const int n = blockIdx.z;
	if (!t.filterable || n >= t.N) return;
	const size_t P = (size_t)t.w * t.h;
	const int nw = t.nbw[n];
	const size_t Pn = (size_t)nw * t.nbh[n];
	const float* __restrict__ src = t.nbDepth[n];
	unsigned long long* dst = t.splat + (size_t)n * P;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < Pn; i += (size_t)gridDim.x * blockDim.x) {
		const float depth = src[i];
		if (depth == 0) continue;
		const int xj = (int)(i % nw), yi = (int)(i / nw);
		double X[3], camX[3];
		pmf_I2W(t.nb[n], (double)xj, (double)yi, (double)depth, X);
		pmf_W2C(t.ref, X, camX);
		if (camX[2] <= 0) continue;
		const double ix = t.ref.K[2] + t.ref.K[0] * (camX[0] / camX[2]), iy = t.ref.K[5] + t.ref.K[4] * (camX[1] / camX[2]); // TransformPointC2I
		const float z = (float)camX[2];
		const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
		const int x0 = (int)floor(ix), x1 = (int)ceil(ix), y0 = (int)floor(iy), y1 = (int)ceil(iy);
		const int px[4] = {x0, x0, x1, x1}, py[4] = {y0, y1, y0, y1};
#pragma unroll
		for (int p = 0; p < 4; ++p)
			if (px[p] >= 0 && py[p] >= 0 && px[p] < t.w && py[p] < t.h)
				atomicMin(dst + (size_t)py[p] * t.w + px[p], key);
	}
}

// the per-pixel vote; bAdjust selects the confidence-weighted fusion (:1141-1212) or the strict agreement test (:1213-1290)
__global__ void pmf_vote_kernel(const PMFTask* __restrict__ tasks, int bAdjust, unsigned nMinViews, unsigned nMinViewsAdjust, float fDepthDiffThreshold) {
	const PMFTask& t = tasks[blockIdx.y];
	if (!t.filterable) return;
	const int w = t.w, h = t.h, N = t.N;
	const size_t P = (size_t)w * h;
	const float thDepthDiff = fDepthDiffThreshold * 1.2f;
	for (size_t xr = (size_t)blockIdx.x * blockDim.x + threadIdx.x; xr < P; xr += (size_t)gridDim.x * blockDim.x) {
		const float depth = t.refDepth[xr];
		float od = 0.f, oc = 0.f;
		if (depth != 0) {
			const int j = (int)(xr % w), i = (int)(xr / w);
			if (bAdjust) {
				float posConf = t.refConf[xr], negConf = 0.f;
				float avgDepth = depth * posConf;
				unsigned nPos = 0, nNeg = 0;
				bool discard = false;
				for (int n = N; n-- > 0; ) {
					const unsigned long long key = t.splat[(size_t)n * P + xr];
					if (key == PMF_EMPTY) { // d == 0
						if (nPos + nNeg + (unsigned)n < nMinViews) { discard = true; break; }
						continue;
					}
					const float d = __uint_as_float((unsigned)(key >> 32));
					const float cn = t.nbConf[n][0xFFFFFFFFu - (unsigned)key]; // confMaps[n](xRef)
					if (pmf_similar(depth, d, thDepthDiff)) { avgDepth += d * cn; posConf += cn; ++nPos; }
					else {
						if (depth > d) negConf += cn; // occlusion
						else {                          // free-space violation
							double X[3], cx[3];
							pmf_I2W(t.ref, (double)j, (double)i, (double)depth, X);
							pmf_W2C(t.nb[n], X, cx);
							const double ux = t.nb[n].K[2] + t.nb[n].K[0] * (cx[0] / cx[2]), uy = t.nb[n].K[5] + t.nb[n].K[4] * (cx[1] / cx[2]);
							const int x = (int)floor(ux + .5), y = (int)floor(uy + .5); // ROUND2INT(double)
							if (x >= 0 && y >= 0 && x < t.nbw[n] && y < t.nbh[n]) { const float c = t.nbConf[n][(size_t)y * t.nbw[n] + x]; negConf += (c > 0 ? c : cn); }   // confMap.isInside(x), :1181
							else negConf += cn;
						}
						++nNeg;
					}
				}
				if (!discard && nPos >= nMinViewsAdjust && posConf > negConf) {
					avgDepth /= posConf;
					if (t.dMin <= avgDepth && avgDepth < t.dMax) { od = avgDepth; oc = posConf - negConf; }
				}
			} else {
				const float thStrict = fDepthDiffThreshold * 0.8f;
				unsigned nGood = 0, nViews = 0;
				for (int n = N; n-- > 0; ) {
					const unsigned long long key = t.splat[(size_t)n * P + xr];
					if (key != PMF_EMPTY) { ++nViews; if (pmf_similar(depth, __uint_as_float((unsigned)(key >> 32)), thStrict)) ++nGood; }
				}
				if (!(nGood < nMinViews || nGood < nViews * 75u / 100u)) {
					nGood = 0; nViews = 0;
					const int dxs[4] = {-1, 1, 0, 0}, dys[4] = {0, 0, -1, 1};
#pragma unroll
					for (int dd = 0; dd < 4; ++dd) {
						const int x = j + dxs[dd], y = i + dys[dd];
						if (!(x >= 0 && y >= 0 && x < w && y < h)) continue;
						for (int n = N; n-- > 0; ) {
							const unsigned long long key = t.splat[(size_t)n * P + (size_t)y * w + x];
							if (key != PMF_EMPTY) { ++nViews; if (pmf_similar(depth, __uint_as_float((unsigned)(key >> 32)), thDepthDiff)) ++nGood; }
						}
					}
					if (!(nGood < nMinViews * 2u || nGood < nViews * 65u / 100u)) { od = depth; oc = t.refConf[xr]; }
				}
			}
		}
		t.outDepth[xr] = od; t.outConf[xr] = oc;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// DepthMapsData::GapInterpolation (SceneDensify.cpp:904-1045).  The sequential row pass only ever reads original
// values (it fills behind the scan position), so every gap is independent: the thread of the valid pixel that closes
// a gap measures it leftwards and fills it.  The column pass works on the row pass' result, hence the two buffers.
struct PMGTask { const float* id; const float* in; const float* ic; float* od; float* on; float* oc; int w, h; };
__global__ void pmf_gap_kernel(const PMGTask* __restrict__ tasks, int rows, unsigned nIpolGapSize, float th) {
	const PMGTask& t = tasks[blockIdx.y];
	const int w = t.w, h = t.h;
	const size_t P = (size_t)w * h;
	for (size_t at = (size_t)blockIdx.x * blockDim.x + threadIdx.x; at < P; at += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(at % w), y = (int)(at / w);
		const int u = rows ? x : y;                 // position along the scan direction
		const size_t step = rows ? 1 : (size_t)w;
		const float d1 = t.id[at];
		// every pixel is copied through; gap pixels are overwritten by the thread that owns the gap (they are invalid, so
		// their own thread writes the unchanged invalid value first only if no gap owner exists -- see below)
		if (d1 <= 0) {
			// is this pixel inside a fillable gap?  then its closing pixel's thread writes it; otherwise copy it
			unsigned left = 0; while ((int)left < u && left <= nIpolGapSize && t.id[at - (left + 1) * step] <= 0) ++left;
			unsigned right = 0; const int lim = (rows ? w : h) - 1 - u;
			while ((int)right < lim && left + right < nIpolGapSize + 1 && t.id[at + (right + 1) * step] <= 0) ++right;
			const unsigned count = left + right + 1;
			bool filled = false;
			if (count <= nIpolGapSize && (int)left < u && (int)right < lim) {
				const float d0 = t.id[at - (left + 1) * step], de = t.id[at + (right + 1) * step];
				filled = d0 > 0 && de > 0 && pmf_similar(d0, de, th);
			}
			if (!filled) { t.od[at] = d1; t.on[at * 3] = t.in[at * 3]; t.on[at * 3 + 1] = t.in[at * 3 + 1]; t.on[at * 3 + 2] = t.in[at * 3 + 2]; t.oc[at] = t.ic[at]; }
			continue;
		}
		t.od[at] = d1; t.on[at * 3] = t.in[at * 3]; t.on[at * 3 + 1] = t.in[at * 3 + 1]; t.on[at * 3 + 2] = t.in[at * 3 + 2]; t.oc[at] = t.ic[at];
		unsigned count = 0;
		while ((int)count < u && count <= nIpolGapSize && t.id[at - (count + 1) * step] <= 0) ++count;
		if (count == 0 || count > nIpolGapSize || !((unsigned)u > count)) continue;
		const size_t af = at - (count + 1) * step;
		const float d0 = t.id[af];
		if (!(d0 > 0) || !pmf_similar(d0, d1, th)) continue;
		const float diff = (d1 - d0) / (float)(count + 1);
		float d = d0;
		const float c = pm_minf(t.ic[af], t.ic[at]);
		float p0 = pm_atan2f(t.in[af * 3 + 1], t.in[af * 3]), p1 = pm_acosf(pm_clampf(t.in[af * 3 + 2], -1.f, 1.f)); // Normal2Dir, Util.inl:754-759
		const float q0 = pm_atan2f(t.in[at * 3 + 1], t.in[at * 3]), q1 = pm_acosf(pm_clampf(t.in[at * 3 + 2], -1.f, 1.f));
		const float dd0 = (q0 - p0) / (float)(count + 1), dd1 = (q1 - p1) / (float)(count + 1);
		for (size_t ac = af + step; ac != at; ac += step) {
			d += diff; t.od[ac] = d;
			p0 += dd0; p1 += dd1;
			float sx, cx, sy, cy; pm_sincosf(p0, &sx, &cx); pm_sincosf(p1, &sy, &cy); // Dir2Normal
			t.on[ac * 3] = cx * sy; t.on[ac * 3 + 1] = sx * sy; t.on[ac * 3 + 2] = cy;
			t.oc[ac] = c;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// DepthMapsData::RemoveSmallSegments (SceneDensify.cpp:809-900).  The reference grows regions from seeds in
// column-major order along DIRECTED edges (cur -> nb iff |d_cur - d_nb| / d_cur < th), so a segment is "everything
// reachable from the first not-yet-claimed pixel".  Exact parallel reformulation:
//   * pixels joined by MUTUAL edges (both directions similar) are always claimed together -> connected components of
//     the mutual graph by lock-free union-find (root = smallest column-major index = the pixel the sequential scan
//     meets first);
//   * the few ASYMMETRIC edges (exactly one direction) form a small quotient graph between components, on which the
//     host replays the seed order (pmhip_scene_remove_small_segments); components without asymmetric edges are
//     decided by their size alone.
__device__ __forceinline__ int pmf_find(int* parent, int a) {
	int p = parent[a];
	while (p != a) { const int gp = parent[p]; if (gp != p) parent[a] = gp; a = p; p = parent[a]; } // parents only decrease: safe under races
	return a;
}
__device__ __forceinline__ void pmf_union(int* parent, int a, int b) {
	for (;;) {
		a = pmf_find(parent, a); b = pmf_find(parent, b);
		if (a == b) return;
		if (a > b) { const int t = a; a = b; b = t; }          // a < b: hook the larger root under the smaller
		const int old = atomicMin(&parent[b], a);
		if (old == b) return;
		b = old;
	}
}
__global__ void pmf_cc_init_kernel(int* parent, int* size, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { parent[i] = i; size[i] = 0; }
}
// ids are column-major: id(x,y) = x*h + y (the reference scans u outer, v inner)
__global__ void pmf_cc_hook_kernel(const float* __restrict__ depth, int* parent, int w, int h, float th) {
	const int n = w * h;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int x = i % w, y = i / w;
		const float d = depth[i];
		if (!(d > 0)) continue;
		if (x + 1 < w) { const float e = depth[i + 1]; if (e > 0 && pmf_similar(d, e, th) && pmf_similar(e, d, th)) pmf_union(parent, x * h + y, (x + 1) * h + y); }
		if (y + 1 < h) { const float e = depth[i + w]; if (e > 0 && pmf_similar(d, e, th) && pmf_similar(e, d, th)) pmf_union(parent, x * h + y, x * h + y + 1); }
	}
}
// read-only find: in the flatten pass the only writer of parent[i] must be thread i (a compressing find of another
// thread could overwrite the final root with a stale grandparent)
__device__ __forceinline__ int pmf_find_ro(const int* parent, int a) { int p = parent[a]; while (p != a) { a = p; p = parent[a]; } return a; }
// Component sizes: the lanes of a wavefront that found the same root add ONE count between them.  A depth map is mostly one component, and
// 8 M atomic increments of a single word took 0.1 s per 3840x2160 view (profiles/r03_config5_32_views_4k.json) -- now one atomic per wave and root.
__global__ void pmf_cc_flatten_kernel(int* parent, int* size, int n) {
	const int lane = threadIdx.x & 63;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int r = pmf_find_ro(parent, i); parent[i] = r;
		bool pending = true;
		for (;;) {
			const unsigned long long todo = __ballot(pending);
			if (todo == 0ull) break;
			const int leader = __ffsll((long long)todo) - 1;
			const int lead = __shfl(r, leader, 64);
			const unsigned long long same = __ballot(pending && r == lead);
			if (pending && r == lead) { if (lane == leader) atomicAdd(&size[lead], (int)__popcll(same)); pending = false; }
		}
	}
}
// asymmetric edges between different components: (root of cur, root of nb) with cur -> nb similar but not nb -> cur
__global__ void pmf_cc_asym_kernel(const float* __restrict__ depth, const int* __restrict__ parent, int w, int h, float th, int* edges, int* nEdges, int cap) {
	const int n = w * h;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int x = i % w, y = i / w;
		const float d = depth[i];
		if (!(d > 0)) continue;
		const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int qx = x + dx[k], qy = y + dy[k];
			if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
			const float e = depth[qy * w + qx];
			if (e > 0 && pmf_similar(d, e, th) && !pmf_similar(e, d, th)) {
				const int ra = parent[x * h + y], rb = parent[qx * h + qy];
				if (ra != rb) { const int at = atomicAdd(nEdges, 1); if (at < cap) { edges[2 * at] = ra; edges[2 * at + 1] = rb; } }
			}
		}
	}
}
// sizes of the components the one-directional edges touch (out[k] = size[ids[k]]): the host replay needs these few, not the whole per-pixel array
__global__ void pmf_cc_gather_kernel(const int* __restrict__ size, const int* __restrict__ ids, int* __restrict__ out, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = size[ids[i]];
}
__global__ void pmf_cc_override_kernel(int* size, const int* __restrict__ pairs, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) size[pairs[2 * i]] = pairs[2 * i + 1];
}
__global__ void pmf_cc_apply_kernel(float* depth, float* normal, float* conf, const int* __restrict__ parent, const int* __restrict__ size, int w, int h, int speckle) {
	const int n = w * h;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int x = i % w, y = i / w;
		if (size[parent[x * h + y]] < speckle) { depth[i] = 0.f; normal[3 * i] = 0.f; normal[3 * i + 1] = 0.f; normal[3 * i + 2] = 0.f; conf[i] = 0.f; }
	}
}
