// mvs_front.cpp -- see include/mvsfront.h.  Plain C++ (no GPU, no third-party library): the scene-side steps in front of the estimator,
// written against the reference's sources (file:line in the header) so that a host without OpenMVS can drive the engine from a scene.mvs.
// Mixed precision follows the reference term by term: cameras in double, the view scores in float with the reference's loop order.
#include "../../include/mvsfront.h"
#include "sml_text.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <memory>
#include <vector>
#include <map>
#include <array>

namespace {
const uint32_t NO_ID = 0xFFFFFFFFu;

struct Cam { std::string name, band; uint32_t w = 0, h = 0; double K[9], R[9], C[3]; };
struct Platform { std::string name; std::vector<Cam> cams; std::vector<double> poses; /* 12 doubles each: R[9], C[3] */ };
struct Image { std::string name, mask; uint32_t platform = NO_ID, camera = NO_ID, pose = NO_ID, id = NO_ID;
	float minDepth = 0, avgDepth = 0, maxDepth = 0;   // stored by archives of version > 6 (Interface.h:551-553)
	std::vector<MVSFViewScore> neighbors;             // Image::neighbors: the archive's view scores (Scene.cpp:158) or a view-neighbours file (Scene.cpp:413-457)
};
}

struct mvsf_scene {
	uint32_t version = 0;
	std::vector<Platform> platforms;
	std::vector<Image> images;
	std::vector<float> X;                    // 3 per point
	std::vector<uint32_t> viewStart, views;  // CSR
	double obbRot[9] = {1,0,0, 0,1,0, 0,0,1}, obbMin[3] = {0,0,0}, obbMax[3] = {0,0,0};
};

namespace {
struct Reader {
	const unsigned char* b; size_t n, o = 0; bool ok = true;
	// every length comes from the file: compare against what is left (k > n - o), never o + k, which wraps for a hostile 64-bit count
	bool take(void* dst, size_t k) { if (!ok || k > n - o) { ok = false; return false; } memcpy(dst, b + o, k); o += k; return true; }
	bool skip(size_t k) { if (!ok || k > n - o) { ok = false; return false; } o += k; return true; }
	// skip `count` records of `rec` bytes; the count is checked before it is multiplied
	bool skipn(uint64_t count, size_t rec) { if (!ok || count > (n - o) / rec) { ok = false; return false; } o += (size_t)count * rec; return true; }
	uint32_t u32() { uint32_t v = 0; take(&v, 4); return v; }
	uint64_t u64() { uint64_t v = 0; take(&v, 8); return v; }
	std::string str() { const uint64_t k = u64(); if (!ok || k > n - o) { ok = false; return std::string(); } std::string s((const char*)b + o, (size_t)k); o += (size_t)k; return s; }
};

// Camera::ScaleK, libs/MVS/Camera.h:146-152
void scaleK(const double* K, double s, double* o) {
	o[0] = K[0] * s; o[1] = K[1] * s; o[2] = (K[2] + 0.5) * s - 0.5;
	o[3] = 0;        o[4] = K[4] * s; o[5] = (K[5] + 0.5) * s - 0.5;
	o[6] = 0; o[7] = 0; o[8] = 1;
}
void mul33(const double* a, const double* b, double* c) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[i*3+k] * b[k*3+j]; c[i*3+j] = s; } }

struct PixCam { double K[9], R[9], C[3], P[12]; int w = 0, h = 0; bool valid = false; };

bool pixelCamera(const mvsf_scene& s, int idx, int w, int h, PixCam& pc) {
	const Image& im = s.images[idx];
	pc.valid = false;
	if (im.pose == NO_ID || im.platform >= s.platforms.size()) return false;
	const Platform& pl = s.platforms[im.platform];
	if (im.camera >= pl.cams.size() || (size_t)im.pose * 12 + 12 > pl.poses.size()) return false;
	const Cam& cam = pl.cams[im.camera];
	double K[9]; memcpy(K, cam.K, sizeof(K));
	if (cam.w > 0 && cam.h > 0) { double t[9]; scaleK(K, 1.0 / (double)(float)std::max(cam.w, cam.h), t); memcpy(K, t, sizeof(K)); }   // Scene::LoadInterface, Scene.cpp:100-104
	if (w <= 0 || h <= 0) { w = (int)cam.w; h = (int)cam.h; }
	if (w <= 0 || h <= 0) return false;
	const double sc = (double)(float)std::max(w, h);               // Camera::GetK, Camera.h:190-200
	if (K[2] != 0 || K[5] != 0) scaleK(K, sc, pc.K);
	else { const double t[9] = {K[0] * sc, 0, 0.5 * (w - 1), 0, K[4] * sc, 0.5 * (h - 1), 0, 0, 1}; memcpy(pc.K, t, sizeof(t)); }
	const double* Rp = &pl.poses[(size_t)im.pose * 12]; const double* Cp = Rp + 9;
	mul33(cam.R, Rp, pc.R);                                          // Platform::GetCamera, Platform.cpp:44-54
	for (int i = 0; i < 3; ++i) { double v = 0; for (int k = 0; k < 3; ++k) v += Rp[k*3+i] * cam.C[k]; pc.C[i] = v + Cp[i]; }
	double M[9]; mul33(pc.K, pc.R, M);                               // Camera::ComposeP
	for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) pc.P[i*4+j] = M[i*3+j]; pc.P[i*4+3] = -(M[i*3] * pc.C[0] + M[i*3+1] * pc.C[1] + M[i*3+2] * pc.C[2]); }
	pc.w = w; pc.h = h; pc.valid = true;
	return true;
}
inline double pointDepth(const PixCam& c, const float* X) { return (X[0] - c.C[0]) * c.R[6] + (X[1] - c.C[1]) * c.R[7] + (X[2] - c.C[2]) * c.R[8]; }   // Util.inl:462-464
inline void projectP(const PixCam& c, const float* X, float* out) {   // Camera::ProjectPointP<float>, Camera.h:308-320
	const float q0 = (float)(c.P[0] * X[0] + c.P[1] * X[1] + c.P[2] * X[2] + c.P[3]);
	const float q1 = (float)(c.P[4] * X[0] + c.P[5] * X[1] + c.P[6] * X[2] + c.P[7]);
	const float q2 = (float)(c.P[8] * X[0] + c.P[9] * X[1] + c.P[10] * X[2] + c.P[11]);
	const float inv = 1.f / q2;
	out[0] = q0 * inv; out[1] = q1 * inv;
}
bool roiBounded(const mvsf_scene& s, float* rot, float* pos, float* ext) {   // TOBB::Set + IsValid, OBB.inl:64-69,272-275
	float mn = 1;
	for (int i = 0; i < 9; ++i) rot[i] = (float)s.obbRot[i];
	for (int i = 0; i < 3; ++i) { const float a = (float)s.obbMin[i], b = (float)s.obbMax[i]; pos[i] = (b + a) * 0.5f; ext[i] = (b - a) * 0.5f; if (i == 0 || ext[i] < mn) mn = ext[i]; }
	return mn > 0;
}
bool roiContains(const float* rot, const float* pos, const float* ext, const float* X) {   // TOBB::Intersects(point), OBB.inl:388-400
	const float d[3] = {X[0] - pos[0], X[1] - pos[1], X[2] - pos[2]};
	for (int i = 0; i < 3; ++i) { const float v = rot[i*3] * d[0] + rot[i*3+1] * d[1] + rot[i*3+2] * d[2]; if (fabsf(v) > ext[i]) return false; }
	return true;
}

bool camerasFor(const mvsf_scene& s, const int* sizes, std::vector<PixCam>& cams, int& nCal) {
	cams.assign(s.images.size(), PixCam()); nCal = 0;
	for (size_t i = 0; i < s.images.size(); ++i) { if (pixelCamera(s, (int)i, sizes ? sizes[2*i] : 0, sizes ? sizes[2*i+1] : 0, cams[i])) ++nCal; }
	return nCal > 0;
}

// Scene::SelectNeighborViews, Scene.cpp:801-934
bool selectNeighborViews(const mvsf_scene& s, const std::vector<PixCam>& cams, int nCalibrated, uint32_t ID, unsigned nMinViews, unsigned nMinPointViews,
		float fOptimAngle, unsigned nInsideROI, std::vector<MVSFViewScore>& neighbors, std::vector<uint32_t>& points, float& avgDepth) {
	struct Score { float score, avgScale, avgAngle; uint32_t points; };
	std::vector<Score> scores(s.images.size(), Score{0, 0, 0, 0});
	if (nMinPointViews > (unsigned)nCalibrated) nMinPointViews = (unsigned)nCalibrated;
	unsigned nPoints = 0;
	avgDepth = 0;
	const float sigmaAngleSmall = -1.f / (2.f * ((fOptimAngle * 0.38f) * (fOptimAngle * 0.38f)));
	const float sigmaAngleLarge = -1.f / (2.f * ((fOptimAngle * 0.7f) * (fOptimAngle * 0.7f)));
	float rot[9], pos[3], ext[3];
	const bool bCheckInsideROI = nInsideROI > 0 && roiBounded(s, rot, pos, ext);
	const PixCam& camA = cams[ID];
	const size_t nPts = s.viewStart.size() - 1;
	points.clear(); neighbors.clear();
	for (size_t idx = 0; idx < nPts; ++idx) {
		const uint32_t* vb = &s.views[s.viewStart[idx]]; const uint32_t* ve = &s.views[s.viewStart[idx + 1]];
		if (std::find(vb, ve, ID) == ve) continue;
		const float* point = &s.X[idx * 3];
		float wROI = 1.f;
		if (bCheckInsideROI && !roiContains(rot, pos, ext, point)) { if (nInsideROI > 1) continue; wROI = 0.7f; }
		const float depth = (float)pointDepth(camA, point);
		if (depth <= 0) continue;
		if ((unsigned)(ve - vb) >= nMinPointViews) points.push_back((uint32_t)idx);
		avgDepth += depth;
		++nPoints;
		const float V1[3] = {(float)(camA.C[0] - point[0]), (float)(camA.C[1] - point[1]), (float)(camA.C[2] - point[2])};
		const float footprint1 = (float)(camA.K[0] / pointDepth(camA, point));
		for (const uint32_t* pv = vb; pv != ve; ++pv) {
			const uint32_t view = *pv;
			if (view == ID) continue;
			const PixCam& camB = cams[view];
			const float V2[3] = {(float)(camB.C[0] - point[0]), (float)(camB.C[1] - point[1]), (float)(camB.C[2] - point[2])};
			float ca = ((V1[0]*V2[0] + V1[1]*V2[1]) + V1[2]*V2[2]) / sqrtf(((V1[0]*V1[0] + V1[1]*V1[1]) + V1[2]*V1[2]) * ((V2[0]*V2[0] + V2[1]*V2[1]) + V2[2]*V2[2]));
			ca = ca < -1.f ? -1.f : (ca > 1.f ? 1.f : ca);
			const float fAngle = acosf(ca);
			const float dA = fAngle - fOptimAngle;
			const float wAngle = expf((dA * dA) * (fAngle < fOptimAngle ? sigmaAngleSmall : sigmaAngleLarge));
			const float footprint2 = (float)(camB.K[0] / pointDepth(camB, point));
			const float fScaleRatio = footprint1 / footprint2;
			float wScale;
			if (fScaleRatio > 1.6f) wScale = (1.6f / fScaleRatio) * (1.6f / fScaleRatio);
			else if (fScaleRatio >= 1.f) wScale = 1.f;
			else wScale = fScaleRatio * fScaleRatio;
			Score& sc = scores[view];
			sc.score += (wAngle > 0.1f ? wAngle : 0.1f) * wScale * wROI;
			sc.avgScale += fScaleRatio;
			sc.avgAngle += fAngle;
			++sc.points;
		}
	}
	if (nPoints > 3) avgDepth /= nPoints;
	std::vector<unsigned char> inPoints(nPts, 0);
	for (uint32_t p : points) inPoints[p] = 1;
	const float boundsA[2] = {(float)camA.w, (float)camA.h};
	for (size_t IDB = 0; IDB < s.images.size(); ++IDB) {
		if (!cams[IDB].valid || IDB == ID) continue;
		const Score& sc = scores[IDB];
		if (sc.points < 3) continue;
		const float boundsB[2] = {(float)cams[IDB].w, (float)cams[IDB].h};
		unsigned char surface[16 * 16] = {0};
		unsigned nProjs = 0;
		for (uint32_t idx : points) {
			const uint32_t* vb = &s.views[s.viewStart[idx]]; const uint32_t* ve = &s.views[s.viewStart[idx + 1]];
			if (std::find(vb, ve, (uint32_t)IDB) == ve) continue;
			float a[2], b[2];
			projectP(camA, &s.X[idx * 3], a); projectP(cams[IDB], &s.X[idx * 3], b);
			if (!(a[0] >= 0 && a[1] >= 0 && a[0] < boundsA[0] && a[1] < boundsA[1]) || !(b[0] >= 0 && b[1] >= 0 && b[0] < boundsB[0] && b[1] < boundsB[1])) continue;
			++nProjs;
			const int cx = (int)floorf((a[0] / boundsA[0]) * 16.f), cy = (int)floorf((a[1] / boundsA[1]) * 16.f);   // ComputeCoveredArea<float,2,16,false>, Util.inl:846-866
			surface[cx * 16 + cy] = 1;
		}
		if (nProjs == 0) continue;
		unsigned cells = 0; for (unsigned char c : surface) cells += c;
		const float area = (float)cells / (16.f * 16.f);
		MVSFViewScore n;
		n.ID = (uint32_t)IDB; n.points = sc.points; n.scale = sc.avgScale / sc.points; n.angle = sc.avgAngle / sc.points; n.area = area;
		n.score = sc.score * (area > 0.01f ? area : 0.01f);
		neighbors.push_back(n);
	}
	std::stable_sort(neighbors.begin(), neighbors.end(), [](const MVSFViewScore& i, const MVSFViewScore& j) { return i.score > j.score; });
	const unsigned need = std::min<unsigned>(nMinViews, (unsigned)nCalibrated - 1);
	return !(points.size() <= 3 || neighbors.size() < need);
}

// Scene::FilterNeighborViews, Scene.cpp:953-968
void filterNeighborViews(std::vector<MVSFViewScore>& nb, float fMinArea, float fMinScale, float fMaxScale, float fMinAngle, float fMaxAngle, unsigned nMaxViews) {
	const unsigned nMinViews = std::max(4u, nMaxViews * 3 / 4);
	for (size_t n = nb.size(); n-- > 0; ) {
		const MVSFViewScore& v = nb[n];
		if (nb.size() > nMinViews && (v.area < fMinArea || !(fMinScale <= v.scale && v.scale < fMaxScale) || !(fMinAngle <= v.angle && v.angle < fMaxAngle)))   // ISINSIDE is half-open, Types.h:1193
			nb.erase(nb.begin() + n);
	}
	if (nb.size() > nMaxViews) nb.resize(nMaxViews);
}

// ---- Delaunay triangulation of the projections (Bowyer-Watson; the reference uses CGAL::Delaunay_triangulation_2) --------------------
struct Tri { int v[3]; };
inline double orient(const double* a, const double* b, const double* c) { return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]); }
inline bool inCircle(const double* a, const double* b, const double* c, const double* d) {   // a, b, c counter-clockwise
	const double ax = a[0] - d[0], ay = a[1] - d[1], bx = b[0] - d[0], by = b[1] - d[1], cx = c[0] - d[0], cy = c[1] - d[1];
	const double det = (ax * ax + ay * ay) * (bx * cy - cx * by) - (bx * bx + by * by) * (ax * cy - cx * ay) + (cx * cx + cy * cy) * (ax * by - bx * ay);
	return det > 0;
}
void delaunay(const std::vector<double>& xy, std::vector<Tri>& out) {
	const int n = (int)(xy.size() / 2);
	out.clear();
	if (n < 3) return;
	double mnx = xy[0], mxx = xy[0], mny = xy[1], mxy = xy[1];
	for (int i = 1; i < n; ++i) { mnx = std::min(mnx, xy[2*i]); mxx = std::max(mxx, xy[2*i]); mny = std::min(mny, xy[2*i+1]); mxy = std::max(mxy, xy[2*i+1]); }
	const double span = std::max(mxx - mnx, mxy - mny) + 1.0, cx = 0.5 * (mnx + mxx), cy = 0.5 * (mny + mxy), big = span * 1e5;
	std::vector<double> P(xy); P.insert(P.end(), {cx - 2 * big, cy - big, cx + 2 * big, cy - big, cx, cy + 2 * big});
	std::vector<Tri> tris; tris.push_back(Tri{{n, n + 1, n + 2}});
	std::vector<std::pair<int,int>> edges;
	std::vector<Tri> keep;
	for (int p = 0; p < n; ++p) {
		const double* pp = &P[2 * p];
		bool dup = false;
		for (int q = 0; q < p && !dup; ++q) dup = P[2*q] == pp[0] && P[2*q+1] == pp[1];
		if (dup) continue;                                        // coincident projections: the first one keeps the vertex
		edges.clear(); keep.clear();
		for (const Tri& t : tris) {
			if (inCircle(&P[2*t.v[0]], &P[2*t.v[1]], &P[2*t.v[2]], pp)) { for (int k = 0; k < 3; ++k) edges.emplace_back(t.v[k], t.v[(k+1)%3]); }
			else keep.push_back(t);
		}
		tris.swap(keep);
		for (size_t i = 0; i < edges.size(); ++i) {
			bool shared = false;
			for (size_t j = 0; j < edges.size() && !shared; ++j) shared = i != j && edges[i].first == edges[j].second && edges[i].second == edges[j].first;
			if (shared) continue;
			Tri t{{edges[i].first, edges[i].second, p}};
			if (orient(&P[2*t.v[0]], &P[2*t.v[1]], &P[2*t.v[2]]) < 0) std::swap(t.v[0], t.v[1]);
			tris.push_back(t);
		}
	}
	for (const Tri& t : tris) if (t.v[0] < n && t.v[1] < n && t.v[2] < n) out.push_back(t);
}
} // namespace

extern "C" {

void mvsf_default_options(MVSFOptions* o) {
	if (!o) return;
	o->nMinViews = 2; o->nMaxViews = 12; o->nMinViewsTrustPoint = 2; o->nNumViews = 0; o->nPointInsideROI = 1;
	o->fViewMinScore = 2.0f; o->fViewMinScoreRatio = 0.03f; o->fMinArea = 0.05f; o->fMinAngle = 3.0f; o->fOptimAngle = 12.0f; o->fMaxAngle = 65.0f;
}

static int loadScene(const char* path, mvsf_scene** out);
// nothing may throw across the C ABI (std::bad_alloc / std::length_error on a corrupt archive): -2 like every other malformed input
int mvsf_load(const char* path, mvsf_scene** out) {
	if (!path || !out) return -1;
	*out = nullptr;
	try { return loadScene(path, out); } catch (...) { *out = nullptr; return -2; }
}
static int loadScene(const char* path, mvsf_scene** out) {
	FILE* f = fopen(path, "rb");
	if (!f) return -2;
	std::vector<unsigned char> buf;
	fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
	if (sz <= 0) { fclose(f); return -2; }
	buf.resize((size_t)sz);
	const bool rd = fread(buf.data(), 1, buf.size(), f) == buf.size();
	fclose(f);
	if (!rd) return -2;
	Reader r{buf.data(), buf.size()};
	std::unique_ptr<mvsf_scene> holder(new mvsf_scene());
	mvsf_scene* s = holder.get();
	if (buf.size() >= 4 && memcmp(buf.data(), "MVSI", 4) == 0) {
		r.skip(4); s->version = r.u32(); r.u32();
		if (s->version > 7) return -2;
	} else {
		const std::string p(path);
		std::string ext = p.size() >= 4 ? p.substr(p.size() - 4) : std::string();
		for (char& c : ext) c = (char)tolower(c);
		if (ext != ".mvs") return -2;
		s->version = 0;
	}
	const uint32_t v = s->version;
	const uint64_t nPl = r.u64();
	for (uint64_t i = 0; i < nPl && r.ok; ++i) {
		Platform pl; pl.name = r.str();
		const uint64_t nc = r.u64();
		for (uint64_t c = 0; c < nc && r.ok; ++c) {
			Cam cam; cam.name = r.str();
			if (v > 3) cam.band = r.str();
			if (v > 0) { cam.w = r.u32(); cam.h = r.u32(); }
			r.take(cam.K, 72); r.take(cam.R, 72); r.take(cam.C, 24);
			pl.cams.push_back(cam);
		}
		const uint64_t np = r.u64();
		if (!r.ok || np > (r.n - r.o) / 96) { r.ok = false; break; }
		pl.poses.resize((size_t)np * 12);
		r.take(pl.poses.data(), (size_t)np * 96);
		s->platforms.push_back(pl);
	}
	const uint64_t nIm = r.ok ? r.u64() : 0;
	for (uint64_t i = 0; i < nIm && r.ok; ++i) {
		Image im; im.name = r.str();
		if (v > 4) im.mask = r.str();
		im.platform = r.u32(); im.camera = r.u32(); im.pose = r.u32();
		if (v > 2) im.id = r.u32();
		if (v > 6) {
			float d[3] = {0, 0, 0}; r.take(d, 12); im.minDepth = d[0]; im.avgDepth = d[1]; im.maxDepth = d[2];
			const uint64_t ns = r.u64();
			if (!r.ok || ns > (r.n - r.o) / sizeof(MVSFViewScore)) { r.ok = false; break; }
			im.neighbors.resize((size_t)ns);
			if (ns) r.take(im.neighbors.data(), (size_t)ns * sizeof(MVSFViewScore));
		}
		s->images.push_back(im);
	}
	const uint64_t nV = r.ok ? r.u64() : 0;
	if (r.ok && nV <= (r.n - r.o) / 20) {
		s->X.resize((size_t)nV * 3); s->viewStart.assign(1, 0);
		for (uint64_t i = 0; i < nV && r.ok; ++i) {
			r.take(&s->X[(size_t)i * 3], 12);
			const uint64_t m = r.u64();
			if (!r.ok || m > (r.n - r.o) / 8) { r.ok = false; break; }
			for (uint64_t k = 0; k < m; ++k) {
				const uint32_t view = r.u32(); r.skip(4);
				if (view >= s->images.size()) { r.ok = false; break; }   // indexes cams[] / scores[] in selectNeighborViews
				s->views.push_back(view);
			}
			s->viewStart.push_back((uint32_t)s->views.size());
		}
	} else r.ok = false;
	if (r.ok) { const uint64_t a = r.u64(); r.skipn(a, 12); const uint64_t b = r.u64(); r.skipn(b, 3); }
	if (r.ok && v > 0) {
		const uint64_t nL = r.u64();
		for (uint64_t i = 0; i < nL && r.ok; ++i) { r.skip(24); const uint64_t m = r.u64(); r.skipn(m, 8); }
		const uint64_t a = r.u64(); r.skipn(a, 12); const uint64_t b = r.u64(); r.skipn(b, 3);
		if (v > 1) { r.skip(128); if (v > 5) { r.take(s->obbRot, 72); r.take(s->obbMin, 24); r.take(s->obbMax, 24); } }
	}
	for (const Image& im : s->images) for (const MVSFViewScore& n : im.neighbors) if (n.ID >= s->images.size()) r.ok = false;   // a view score names an image of this scene
	// an image that names a platform / camera / pose the archive does not hold is uncalibrated (pixelCamera checks the ranges again); one
	// whose platform index is not even NO_ID-or-valid is a corrupt file
	for (const Image& im : s->images)
		if (im.pose != NO_ID && (im.platform >= s->platforms.size() || im.camera >= s->platforms[im.platform].cams.size())) r.ok = false;
	if (!r.ok) return -2;
	*out = holder.release();
	return 0;
}
void mvsf_free(mvsf_scene* s) { delete s; }
int mvsf_version(const mvsf_scene* s) { return s ? (int)s->version : -1; }
int mvsf_num_images(const mvsf_scene* s) { return s ? (int)s->images.size() : -1; }
int mvsf_num_points(const mvsf_scene* s) { return s ? (int)(s->viewStart.size() - 1) : -1; }

int mvsf_image_info(const mvsf_scene* s, int idx, char* name, int nameCap, int* w, int* h, int* valid) {
	if (!s || idx < 0 || idx >= (int)s->images.size()) return -1;
	const Image& im = s->images[idx];
	if (name && nameCap > 0) { strncpy(name, im.name.c_str(), (size_t)nameCap - 1); name[nameCap - 1] = 0; }
	PixCam pc; const bool ok = pixelCamera(*s, idx, 0, 0, pc);
	if (w) *w = ok ? pc.w : 0; if (h) *h = ok ? pc.h : 0;
	if (valid) *valid = im.pose != NO_ID ? 1 : 0;
	return 0;
}
int mvsf_point(const mvsf_scene* s, int i, float X[3], uint32_t* views, int viewsCap, int* nViews) {
	if (!s || i < 0 || i + 1 >= (int)s->viewStart.size()) return -1;
	if (X) memcpy(X, &s->X[(size_t)i * 3], 12);
	const int n = (int)(s->viewStart[i + 1] - s->viewStart[i]);
	if (nViews) *nViews = n;
	if (views) for (int k = 0; k < n && k < viewsCap; ++k) views[k] = s->views[s->viewStart[i] + k];
	return 0;
}
int mvsf_camera(const mvsf_scene* s, int idx, int w, int h, double K[9], double R[9], double C[3]) {
	if (!s || idx < 0 || idx >= (int)s->images.size() || !K || !R || !C) return -1;
	PixCam pc;
	if (!pixelCamera(*s, idx, w, h, pc)) return -3;
	memcpy(K, pc.K, 72); memcpy(R, pc.R, 72); memcpy(C, pc.C, 24);
	return 0;
}

int mvsf_select_neighbor_views(const mvsf_scene* s, int idx, const int* sizes, uint32_t nMinViews, uint32_t nMinPointViews, float fOptimAngleDeg, uint32_t nInsideROI,
		MVSFViewScore* neighbors, int cap, int* nNeighbors, uint32_t* points, int pointsCap, int* nPoints, float* avgDepth) {
	if (!s || idx < 0 || idx >= (int)s->images.size()) return -1;
	std::vector<PixCam> cams; int nCal;
	if (!camerasFor(*s, sizes, cams, nCal) || !cams[idx].valid) return -3;
	std::vector<MVSFViewScore> nb; std::vector<uint32_t> pts; float avg = 0;
	const bool ok = selectNeighborViews(*s, cams, nCal, (uint32_t)idx, nMinViews, nMinPointViews, fOptimAngleDeg * (3.14159265358979323846f / 180.f), nInsideROI, nb, pts, avg);
	if (nNeighbors) *nNeighbors = (int)nb.size(); if (nPoints) *nPoints = (int)pts.size(); if (avgDepth) *avgDepth = avg;
	if (neighbors) for (int i = 0; i < (int)nb.size() && i < cap; ++i) neighbors[i] = nb[i];
	if (points) for (int i = 0; i < (int)pts.size() && i < pointsCap; ++i) points[i] = pts[i];
	return ok ? 0 : -3;
}

int mvsf_select_views(const mvsf_scene* s, int idx, const int* sizes, const MVSFOptions* o, MVSFViewScore* neighbors, int cap, int* nNeighbors,
		uint32_t* points, int pointsCap, int* nPoints, float* avgDepth) {
	if (!s || !o || idx < 0 || idx >= (int)s->images.size()) return -1;
	std::vector<PixCam> cams; int nCal;
	if (!camerasFor(*s, sizes, cams, nCal) || !cams[idx].valid) return -3;
	std::vector<MVSFViewScore> nb; std::vector<uint32_t> pts; float avg = 0;
	const float d2r = 3.14159265358979323846f / 180.f;
	// DepthMapsData::SelectViews, SceneDensify.cpp:278-281: a list the scene already carries is taken as it is (no seed points then)
	if (!s->images[idx].neighbors.empty()) { nb = s->images[idx].neighbors; avg = s->images[idx].avgDepth; }
	else if (!selectNeighborViews(*s, cams, nCal, (uint32_t)idx, o->nMinViews, o->nMinViewsTrustPoint > 1 ? o->nMinViewsTrustPoint : 2, o->fOptimAngle * d2r, o->nPointInsideROI, nb, pts, avg)) return -3;
	filterNeighborViews(nb, o->fMinArea, 0.2f, 3.2f, o->fMinAngle * d2r, o->fMaxAngle * d2r, o->nMaxViews);   // DepthMapsData::SelectViews, SceneDensify.cpp:283-292
	if (nb.empty()) return -3;
	const float fMinScore = std::max(nb[0].score * o->fViewMinScoreRatio, o->fViewMinScore);                   // InitViews, SceneDensify.cpp:333-340
	size_t cut = nb.size();
	for (size_t i = 0; i < nb.size(); ++i) if ((o->nNumViews && i + 1 > o->nNumViews) || nb[i].score < fMinScore) { cut = i; break; }
	nb.resize(cut);
	if (nb.empty()) return -3;
	if (nNeighbors) *nNeighbors = (int)nb.size(); if (nPoints) *nPoints = (int)pts.size(); if (avgDepth) *avgDepth = avg;
	if (neighbors) for (int i = 0; i < (int)nb.size() && i < cap; ++i) neighbors[i] = nb[i];
	if (points) for (int i = 0; i < (int)pts.size() && i < pointsCap; ++i) points[i] = pts[i];
	return 0;
}

// TriangulatePointsDelaunay (libs/MVS/DepthMap.cpp:1019-1115): projections, camera-space vertices and the Delaunay faces (stored reversed, :1110) of
// the sparse points, in a canonical face order (smallest index first -- a rotation keeps the winding -- then sorted; the order only matters for
// pixels exactly on a shared edge).  withCorners: the four image corners as extra vertices at a depth extrapolated from the nearby faces (:1050-1107).
struct PointMesh { std::vector<float> proj, vert; std::vector<std::array<int,3>> faces; float zMin = 3.402823466e+38f, zMax = 0.f; };
static inline void i2c(const PixCam& cam, float x, float y, float d, float* o) {         // TransformPointI2C, Camera.h:338-344
	o[0] = (float)((x - cam.K[2]) * d / cam.K[0]); o[1] = (float)((y - cam.K[5]) * d / cam.K[4]); o[2] = d;
}
static void triangulatePoints(const mvsf_scene& s, const PixCam& cam, const uint32_t* points, int nPoints, bool withCorners, float avgDepth, PointMesh& m) {
	const int w = cam.w, h = cam.h;
	std::vector<double> xy((size_t)nPoints * 2);
	m.proj.resize((size_t)nPoints * 2); m.vert.resize((size_t)nPoints * 3);
	for (int i = 0; i < nPoints; ++i) {
		const float* X = &s.X[(size_t)points[i] * 3];
		const float q0 = (float)(cam.P[0] * X[0] + cam.P[1] * X[1] + cam.P[2] * X[2] + cam.P[3]);
		const float q1 = (float)(cam.P[4] * X[0] + cam.P[5] * X[1] + cam.P[6] * X[2] + cam.P[7]);
		const float z = (float)(cam.P[8] * X[0] + cam.P[9] * X[1] + cam.P[10] * X[2] + cam.P[11]);
		const float x = q0 / z, y = q1 / z;
		m.proj[2*i] = x; m.proj[2*i+1] = y; xy[2*i] = x; xy[2*i+1] = y;
		i2c(cam, x, y, z, &m.vert[3*i]);
		m.zMin = std::min(m.zMin, z); m.zMax = std::max(m.zMax, z);
	}
	const bool corners = withCorners && nPoints >= 3;
	if (corners) {
		const float cxy[4][2] = {{0.f, 0.f}, {(float)(w - 1), 0.f}, {0.f, (float)(h - 1)}, {(float)(w - 1), (float)(h - 1)}};
		for (const auto& c : cxy) {
			m.proj.push_back(c[0]); m.proj.push_back(c[1]); xy.push_back(c[0]); xy.push_back(c[1]);
			float v[3]; i2c(cam, c[0], c[1], avgDepth, v); m.vert.insert(m.vert.end(), v, v + 3);
		}
	}
	std::vector<Tri> tris; delaunay(xy, tris);
	m.faces.clear(); m.faces.reserve(tris.size());
	for (const Tri& t : tris) {
		std::array<int,3> f = {t.v[2], t.v[1], t.v[0]};
		const int r = (int)(std::min_element(f.begin(), f.end()) - f.begin());
		m.faces.push_back({f[r], f[(r+1)%3], f[(r+2)%3]});
	}
	std::sort(m.faces.begin(), m.faces.end());
	if (!corners) return;
	std::map<std::pair<int,int>, std::vector<int>> edges;                                  // undirected edge -> faces
	for (int fi = 0; fi < (int)m.faces.size(); ++fi) for (int k = 0; k < 3; ++k) {
		const int a = m.faces[fi][k], b = m.faces[fi][(k+1)%3];
		edges[{std::min(a, b), std::max(a, b)}].push_back(fi);
	}
	for (int ci = nPoints; ci < nPoints + 4; ++ci) {
		const double posA[2] = {m.proj[2*ci], m.proj[2*ci+1]};
		double d[3] = {m.vert[3*ci], m.vert[3*ci+1], m.vert[3*ci+2]};
		const double dn = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
		for (double& v : d) v /= dn;                                                       // Ray3d(0, normalized(vertex))
		std::vector<std::pair<float,float>> top;                                           // (score, depth)
		for (int fi = 0; fi < (int)m.faces.size(); ++fi) {
			const auto& f = m.faces[fi];
			if (f[0] != ci && f[1] != ci && f[2] != ci) continue;
			int o[2], no = 0; for (int v : f) if (v != ci) o[no++] = v;
			const auto& sh = edges[{std::min(o[0], o[1]), std::max(o[0], o[1])}];
			int nb = -1; for (int g : sh) if (g != fi) { nb = g; break; }
			if (nb < 0) continue;                                                          // hull edge: the neighbour is the infinite face
			const auto& fb = m.faces[nb];
			double p[3][3]; for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) p[k][c] = m.vert[3*fb[k]+c];
			const double e1[3] = {p[1][0]-p[0][0], p[1][1]-p[0][1], p[1][2]-p[0][2]}, e2[3] = {p[2][0]-p[0][0], p[2][1]-p[0][1], p[2][2]-p[0][2]};
			double nr[3] = {e1[1]*e2[2]-e1[2]*e2[1], e1[2]*e2[0]-e1[0]*e2[2], e1[0]*e2[1]-e1[1]*e2[0]};
			const double nn = sqrt(nr[0]*nr[0] + nr[1]*nr[1] + nr[2]*nr[2]);
			if (nn == 0) continue;
			for (double& v : nr) v /= nn;
			const double Vd = nr[0]*d[0] + nr[1]*d[1] + nr[2]*d[2];
			const double t = Vd == 0 ? 0.0 : (nr[0]*p[0][0] + nr[1]*p[0][1] + nr[2]*p[0][2]) / Vd;   // TRay::IntersectsDist, libs/Common/Ray.inl:600-610
			const double zB = d[2] * t;
			if (zB <= 0) continue;
			const double bx = ((double)m.proj[2*fb[0]] + m.proj[2*fb[1]] + m.proj[2*fb[2]]) / 3.0, by = ((double)m.proj[2*fb[0]+1] + m.proj[2*fb[1]+1] + m.proj[2*fb[2]+1]) / 3.0;
			const double dist = sqrt((bx - posA[0]) * (bx - posA[0]) + (by - posA[1]) * (by - posA[1]));
			top.emplace_back(1.f / (float)dist, std::min(std::max((float)zB, m.zMin), m.zMax));
		}
		std::stable_sort(top.begin(), top.end(), [](const std::pair<float,float>& a, const std::pair<float,float>& b) { return a.first > b.first; });
		if (top.size() > 3) top.resize(3);
		if (top.empty()) continue;                                                         // (the reference asserts three)
		float sum = 0.f; for (const auto& e : top) sum += e.first;
		const float inv = 1.f / sum;
		float depth = 0.f; for (const auto& e : top) depth += (e.first * inv) * e.second;
		i2c(cam, m.proj[2*ci], m.proj[2*ci+1], depth, &m.vert[3*ci]);
	}
}

// One face of the dense initialisation: TImage::RasterizeTriangleBary (libs/Common/Types.inl:2629-2669) driving the RasterDepth functor of
// TriangulatePoints2DepthMap (libs/MVS/DepthMap.cpp:1159-1187): perspective-correct barycentric depth and normal at every covered pixel centre.
// EdgeFunction, Util.inl:602-604: (x2 - x0).cross(x1 - x0) -- the differences in float, the cross product in double (cv::Point_::cross returns double), the result a float
static inline float edgeFn(const float* a, const float* b, const float* c) {
	const float dx2 = c[0] - a[0], dy2 = c[1] - a[1], dx1 = b[0] - a[0], dy1 = b[1] - a[1];
	return (float)((double)dx2 * dy1 - (double)dy2 * dx1);
}
static void rasterFace(const float* v1, const float* v2, const float* v3, float z0, float z1, float z2, const float* n0, const float* n1, const float* n2,
		int w, int h, float* depthMap, float* normalMap) {
	const float mnx = std::min(v1[0], std::min(v2[0], v3[0])), mxx = std::max(v1[0], std::max(v2[0], v3[0]));
	const float mny = std::min(v1[1], std::min(v2[1], v3[1])), mxy = std::max(v1[1], std::max(v2[1], v3[1]));
	if (mxx < 0 || mnx > (float)(w - 1) || mxy < 0 || mny > (float)(h - 1)) return;
	const int x0 = std::max((int)floorf(mnx), 0), x1 = std::min((int)ceilf(mxx), w - 1), y0 = std::max((int)floorf(mny), 0), y1 = std::min((int)ceilf(mxy), h - 1);
	const float area = edgeFn(v1, v2, v3);
	if (area <= 0) return;
	const float inv = 1.f / area;
	for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
		const float p[2] = {(float)x, (float)y};
		const float b1 = edgeFn(v2, v3, p) * inv; if (b1 < 0) continue;
		const float b2 = edgeFn(v3, v1, p) * inv; if (b2 < 0) continue;
		const float b3 = edgeFn(v1, v2, p) * inv; if (b3 < 0) continue;
		float pb[3] = {b1 * z1 * z2, b2 * z0 * z2, b3 * z0 * z1};                          // PerspectiveCorrectBarycentricCoordinates, Util.inl:746-749
		const float invd = 1.f / (pb[0] + pb[1] + pb[2]);
		for (float& v : pb) v = invd * v;
		depthMap[(size_t)y * w + x] = pb[0] * z0 + pb[1] * z1 + pb[2] * z2;
		if (!normalMap) continue;
		float n[3];
		for (int k = 0; k < 3; ++k) n[k] = n0[k] * pb[0] + n1[k] * pb[1] + n2[k] * pb[2];
		const double nn = sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
		const double in = nn ? 1.0 / nn : 0.0;
		for (int k = 0; k < 3; ++k) normalMap[((size_t)y * w + x) * 3 + k] = (float)(n[k] * in);
	}
}

static int initDepthMap(const mvsf_scene* s, int idx, int w, int h, const uint32_t* points, int nPoints, uint32_t nMinViewsTrustPoint, bool dense,
		float* depthMap, float* normalMap, float* dMin, float* dMax) {
	if (!s || idx < 0 || idx >= (int)s->images.size() || !depthMap || !normalMap || !dMin || !dMax || nPoints < 0 || (nPoints && !points)) return -1;
	PixCam cam;
	if (!pixelCamera(*s, idx, w, h, cam)) return -3;
	w = cam.w; h = cam.h;
	memset(depthMap, 0, sizeof(float) * (size_t)w * h); memset(normalMap, 0, sizeof(float) * 3 * (size_t)w * h);
	if (nPoints == 0) { *dMin = 1e-1f; *dMax = 1e+2f; return 0; }                       // SceneDensify.cpp:424-427
	const int nAll = mvsf_num_points(s);
	for (int i = 0; i < nPoints; ++i) if ((int)points[i] >= nAll) return -1;
	float mn = 3.402823466e+38f, mx = 0.f;
	if (nMinViewsTrustPoint < 2) {                                                        // :428-452: 5x5 splats, zero normals
		for (int i = 0; i < nPoints; ++i) {
			const float* X = &s->X[(size_t)points[i] * 3];
			double cx[3];
			for (int r = 0; r < 3; ++r) cx[r] = cam.R[r*3] * (X[0] - cam.C[0]) + cam.R[r*3+1] * (X[1] - cam.C[1]) + cam.R[r*3+2] * (X[2] - cam.C[2]);
			const int px = (int)floor(cam.K[0] * cx[0] / cx[2] + cam.K[2] + 0.5), py = (int)floor(cam.K[4] * cx[1] / cx[2] + cam.K[5] + 0.5);
			const float d = (float)cx[2];
			for (int y = std::max(py - 2, 0); y <= std::min(py + 2, h - 1); ++y) for (int x = std::max(px - 2, 0); x <= std::min(px + 2, w - 1); ++x) depthMap[(size_t)y * w + x] = d;
			mn = std::min(mn, d); mx = std::max(mx, d);
		}
		*dMin = mn * 0.9f; *dMax = mx * 1.1f;
		return 0;
	}
	// TriangulatePoints2DepthMap, DepthMap.cpp:1117-1192
	PointMesh m;
	triangulatePoints(*s, cam, points, nPoints, false, 0.f, m);
	const std::vector<float>& proj = m.proj; const std::vector<float>& vert = m.vert;
	std::vector<float> nrm((size_t)nPoints * 3, 0.f);
	for (const auto& f : m.faces) {                                                       // Mesh::ComputeNormalVertices, Mesh.cpp:356-371
		const int f0 = f[0], f1 = f[1], f2 = f[2];
		const float a[3] = {vert[3*f1] - vert[3*f0], vert[3*f1+1] - vert[3*f0+1], vert[3*f1+2] - vert[3*f0+2]};
		const float b[3] = {vert[3*f2] - vert[3*f0], vert[3*f2+1] - vert[3*f0+1], vert[3*f2+2] - vert[3*f0+2]};
		const float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
		for (int v : {f0, f1, f2}) for (int k = 0; k < 3; ++k) nrm[3*v+k] += c[k];
	}
	for (int i = 0; i < nPoints; ++i) {
		const double nn = sqrt((double)nrm[3*i] * nrm[3*i] + (double)nrm[3*i+1] * nrm[3*i+1] + (double)nrm[3*i+2] * nrm[3*i+2]);
		const double inv = nn ? 1.0 / nn : 0.0;
		for (int k = 0; k < 3; ++k) nrm[3*i+k] = (float)(nrm[3*i+k] * inv);
	}
	*dMin = m.zMin * 0.9f; *dMax = m.zMax * 1.1f;
	if (dense) {                                                                          // DepthMap.cpp:1158-1190
		for (const auto& f : m.faces) rasterFace(&proj[2*f[0]], &proj[2*f[1]], &proj[2*f[2]], vert[3*f[0]+2], vert[3*f[1]+2], vert[3*f[2]+2],
		                                         &nrm[3*f[0]], &nrm[3*f[1]], &nrm[3*f[2]], w, h, depthMap, normalMap);
		return 0;
	}
	for (int i = 0; i < nPoints; ++i) {
		const int ix = (int)floorf(proj[2*i]), iy = (int)floorf(proj[2*i+1]);
		for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {                   // (0,0),(1,0),(0,1),(1,1): the same 4 pixels whatever the order
			const int ax = ix + dx, ay = iy + dy;
			if (ax < 0 || ay < 0 || ax >= w || ay >= h) continue;
			depthMap[(size_t)ay * w + ax] = vert[3*i+2];
			for (int k = 0; k < 3; ++k) normalMap[((size_t)ay * w + ax) * 3 + k] = nrm[3*i+k];
		}
	}
	return 0;
}

int mvsf_init_depth_map(const mvsf_scene* s, int idx, int w, int h, const uint32_t* points, int nPoints, uint32_t nMinViewsTrustPoint,
		float* depthMap, float* normalMap, float* dMin, float* dMax) {
	return initDepthMap(s, idx, w, h, points, nPoints, nMinViewsTrustPoint, false, depthMap, normalMap, dMin, dMax);
}
int mvsf_triangulate_depth_map(const mvsf_scene* s, int idx, int w, int h, const uint32_t* points, int nPoints, int addCorners, float avgDepth, int sparseOnly,
		float* depthMap, float* dMin, float* dMax) {
	if (!s || idx < 0 || idx >= (int)s->images.size() || !depthMap || !dMin || !dMax || nPoints < 0 || (nPoints && !points)) return -1;
	PixCam cam;
	if (!pixelCamera(*s, idx, w, h, cam)) return -3;
	w = cam.w; h = cam.h;
	const int nAll = mvsf_num_points(s);
	for (int i = 0; i < nPoints; ++i) if ((int)points[i] >= nAll) return -1;
	memset(depthMap, 0, sizeof(float) * (size_t)w * h);
	PointMesh m;
	triangulatePoints(*s, cam, points, nPoints, addCorners != 0, avgDepth, m);
	*dMin = m.zMin; *dMax = m.zMax;
	if (sparseOnly) {
		for (size_t i = 0; i < m.vert.size() / 3; ++i) {
			const int ix = (int)floorf(m.proj[2*i]), iy = (int)floorf(m.proj[2*i+1]);
			for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
				const int ax = ix + dx, ay = iy + dy;
				if (ax >= 0 && ay >= 0 && ax < w && ay < h) depthMap[(size_t)ay * w + ax] = m.vert[3*i+2];
			}
		}
		return 0;
	}
	for (const auto& f : m.faces) rasterFace(&m.proj[2*f[0]], &m.proj[2*f[1]], &m.proj[2*f[2]], m.vert[3*f[0]+2], m.vert[3*f[1]+2], m.vert[3*f[2]+2],
	                                         nullptr, nullptr, nullptr, w, h, depthMap, nullptr);
	return 0;
}
int mvsf_init_depth_map_dense(const mvsf_scene* s, int idx, int w, int h, const uint32_t* points, int nPoints,
		float* depthMap, float* normalMap, float* dMin, float* dMax) {
	return initDepthMap(s, idx, w, h, points, nPoints, 2, true, depthMap, normalMap, dMin, dMax);
}

// ---- neighbour lists the scene carries (include/mvsfront.h) ------------------------------------------------------------------------------------------------
int mvsf_get_neighbors(const mvsf_scene* s, int idx, MVSFViewScore* neighbors, int cap, int* nNeighbors) {
	if (!s || idx < 0 || idx >= (int)s->images.size()) return -1;
	const std::vector<MVSFViewScore>& nb = s->images[idx].neighbors;
	if (nNeighbors) *nNeighbors = (int)nb.size();
	if (neighbors) for (int i = 0; i < (int)nb.size() && i < cap; ++i) neighbors[i] = nb[i];
	return 0;
}
int mvsf_set_neighbors(mvsf_scene* s, int idx, const MVSFViewScore* neighbors, int nNeighbors) {
	if (!s || idx < 0 || idx >= (int)s->images.size() || nNeighbors < 0 || (nNeighbors && !neighbors)) return -1;
	for (int i = 0; i < nNeighbors; ++i) if (neighbors[i].ID >= s->images.size()) return -1;
	s->images[idx].neighbors.assign(neighbors, neighbors + nNeighbors);
	return 0;
}
int mvsf_image_depths(const mvsf_scene* s, int idx, float* minDepth, float* avgDepth, float* maxDepth) {
	if (!s || idx < 0 || idx >= (int)s->images.size()) return -1;
	const Image& im = s->images[idx];
	if (minDepth) *minDepth = im.minDepth; if (avgDepth) *avgDepth = im.avgDepth; if (maxDepth) *maxDepth = im.maxDepth;
	return 0;
}

namespace {
// an image index: decimal digits only (String::FromString<IIndex> reads through an istream; anything it would turn into NO_ID or garbage is an error here)
bool parseIndex(const std::string& w, size_t nImages, uint32_t& out) {
	if (w.empty() || w.size() > 10) return false;
	unsigned long long v = 0;
	for (const char c : w) { if (c < '0' || c > '9') return false; v = v * 10 + (unsigned)(c - '0'); }
	if (v >= nImages) return false;
	out = (uint32_t)v;
	return true;
}
} // namespace

// Scene::LoadViewNeighbors, libs/MVS/Scene.cpp:413-457
int mvsf_load_view_neighbors(mvsf_scene* s, const char* path) {
	if (!s || !path) return -1;
	try {
		std::vector<std::pair<std::string, std::string>> items;
		if (smltext::rootValues(path, items) < 0) return -2;              // (the reference ignores what its reader returns: the entries in front of a malformed section count)
		std::vector<std::pair<uint32_t, std::vector<MVSFViewScore>>> lists;   // nothing is installed unless the whole file is good
		std::vector<std::string> words;
		for (const auto& it : items) {
			smltext::splitWords(it.second, words);
			if (!words.empty() && words[0][0] == '#') continue;
			if (words.size() < 2) continue;                                  // "Invalid image IDs list": skipped, as there
			uint32_t id;
			if (!parseIndex(words[0], s->images.size(), id)) return -2;
			std::vector<MVSFViewScore> nb(words.size() - 1);
			for (size_t i = 1; i < words.size(); ++i) {
				uint32_t n;
				if (!parseIndex(words[i], s->images.size(), n)) return -2;
				nb[i - 1] = MVSFViewScore{n, 0u, 1.f, 15.f * (3.14159265358979323846f / 180.f), 0.5f, 3.f};   // Scene.cpp:451
			}
			lists.emplace_back(id, std::move(nb));
		}
		for (auto& l : lists) s->images[l.first].neighbors = std::move(l.second);
		return 0;
	} catch (...) { return -2; }
}
// Scene::SaveViewNeighbors, libs/MVS/Scene.cpp:458-480
int mvsf_save_view_neighbors(const mvsf_scene* s, const char* path) {
	if (!s || !path) return -1;
	FILE* f = fopen(path, "wb");
	if (!f) return -2;
	for (size_t id = 0; id < s->images.size(); ++id) {
		fprintf(f, "%u", (unsigned)id);
		for (const MVSFViewScore& n : s->images[id].neighbors) fprintf(f, " %u", (unsigned)n.ID);
		fprintf(f, "\n");
	}
	return fclose(f) == 0 ? 0 : -2;
}

// MVS::EstimateNormalMap, libs/MVS/DepthMap.cpp:1522-1613 (the active least-squares branch)
int mvsf_estimate_normal_map(const double K[9], const float* depth, int w, int h, float* normal) {
	if (!K || !depth || !normal || w <= 0 || h <= 0) return -1;
	const float k00 = (float)K[0], k11 = (float)K[4], k02 = (float)K[2], k12 = (float)K[5];
	for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) {
		float* n = normal + 3 * ((size_t)r * w + c);
		n[0] = n[1] = n[2] = 0.f;
		const float d = depth[(size_t)r * w + c];
		if (d <= 0) continue;
		int sxx = 0, sxy = 0, syy = 0, count = 0;
		float gx = 0, gy = 0;
		for (int y = -1; y <= 1; ++y) for (int x = -1; x <= 1; ++x) {
			if ((x == 0 && y == 0) || c + x < 0 || r + y < 0 || c + x >= w || r + y >= h) continue;
			const float di = depth[(size_t)(r + y) * w + (c + x)];
			if (!(di > 0 && fabsf(d - di) / d < 0.03f)) continue;          // IsDepthSimilar(d, di, 0.03f), libs/Common/Util.inl:797-809
			sxx += x * x; sxy += x * y; syy += y * y;
			gx += (di - d) * (float)x; gy += (di - d) * (float)y;
			++count;
		}
		const int det = sxx * syy - sxy * sxy;
		if (count < 3 || det == 0) continue;
		const float inv = 1.f / (float)det;
		const float dx = ((float)syy * gx - (float)sxy * gy) * inv, dy = ((float)(-sxy) * gx + (float)sxx * gy) * inv;
		const float v[3] = {k00 * dx, k11 * dy, (k02 - (float)c) * dx + (k12 - (float)r) * dy - d};
		const double nv = sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]);      // cv::normalize: v * (1 / |v|), norm and product in double
		const double a = nv ? 1. / nv : 0.;
		n[0] = (float)(v[0] * a); n[1] = (float)(v[1] * a); n[2] = (float)(v[2] * a);
	}
	return 0;
}

} // extern "C"
