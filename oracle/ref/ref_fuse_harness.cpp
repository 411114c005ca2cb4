// ref_fuse_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  The reference's depth-map fusion: DepthMapsData::MergeDepthMaps and DepthMapsData::FuseDepthMaps
// (libs/MVS/SceneDensify.cpp:1305-1366, :1372-1646) with Conf2Weight (:120-122), cut VERBATIM from /root/reference by oracle/ref/build_ref.py and compiled against
// oracle/ref/shim with the reference's own TPixel, camera members and containers' semantics.  Same C interface as oracle/fuse_oracle.cpp (OrcFuseView in, OrcFuseCloud out),
// plus the processing order the reference's own std::sort produced (ties among equally connected images are the standard library's: the oracle takes the order as an input).
// Third-party arithmetic restated because the libraries are absent: cv::normalize(Vec3f) (opencv2/core/matx.hpp: v * (1 / norm), the norm accumulated in double),
// AssembleProjectionMatrix's cv::Mat products (Camera.cpp:173-180).  File access (DepthData::IncRef / DecRef / Save), progress display and EstimateNormalMap (only reached when
// a depth map has no normal map) are stubs.
#define REF_SCENE 1
#define REF_FUSE 1
#include <float.h>
#include "seacave_min.h"
#define DEBUG_EXTRA(...) ((void)0)
#define _T(x) x
#define NO_ID ((uint32_t)-1)               // Common.h:121
namespace SEACAVE {
#include "snip/types_h_indexscore.inc"    // Types.h:2462-2485: TIndexScore (compare by score, decreasing)
typedef TIndexScore<uint32_t, float> IndexScore;
typedef CLISTDEF0(IndexScore) IndexScoreArr;
#include "snip/types_h_cuint32.inc"       // Types.h:2547-2557: cuint32_t
// normalized(TPoint3) -> cv::normalize(Vec3f) (Types.inl:1048-1051; OpenCV restated: see the header)
template <typename TYPE> inline TPoint3<TYPE> normalized(const cv::Point3_<TYPE>& v) {
	const double nv = std::sqrt((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z);
	const double a = nv ? 1. / nv : 0.;
	return TPoint3<TYPE>(cv::saturate_cast<TYPE>(v.x * a), cv::saturate_cast<TYPE>(v.y * a), cv::saturate_cast<TYPE>(v.z * a));
}
namespace Util { struct Progress { Progress(const char*, size_t) {} void display(size_t) {} void close() {} }; }
struct LogConsole { void Pause() {} void Play() {} };
inline LogConsole& GET_LOGCONSOLE() { static LogConsole l; return l; }
}
#include "snip/depthmap_h.inc"            // libs/MVS/DepthMap.h:41-468 (DepthData; opens namespace MVS, closed right below)
} // namespace MVS
namespace MVS {
struct PointCloud {
	typedef TPoint3<float> Point; typedef uint32_t View; typedef float Weight; typedef TPoint3<float> Normal; typedef Pixel8U Color;
	typedef SEACAVE::cList<View, const View, 0, 4, uint32_t> ViewArr;
	typedef SEACAVE::cList<Weight, const Weight, 0, 4, uint32_t> WeightArr;
	SEACAVE::cList<Point> points; SEACAVE::cList<ViewArr> pointViews; SEACAVE::cList<WeightArr> pointWeights; SEACAVE::cList<Normal> normals; SEACAVE::cList<Color> colors;
	size_t GetSize() const { return points.size(); }
};
bool EstimateNormalMap(const Matrix3x3f& K, const DepthMap&, NormalMap&);                       // DepthMap.h:494; the function itself is cut in below (DepthMap.cpp:1522-1613)
typedef Image SceneImage;
class Scene { public: ImageArr images; };
class DepthMapsData {
public:
	DepthMapsData(Scene& s) : scene(s) {}
	void MergeDepthMaps(PointCloud& pointcloud, bool bEstimateColor, bool bEstimateNormal);
	void FuseDepthMaps(PointCloud& pointcloud, bool bEstimateColor, bool bEstimateNormal);
	Scene& scene;
	DepthDataArr arrDepthData;
};
void AssembleProjectionMatrix(const KMatrix& K, const RMatrix& R, const CMatrix& C, PMatrix& P) {   // Camera.cpp:173-180 (cv::Mat products restated)
	const Matrix3x3 M(K * R);
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) P(i, j) = M(i, j);
		REAL s = 0; const REAL c[3] = {-C.x, -C.y, -C.z};
		for (int k = 0; k < 3; ++k) s += M(i, k) * c[k];
		P(i, 3) = s;
	}
}
namespace OPTDENSE { unsigned nMinViewsFuse = 2; float fDepthDiffThreshold = 0.01f, fNormalDiffThreshold = 25.f; }
// the file-backed reference counting of DepthData (DepthMap.cpp:1990-2037): the maps are resident here
unsigned DepthData::IncRef(const String&) { return 1; }
unsigned DepthData::DecRef() { return 1; }
bool DepthData::Save(const String&) const { return true; }
}
using namespace MVS;
#undef ComposeDepthFilePath
#define ComposeDepthFilePath(i, e) String("")
#define TD_TIMER_STARTD() ((void)0)
#define TD_TIMER_GET_FMT() String()
#include "snip/camera_cpp_pointdepth.inc"   // Camera.cpp:112-115
#include "snip/depthmap_cpp_copy.inc"       // DepthMap.cpp:121-133: DepthData's copy constructor
// DepthData::GetNormal (DepthMap.cpp:137-205): its first branch -- the depth map has a normal map (:142-146); the estimate from neighbouring depths is not reached here
void DepthData::GetNormal(const ImageRef& ir, Point3f& N, const TImage<Point3f>*) const {
	const Camera& camera = images.First().camera;
	if (normalMap.empty()) abort();
	N = camera.R.t()*Cast<REAL>(normalMap(ir));
}
// ---- the triangle rasteriser of the dense initialisation: TImage::RasterizeTriangleBary (Types.inl:2625-2669) with EdgeFunction, the perspective-correct barycentric
// coordinates, TRasterMeshBase (Mesh.h:283-325) and the RasterDepth functor of TriangulatePoints2DepthMap (DepthMap.cpp:1156-1178), all verbatim ----
namespace SEACAVE {
#include "snip/types_h_minf3.inc"            // Types.h:346-353: MINF3, MAXF3
#include "snip/util_inl_edge.inc"            // Util.inl:600-604: EdgeFunction
#include "snip/util_inl_perspbary.inc"       // Util.inl:745-749: PerspectiveCorrectBarycentricCoordinates
#include "snip/types_inl_rasterbary.inc"     // Types.inl:2625-2669
}
namespace MVS {
struct Mesh { typedef TPoint3<float> Normal; typedef SEACAVE::cList<Normal> NormalArr; typedef TPoint3<uint32_t> Face; };   // Mesh.h: the three names the functor uses
#include "snip/mesh_h_rasterbase.inc"        // Mesh.h:283-325
#include "snip/depthmap_cpp_rasterdepth.inc" // DepthMap.cpp:1156-1178 (a local struct of TriangulatePoints2DepthMap there)
}
#include "snip/depthmap_cpp_estnormal.inc"   // DepthMap.cpp:1522-1613: EstimateNormalMap (what FuseDepthMaps, the SGM fuse mode and a resumed .dmap without normals call)
#include "snip/scenedensify_conf2weight.inc" // SceneDensify.cpp:119-122: Conf2Weight
#include "snip/scenedensify_fuse.inc"       // SceneDensify.cpp:1303-1646: MergeDepthMaps, FuseDepthMaps

extern "C" {
struct OrcFuseView {                      // same layout as oracle/fuse_oracle.cpp's
	const float* depth; const float* normal; const float* conf; const uint8_t* bgr;
	double K[9], R[9], C[3];
	const uint32_t* neighbors; uint32_t nNeighbors;
	int w, h;                             // this view's own size, 0 = the call's
};
struct OrcFuseCloud {
	uint64_t nPoints, nDepths, nViews;
	float* points; uint32_t* viewStart; uint32_t* views; float* weights; uint16_t* projs; uint8_t* colors; float* normals;
};
// The face loop of TriangulatePoints2DepthMap (DepthMap.cpp:1178-1188) over nFaces triangles: projs = 2-D projections (floats), z = camera-space depths of the vertices,
// normals = vertex normals (3 floats each, or NULL: the depth-only variant, :1229-1247, has the same functor without them); maps are zero where no face lands
void ref_raster_faces(int w, int h, int nVerts, const float* projs, const float* z, const float* normals, int nFaces, const uint32_t* faces, float* depthOut, float* normalOut) {
	Camera camera;
	DepthMap depthMap(cv::Size(w, h)); NormalMap normalMap(cv::Size(w, h));
	memset((void*)depthMap.data(), 0, sizeof(float) * (size_t)w * h); memset((void*)normalMap.data(), 0, sizeof(float) * 3 * (size_t)w * h);
	Mesh::NormalArr vertexNormals; vertexNormals.resize((unsigned)nVerts);
	for (int i = 0; i < nVerts; ++i) vertexNormals[i] = normals ? Mesh::Normal(normals[3*i], normals[3*i+1], normals[3*i+2]) : Mesh::Normal(0, 0, -1);
	RasterDepth rasterer = {vertexNormals, camera, depthMap, normalMap};
	for (int f = 0; f < nFaces; ++f) {
		const Mesh::Face face(faces[3*f], faces[3*f+1], faces[3*f+2]);
		rasterer.face = face;
		rasterer.ptc[0].z = z[face[0]];
		rasterer.ptc[1].z = z[face[1]];
		rasterer.ptc[2].z = z[face[2]];
		Image8U::RasterizeTriangleBary(
			Point2f(projs[2*face[0]], projs[2*face[0]+1]),
			Point2f(projs[2*face[1]], projs[2*face[1]+1]),
			Point2f(projs[2*face[2]], projs[2*face[2]+1]), rasterer);
	}
	memcpy(depthOut, (const void*)depthMap.data(), sizeof(float) * (size_t)w * h);
	if (normalOut) memcpy(normalOut, (const void*)normalMap.data(), sizeof(float) * 3 * (size_t)w * h);
}
// DepthMapsData::InitViews with OPTDENSE::nMinViewsTrustPoint < 2 (SceneDensify.cpp:418-451, the body of that branch verbatim): the depth range of the image's sparse
// points and their depths splatted on 5x5 blocks.  X: all scene points (3 floats each); pts: the indices this image sees
void ref_init_views_splat(const double* K, const double* R, const double* C, int w, int h, const float* X, int nX, const uint32_t* pts, int nPts, float* depthOut, float* normalOut,
		float* dMin, float* dMax) {
	struct { struct { SEACAVE::cList<TPoint3<float>> points; } pointcloud; } scene;
	scene.pointcloud.points.resize((unsigned)nX);
	for (int i = 0; i < nX; ++i) scene.pointcloud.points[i] = TPoint3<float>(X[3*i], X[3*i+1], X[3*i+2]);
	DepthData depthData;
	depthData.points.resize((unsigned)nPts);
	for (int i = 0; i < nPts; ++i) depthData.points[i] = pts[i];
	DepthData::ViewData viewRef;
	for (int k = 0; k < 9; ++k) { viewRef.camera.K.val[k] = K[k]; viewRef.camera.R.val[k] = R[k]; }
	viewRef.camera.C.x = C[0]; viewRef.camera.C.y = C[1]; viewRef.camera.C.z = C[2];
	viewRef.image.create(cv::Size(w, h));
	{
#include "snip/scenedensify_initsplat.inc"
	}
	memcpy(depthOut, (const void*)depthData.depthMap.data(), sizeof(float) * (size_t)w * h);
	memcpy(normalOut, (const void*)depthData.normalMap.data(), sizeof(float) * 3 * (size_t)w * h);
	*dMin = depthData.dMin; *dMax = depthData.dMax;
}
// MVS::EstimateNormalMap: K 9 floats row-major, depth w*h -> normal w*h*3
int ref_estimate_normal_map(const float* K, const float* depth, int w, int h, float* normal) {
	Matrix3x3f Kf; for (int k = 0; k < 9; ++k) Kf.val[k] = K[k];
	DepthMap d; d.create(cv::Size(w, h)); memcpy((void*)d.data(), depth, sizeof(float) * (size_t)w * h);
	NormalMap n;
	if (!EstimateNormalMap(Kf, d, n)) return -1;
	memcpy(normal, (const void*)n.data(), sizeof(float) * 3 * (size_t)w * h);
	return 0;
}
void ref_fuse_free(OrcFuseCloud* c) { free(c->points); free(c->viewStart); free(c->views); free(c->weights); free(c->projs); free(c->colors); free(c->normals); memset(c, 0, sizeof(*c)); }

static void setup(Scene& scene, DepthMapsData& dm, const OrcFuseView* views, int nImages, int w, int h) {
	scene.images.resize((IIndex)nImages);
	dm.arrDepthData.resize((IIndex)nImages);
	const int w0 = w, h0 = h;
	for (int i = 0; i < nImages; ++i) {
		SceneImage& im = scene.images[(IIndex)i]; const OrcFuseView& s = views[i];
		const int w = s.w ? s.w : w0, h = s.h ? s.h : h0;               // every image / depth map of its own size (DepthMapsData::InitViews)
		const cv::Size size(w, h);
		im.ID = (uint32_t)i; im.size = size; im.width = (uint32_t)w; im.height = (uint32_t)h;
		for (int k = 0; k < 9; ++k) { im.camera.K.val[k] = s.K[k]; im.camera.R.val[k] = s.R[k]; }
		im.camera.C.x = s.C[0]; im.camera.C.y = s.C[1]; im.camera.C.z = s.C[2];
		im.camera.ComposeP();
		im.neighbors.resize((IIndex)s.nNeighbors);
		for (uint32_t n = 0; n < s.nNeighbors; ++n) im.neighbors[(IIndex)n].ID = s.neighbors[n];
		if (s.bgr) { im.image.create(size); memcpy((void*)im.image.data(), s.bgr, (size_t)w * h * 3); }
		DepthData& dd = dm.arrDepthData[(IIndex)i];
		if (!s.depth) continue;                                        // DepthData::IsValid() false: no images
		dd.images.resize(1);
		dd.images[0].pImageData = &im; dd.images[0].camera = im.camera;
		dd.neighbors = im.neighbors;
		dd.depthMap.create(size); memcpy(dd.depthMap.data(), s.depth, sizeof(float) * (size_t)w * h);
		if (s.normal) { dd.normalMap.create(size); memcpy((void*)dd.normalMap.data(), s.normal, sizeof(float) * 3 * (size_t)w * h); }
		if (s.conf) { dd.confMap.create(size); memcpy(dd.confMap.data(), s.conf, sizeof(float) * (size_t)w * h); }
		dd.dMin = 0; dd.dMax = 1e30f;
	}
}
static void pack(const PointCloud& pc, bool withWeights, OrcFuseCloud* out) {
	const size_t n = pc.points.size();
	out->nPoints = n;
	out->points = (float*)malloc(12 * (n + 1)); out->viewStart = (uint32_t*)malloc(4 * (n + 1));
	uint64_t nv = 0; for (size_t i = 0; i < n; ++i) nv += pc.pointViews[i].size();
	out->nViews = nv;
	out->views = (uint32_t*)malloc(4 * (nv + 1)); out->weights = (float*)calloc(nv + 1, 4); out->projs = (uint16_t*)calloc(2 * (nv + 1), 2);
	uint32_t at = 0;
	for (size_t i = 0; i < n; ++i) {
		out->points[3 * i] = pc.points[i].x; out->points[3 * i + 1] = pc.points[i].y; out->points[3 * i + 2] = pc.points[i].z;
		out->viewStart[i] = at;
		for (uint32_t v = 0; v < pc.pointViews[i].size(); ++v, ++at) { out->views[at] = pc.pointViews[i][v]; if (withWeights) out->weights[at] = pc.pointWeights[i][v]; }
	}
	out->viewStart[n] = at;
	if (!pc.colors.empty()) { out->colors = (uint8_t*)malloc(3 * n + 8); memcpy(out->colors, (const void*)pc.colors.data(), 3 * n); }
	if (!pc.normals.empty()) { out->normals = (float*)malloc(12 * n + 8); memcpy(out->normals, (const void*)pc.normals.data(), 12 * n); }
}
// DepthMapsData::FuseDepthMaps.  orderOut (nImages entries): the images in the order the reference processed them (its own Sort of the connection scores); *nOrder their number.
int ref_fuse_depth_maps(const OrcFuseView* views, int nImages, int w, int h, uint32_t* orderOut, int* nOrder,
		unsigned nMinViewsFuse, float fDepthDiffThreshold, float fNormalDiffThresholdDeg, int bEstimateColor, int bEstimateNormal, OrcFuseCloud* out) {
	memset(out, 0, sizeof(*out));
	Scene scene; DepthMapsData dm(scene);
	setup(scene, dm, views, nImages, w, h);
	OPTDENSE::nMinViewsFuse = nMinViewsFuse; OPTDENSE::fDepthDiffThreshold = fDepthDiffThreshold; OPTDENSE::fNormalDiffThreshold = fNormalDiffThresholdDeg;
	{	// the order: the same container, scores and Sort() as SceneDensify.cpp:1400-1453
		IndexScoreArr connections((size_t)nImages);
		for (int i = 0; i < nImages; ++i) {
			if (!dm.arrDepthData[(IIndex)i].IsValid()) { connections[(size_t)i].idx = NO_ID; connections[(size_t)i].score = 0; continue; }
			connections[(size_t)i].idx = (uint32_t)i; connections[(size_t)i].score = (float)scene.images[(IIndex)i].neighbors.size();
		}
		connections.Sort();
		while (!connections.empty() && connections.back().score <= 0) connections.pop_back();
		*nOrder = (int)connections.size();
		for (size_t i = 0; i < connections.size(); ++i) orderOut[i] = connections[i].idx;
	}
	PointCloud pc;
	dm.FuseDepthMaps(pc, bEstimateColor != 0, bEstimateNormal != 0);
	pack(pc, true, out);
	return 0;
}
int ref_merge_depth_maps(const OrcFuseView* views, int nImages, int w, int h, int bEstimateColor, int bEstimateNormal, OrcFuseCloud* out) {
	memset(out, 0, sizeof(*out));
	Scene scene; DepthMapsData dm(scene);
	setup(scene, dm, views, nImages, w, h);
	PointCloud pc;
	dm.MergeDepthMaps(pc, bEstimateColor != 0, bEstimateNormal != 0);
	pack(pc, false, out);
	return 0;
}
}
