"""oracle/_ref: the reference's own hot-path sources, compiled where they lie.            *** TEST INFRASTRUCTURE ONLY ***

    python oracle/ref/build_ref.py            # needs /root/reference (this container); the GPU box uses the prebuilt oracle/_ref/*.so

What is compiled is the text of /root/reference (OpenMVS v2.3.0) itself: this script cuts the line ranges listed in SNIPPETS out of the reference's
headers and sources into a scratch directory (deleted afterwards; nothing of the reference is copied into this repository), and compiles them,
unmodified, together with oracle/ref/ref_harness.cpp against oracle/ref/shim/ -- the minimal stand-ins for OpenCV, Eigen and the SEACAVE containers
that the image lacks.  Every range is checked against the first and last line it must start and end with, so a different reference revision fails
loudly instead of compiling something else.

Outputs (git-ignored, shipped to the GPU box with the snapshot):
    oracle/_ref/libref_pm.so        transcendentals routed to csrc/pm_math.h  -> compared bit for bit with oracle/pm_oracle.cpp
    oracle/_ref/libref_pm_libm.so   transcendentals from libm (as a reference binary) -> bounds what the pm_math.h substitution changes
    oracle/_ref/libref_sgm.so       SemiGlobalMatcher::Match (cost volume, 8-path aggregation, WTA) -> compared bit for bit with oracle/sgm_oracle.cpp
"""
import os, shutil, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("OPENMVS_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name: (file relative to the reference, first line, last line, text the first line must contain, text the last line must contain)
SNIPPETS = {
    "types_h_funcs":         ("libs/Common/Types.h", 615, 644, "template<typename T>", "}"),
    "types_h_tests":         ("libs/Common/Types.h", 1185, 1237, "inline bool   ISINFORNAN(float x)", "SAFEDIVIDE"),
    "types_h_tpoint2":       ("libs/Common/Types.h", 1256, 1345, "// 2D point struct", "typedef TPoint2<double> Point2d;"),
    "types_h_tpoint3":       ("libs/Common/Types.h", 1349, 1438, "// 3D point struct", "typedef TPoint3<double> Point3d;"),
    "types_h_tmatrix":       ("libs/Common/Types.h", 1442, 1548, "// matrix struct", "TMatrix<TYPE,m,n>::INF("),
    "types_h_isinside":      ("libs/Common/Types.h", 1617, 1651, "/// Is this coordinate inside the 2D matrix?", "}"),
    "types_inl_round_pt":    ("libs/Common/Types.inl", 573, 588, "// round", "}"),
    "types_inl_normsq":      ("libs/Common/Types.inl", 794, 803, "template <typename TYPE>", "}"),
    "types_inl_norm":        ("libs/Common/Types.inl", 1021, 1033, "template <typename TYPE>", "}"),
    "types_inl_point_ops":   ("libs/Common/Types.inl", 1178, 1366, "// operators", "}"),
    "types_inl_matrix_ops":  ("libs/Common/Types.inl", 1430, 1474, "// TMatrix operators", "}"),
    "types_inl_cast":        ("libs/Common/Types.inl", 1677, 1709, "// Point2", "}"),
    "types_inl_tmatrix9":    ("libs/Common/Types.inl", 1830, 1839, "template <typename TYPE, int m, int n>", "}"),
    "types_inl_abs_pt":      ("libs/Common/Types.inl", 496, 500, "template <typename TYPE>", "}"),
    "types_inl_initto":      ("libs/Common/Types.inl", 671, 676, "// initializing both scalar and matrix variables", "}"),
    "types_inl_tmatrix4":    ("libs/Common/Types.inl", 1787, 1794, "template <typename TYPE, int m, int n>", "}"),
    "types_inl_getpixel":    ("libs/Common/Types.inl", 2253, 2266, "// Find a pixel inside the image", "}"),
    "types_inl_samplesafe_f": ("libs/Common/Types.inl", 2315, 2332, "template <typename TYPE>", "}"),
    "types_h_accumulator":   ("libs/Common/Types.h", 2398, 2458, "// weighted accumulator class that operates on arbitrary types", "};"),
    "util_inl_project22":    ("libs/Common/Util.inl", 387, 393, "// (optimized ProjectVertex for H[3,3] and X[2,1], output pt[2,1])", "} // ProjectVertex_3x3_2_2"),
    "image_cpp_disp2depth":  ("libs/MVS/Image.cpp", 372, 412, "template <typename TYPE>", "}"),
    "image_cpp_depth2disp":  ("libs/MVS/Image.cpp", 423, 433, "// converts the given depth at the un-rectified image coordinates to", "}"),
    "sgm_cpp_range":         ("libs/MVS/SemiGlobalMatcher.cpp", 1350, 1444, "SemiGlobalMatcher::Index SemiGlobalMatcher::Disparity2RangeMap(", "}"),
    "sgm_cpp_conv":          ("libs/MVS/SemiGlobalMatcher.cpp", 1837, 2039, "// Compute the disparity-map for the rectified image from the given depth-map of the un-rectified image;", "}"),
    "sgm_cpp_fuse_pairdata": ("libs/MVS/SemiGlobalMatcher.cpp", 744, 749, "struct PairData {", "};"),
    "sgm_cpp_fuse_loop":     ("libs/MVS/SemiGlobalMatcher.cpp", 795, 848, "// fuse available depth-maps such that for each pixel set its depth as the average of the largest cluster of agreeing depths;", "}"),
    "types_h_tpixel":        ("libs/Common/Types.h", 1874, 1988, "template <typename TYPE>", "};"),
    "types_inl_cast_pixel":  ("libs/Common/Types.inl", 1695, 1699, "// Pixel", "}"),
    "types_h_indexscore":    ("libs/Common/Types.h", 2462, 2485, "// structure used for sorting some indices by their score (decreasing by default)", "};"),
    "types_h_cuint32":       ("libs/Common/Types.h", 2547, 2557, "struct cuint32_t {", "};"),
    "depthmap_cpp_getnormal": ("libs/MVS/DepthMap.cpp", 137, 210, "void DepthData::GetNormal(const ImageRef& ir, Point3f& N, const TImage<Point3f>* pPointMap) const", "} // GetNormal"),
    "scenedensify_conf2weight": ("libs/MVS/SceneDensify.cpp", 119, 122, "// convert the ZNCC score to a weight used to average the fused points", "}"),
    "scenedensify_fuse":     ("libs/MVS/SceneDensify.cpp", 1303, 1646, "// fuse all depth-maps by simply projecting them in a 3D point cloud", "} // FuseDepthMaps"),
    "types_inl_sample":      ("libs/Common/Types.inl", 2270, 2281, "// sample by bilinear interpolation", "}"),
    "types_inl_sample_f":    ("libs/Common/Types.inl", 2296, 2314, "// sample by bilinear interpolation, using only pixels that meet the user condition", "}"),
    "util_inl_project":      ("libs/Common/Util.inl", 380, 386, "// (optimized ProjectVertex for H[3,3] and X[2,1], output pt[3,1])", "} // ProjectVertex_3x3_2_3"),
    "util_inl_angle":        ("libs/Common/Util.inl", 540, 546, "// given two 3D vectors,", "} // ComputeAngle"),
    "util_inl_dir":          ("libs/Common/Util.inl", 752, 766, "// Encodes/decodes a normalized 3D vector in two parameters for the direction", "}"),
    "util_inl_depth":        ("libs/Common/Util.inl", 789, 809, "template<typename T>", "}"),
    "rotation_h_class":      ("libs/Common/Rotation.h", 265, 584, "template <typename TYPE>", "}; // class"),
    "rotation_inl_ctors":    ("libs/Common/Rotation.inl", 519, 558, "template <typename TYPE>", "}"),
    "rotation_inl_set":      ("libs/Common/Rotation.inl", 700, 729, "template <typename TYPE>", "}"),
    "random_h":              ("libs/Common/Random.h", 100, 159, "// Encapsulates state for random number generation", "};"),
    "camera_h_scalek":       ("libs/MVS/Camera.h", 159, 173, "template<typename TYPE>", "}"),
    "camera_h_projectp":     ("libs/MVS/Camera.h", 307, 320, "template <typename TYPE>", "}"),
    "camera_h_isinside":     ("libs/MVS/Camera.h", 402, 405, "template <typename TYPE>", "}"),
    "camera_h_footprint":    ("libs/MVS/Camera.h", 437, 446, "template <typename TYPE>", "}"),
    "camera_cpp_pointdepth": ("libs/MVS/Camera.cpp", 112, 115, "REAL Camera::PointDepth(const Point3& X) const", "} // PointDepth"),
    "scene_cpp_select":      ("libs/MVS/Scene.cpp", 801, 934, "bool Scene::SelectNeighborViews(uint32_t ID, IndexArr& points,", "} // SelectNeighborViews"),
    "scene_cpp_filter":      ("libs/MVS/Scene.cpp", 953, 968, "bool Scene::FilterNeighborViews(ViewScoreArr& neighbors,", "} // FilterNeighborViews"),
    "camera_h_composek":     ("libs/MVS/Camera.h", 106, 122, "// returns the scale used to normalize the intrinsics", "}"),
    "camera_h_scalek1":      ("libs/MVS/Camera.h", 144, 155, "// return scaled K (assuming standard K format)", "}"),
    "camera_h_getk":         ("libs/MVS/Camera.h", 190, 201, "// returns full K and the inverse of K (assuming standard K format)", "}"),
    "platform_cpp_getcamera": ("libs/MVS/Platform.cpp", 43, 54, "// return the normalized absolute camera pose", "} // GetCamera"),
    "camera_h_invk":         ("libs/MVS/Camera.h", 175, 188, "// return K.inv() (assuming standard K format and no shear)", "}"),
    "camera_h_i2c":          ("libs/MVS/Camera.h", 329, 344, "// un-project from image pixel coords to the camera space (z=1 plane by default)", "}"),
    "camera_h_c2w_i2w":      ("libs/MVS/Camera.h", 345, 356, "template <typename TYPE>", "}"),
    "camera_h_c2i":          ("libs/MVS/Camera.h", 368, 374, "// project from the camera z=1 plane to image pixels", "}"),
    "camera_h_c2i3_w2c_w2i": ("libs/MVS/Camera.h", 382, 394, "// project from the camera space to image pixels", "}"),
    "types_h_float2int":     ("libs/Common/Types.h", 916, 963, "template <typename INTTYPE=int>", "}"),
    "plane_inl_distance":    ("libs/Common/Plane.inl", 185, 190, "// Calculate distance to point. Plane normal must be normalized.", "}"),
    "depthmap_h":            ("libs/MVS/DepthMap.h", 41, 468, "// D E F I N E S", "};"),
    "depthmap_cpp_copy":     ("libs/MVS/DepthMap.cpp", 121, 133, "//constructor from reference of DepthData", "{}"),
    "depthmap_cpp_applymask": ("libs/MVS/DepthMap.cpp", 214, 230, "// apply mask to the depth map", "} // ApplyIgnoreMask"),
    "depthmap_cpp":          ("libs/MVS/DepthMap.cpp", 325, 972, "// create the map for converting index to matrix position", "#endif"),
    "sgm_h_defines":         ("libs/MVS/SemiGlobalMatcher.h", 44, 46, "#define SGM_SIMILARITY_WZNCC 1", "#define SGM_SIMILARITY SGM_SIMILARITY_WZNCC"),
    "sgm_h_class":           ("libs/MVS/SemiGlobalMatcher.h", 57, 206, "// An implementation of the popular Semi-Global Matching (SGM) algorithm.", "};"),
    "sgm_cpp_events":        ("libs/MVS/SemiGlobalMatcher.cpp", 438, 492, "enum EVENT_TYPE {", "};"),
    "sgm_cpp_ctor":          ("libs/MVS/SemiGlobalMatcher.cpp", 506, 524, "SemiGlobalMatcher::SemiGlobalMatcher(SgmSubpixelMode _subpixelMode", "}"),
    "sgm_cpp_match":         ("libs/MVS/SemiGlobalMatcher.cpp", 863, 1302, "void SemiGlobalMatcher::Match(const ViewData& leftImage", "}"),
    "sgm_cpp_post":          ("libs/MVS/SemiGlobalMatcher.cpp", 1446, 1811, "// Check for consistency between a left-to-right and right-to-left pair of stereo results;", "}"),
    "scenedensify_cpp":      ("libs/MVS/SceneDensify.cpp", 489, 576, "// initialize the confidence map (NCC score map) with the score of the current estimates", "}"),
    "scenedensify_scale":    ("libs/MVS/SceneDensify.cpp", 578, 601, "DepthData DepthMapsData::ScaleDepthData(const DepthData& inputDeptData, float scale) {", "}"),
    "scenedensify_estimate": ("libs/MVS/SceneDensify.cpp", 616, 805, "bool DepthMapsData::EstimateDepthMap(IIndex idxImage, int nGeometricIter)", "} // EstimateDepthMap"),
    "scenedensify_filters":  ("libs/MVS/SceneDensify.cpp", 809, 1045, "// filter out small depth segments from the given depth map", "} // GapInterpolation"),
    "scenedensify_filterdm": ("libs/MVS/SceneDensify.cpp", 1049, 1299, "// filter depth-map, one pixel at a time, using confidence based fusion or neighbor pixels", "} // FilterDepthMap"),
    # the triangle rasteriser of the dense initialisation (ref_fuse_harness.cpp)
    "types_h_minf3":         ("libs/Common/Types.h", 346, 353, "template<typename T>", "}"),
    "types_h_clip":          ("libs/Common/Types.h", 1653, 1663, "template <typename T, int border=0>", "}"),
    "types_inl_rasterbary":  ("libs/Common/Types.inl", 2625, 2669, "// same as above, but raster a triangle using barycentric coordinates:", "}"),
    "util_inl_edge":         ("libs/Common/Util.inl", 600, 604, "// compute area for a triangle defined by three 2D points", "}"),
    "util_inl_perspbary":    ("libs/Common/Util.inl", 745, 749, "template <typename TYPE>", "}"),
    "mesh_h_rasterbase":     ("libs/MVS/Mesh.h", 283, 325, "// used to render a 3D triangle", "};"),
    "depthmap_cpp_rasterdepth": ("libs/MVS/DepthMap.cpp", 1156, 1178, "struct RasterDepth : TRasterMeshBase<RasterDepth> {", "};"),
    "scenedensify_initsplat": ("libs/MVS/SceneDensify.cpp", 418, 451, "// compute depth range and initialize known depths, else random", "}"),
    "depthmap_cpp_estnormal": ("libs/MVS/DepthMap.cpp", 1522, 1613, "bool MVS::EstimateNormalMap(const Matrix3x3f& K, const DepthMap& depthMap, NormalMap& normalMap)", "} // EstimateNormalMap"),
    # the two text files in front of the path (ref_text_harness.cpp)
    "util_h_flags":          ("libs/Common/Util.h", 55, 89, "template <typename TYPE>", "typedef class GENERAL_API TFlags<uint32_t> Flags;"),
    "sml_cpp":               ("libs/Common/SML.cpp", 11, 419, "using namespace SEACAVE;", "/*----------------------------------------------------------------*/"),
    "configtable_cpp":       ("libs/Common/ConfigTable.cpp", 11, 153, "using namespace SEACAVE;", "/*----------------------------------------------------------------*/"),
    "util_cpp_argv":         ("libs/Common/Util.cpp", 740, 805, "LPSTR* Util::CommandLineToArgvA(LPCSTR CmdLine, size_t& _argc)", "/*----------------------------------------------------------------*/"),
    "common_h_defvar":       ("libs/Common/Common.h", 101, 170, "// macros simplifying the task of managing options", "#define TDEFVAR_float(SPACE, name, title, desc, ...)"),
    "scene_cpp_loadnb":      ("libs/MVS/Scene.cpp", 423, 457, "bool Scene::LoadViewNeighbors(const String& fileName)", "} // LoadViewNeighbors"),
    "scene_cpp_savenb":      ("libs/MVS/Scene.cpp", 458, 479, "bool Scene::SaveViewNeighbors(const String& fileName) const", "} // SaveViewNeighbors"),
    "types_inl_convert":     ("libs/Common/Types.inl", 1590, 1659, "namespace CONVERT {", "} // namespace CONVERT"),
    "types_inl_computeresize": ("libs/Common/Types.inl", 2437, 2477, "// compute scaled size such that the biggest dimension is scaled as desired", "}"),
    "image_cpp_resize":      ("libs/MVS/Image.cpp", 139, 154, "float Image::ResizeImage(unsigned nMaxResolution)", "} // ResizeImage"),
    "depthmap_cpp_optdense": ("libs/MVS/DepthMap.cpp", 50, 115, "#define DEFVAR_OPTDENSE_string(name, title, desc, ...)", "}"),
}

# whole headers of libs/Common, cut under their own names into <scratch>/common/ so that their #includes of each other resolve there (ref_text_harness.cpp):
# file: (number of lines, text the second line must contain, text the last line must contain)
WHOLE = {
    "libs/Common/AutoPtr.h":     (275, "// AutoPtr.h", "#endif // __SEACAVE_AUTOPTR_H__"),
    "libs/Common/Streams.h":     (244, "// Streams.h", "#endif // __SEACAVE_STREAMS_H__"),
    "libs/Common/Strings.h":     (239, "// Strings.h", "#endif // __SEACAVE_STRING_H__"),
    "libs/Common/List.h":        (1704, "// List.h", "#endif // __SEACAVE_LIST_H__"),
    "libs/Common/Hash.h":        (303, "// Hash.h", "#endif // __SEACAVE_HASH_H__"),
    "libs/Common/File.h":        (686, "// File.h", "#endif // __SEACAVE_FILE_H__"),
    "libs/Common/MemFile.h":     (182, "// MemFile.h", "#endif // __SEACAVE_MEMFILE_H__"),
    "libs/Common/Filters.h":     (612, "// Filters.h", "#endif // __SEACAVE_FILTERS_H__"),
    "libs/Common/SML.h":         (116, "// SML.h", "#endif // __SEACAVE_SML_H__"),
    "libs/Common/ConfigTable.h": (82, "// ConfigTable.h", "#endif // __SEACAVE_CONFIGTABLE_H__"),
}


ALL = ("libref_pm.so", "libref_pm_libm.so", "libref_sgm.so", "libref_scene.so", "libref_fuse.so", "libref_driver.so", "libref_driver_libm.so", "libref_text.so")


def cut(dst):
    os.makedirs(os.path.join(dst, "snip"), exist_ok=True)
    for name, (rel, a, b, first, last) in SNIPPETS.items():
        lines = open(os.path.join(REF, rel), encoding="utf-8", errors="replace").read().split("\n")
        body = lines[a - 1:b]
        if first not in body[0] or last not in body[-1]:
            raise SystemExit("%s:%d-%d does not start / end as expected (%r ... %r): not the reference revision this recipe was written for" % (rel, a, b, body[0], body[-1]))
        with open(os.path.join(dst, "snip", name + ".inc"), "w") as f:
            f.write("#line %d \"%s\"\n" % (a, os.path.join(REF, rel)))
            f.write("\n".join(body) + "\n")
    os.makedirs(os.path.join(dst, "common"), exist_ok=True)
    for rel, (n, second, last) in WHOLE.items():
        lines = open(os.path.join(REF, rel), encoding="utf-8", errors="replace").read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()
        if len(lines) != n or second not in lines[1] or last not in lines[-1]:
            raise SystemExit("%s is not the file this recipe was written for (%d lines, %r ... %r)" % (rel, len(lines), lines[1], lines[-1]))
        with open(os.path.join(dst, "common", os.path.basename(rel)), "w") as f:
            f.write("#line 1 \"%s\"\n" % os.path.join(REF, rel))
            f.write("\n".join(lines) + "\n")


def build(verbose=False):
    if not os.path.isdir(REF):
        return [p for p in (os.path.join(OUT, n) for n in ALL) if os.path.exists(p)]
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="refsnip_")
    outs = []
    try:
        cut(tmp)
        from oracle import pyoracle
        pyoracle.build()                                                   # libref_driver*.so resolve the oracle's resamplers (cv::resize stand-in) from oracle/libpm_oracle.so
        link_orc = ["-pthread", "-L", os.path.join(ROOT, "oracle"), "-l:libpm_oracle.so", "-Wl,-rpath,$ORIGIN/.."]
        for name, flags, src in (("libref_pm.so", ["-DREF_MATH_PM"], "ref_harness.cpp"), ("libref_pm_libm.so", [], "ref_harness.cpp"),
                                 ("libref_sgm.so", ["-DREF_MATH_PM"], "ref_sgm_harness.cpp"),
                                 ("libref_scene.so", [], "ref_scene_harness.cpp"),
                                 ("libref_fuse.so", [], "ref_fuse_harness.cpp"),
                                 ("libref_text.so", [], "ref_text_harness.cpp"),
                                 ("libref_driver.so", ["-DREF_MATH_PM"] + link_orc, "ref_driver_harness.cpp"),
                                 ("libref_driver_libm.so", ["-O3", "-march=x86-64-v3"] + link_orc, "ref_driver_harness.cpp")):
            out = os.path.join(OUT, name)
            cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-w", "-I", tmp, "-I", os.path.join(HERE, "shim")] + \
                  [os.path.join(HERE, src), "-o", out] + flags
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            outs.append(out)
    finally:
        if not os.environ.get("REF_KEEP_SNIPPETS"):
            shutil.rmtree(tmp, ignore_errors=True)
        else:
            print("snippets kept in", tmp)
    return outs


if __name__ == "__main__":
    for p in build(verbose=True):
        print("built", p)
