"""The reference's dense option table and the two text files in front of the path (`--dense-config-file`, `--view-neighbors-file`): include/optdense.h,
mvsf_load_view_neighbors / mvsf_save_view_neighbors (libmvsfront.so) and their numpy mirrors (openmvs_amd/optdense.py, mvsi.py, views.py)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from openmvs_amd import mvsfront, mvsi, optdense, views

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "data", "scene", "scene.mvs")
REF_DEPTHMAP = "/root/reference/libs/MVS/DepthMap.cpp"


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(HERE, "..", "include", "optdense.h")).read()
    names = sorted(set(re.findall(r"\b(mvsf_optdense_\w+)\s*\(", hdr)))
    assert names == sorted(optdense.EXPORTS)
    optdense.defaults()


@pytest.mark.pinning
@pytest.mark.skipif(not os.path.exists(REF_DEPTHMAP), reason="needs /root/reference")
def test_table_is_the_reference_option_list():
    """Variable, title, type and first default of every DEFVAR_OPTDENSE_* / MDEFVAR_OPTDENSE_* line of libs/MVS/DepthMap.cpp, in order."""
    ref = []
    for line in open(REF_DEPTHMAP, encoding="utf-8", errors="replace"):
        m = re.match(r'M?DEFVAR_OPTDENSE_(\w+)\((\w+), "([^"]*)", "[^"]*", "([^"]*)"', line)
        if m:
            ref.append((m.group(2), m.group(3), m.group(1), m.group(4)))
    assert len(ref) == 46
    assert optdense.table() == ref


def test_defaults_agree_with_every_other_statement_of_them():
    """The defaults are restated where they are consumed (PMHipParams, MVSFOptions, views.DenseOptions, DenseDepthMapsHIP::Options): one table must give them all."""
    o = optdense.defaults()
    for name, title, kind, text in optdense.table():
        v = getattr(o, name)
        assert (v == float(np.float32(float(text)))) if kind == "float" else (v == int(text)), (name, v, text)
        assert (np.float32(float(o.get(title))) == np.float32(float(text))) if kind == "float" else (o.get(title) == text), (title, o.get(title))
    f, f0 = o.front_options(), mvsfront.default_options()
    assert all(getattr(f, n) == getattr(f0, n) for n, _ in mvsfront.MVSFOptions._fields_)
    d, d0 = o.dense_options(), views.DenseOptions()
    assert all((np.float32(a) == np.float32(b)) if isinstance(b, float) else (a == b and type(a) is type(b)) for a, b in ((getattr(d, k), getattr(d0, k)) for k in vars(d0)))
    from openmvs_amd.patchmatch import default_params
    p, p0 = o.params(seed=5), default_params(seed=5)                         # pmhip_default_params of libpmhip.so
    assert all(getattr(p, n) == getattr(p0, n) for n, _ in p._fields_)
    hpp = open(os.path.join(HERE, "..", "include", "DenseDepthMapsHIP.hpp")).read()
    for name in ("nSpeckleSize", "nIpolGapSize", "fDepthDiffThreshold", "fNormalDiffThreshold", "nMinViewsFilter", "nMinViewsFilterAdjust", "nMinViewsFuse"):
        m = re.search(r"\b%s = ([0-9.]+)f?\b" % name, hpp)
        assert m and float(m.group(1)) == float(np.float32(getattr(o, name))) or abs(float(m.group(1)) - getattr(o, name)) < 1e-6, name


def test_config_file_round_trip_and_syntax(tmp_path):
    p = str(tmp_path / "dense.ini")
    missing = optdense.load(p)                                               # no file: defaults, and the caller writes the template (DensifyPointCloud.cpp:253-254)
    assert not missing.loaded and missing.as_dict() == optdense.defaults().as_dict()
    missing.save(p)
    again = optdense.load(p)
    assert again.loaded and again.unknown == 0 and again.as_dict() == missing.as_dict()
    lines = open(p).read().splitlines()
    assert len(lines) == 46 and "Min Views Trust Point = 2" in lines and "Random Depth Ratio = 0.003" in lines and "Ignore Mask Label = -1" in lines
    # hand-written file: blanks, CRLF, a title twice (the later wins), an unknown title, an unnamed entry, a bracketed child section that is not read
    open(p, "wb").write(b"  Min Views Trust Point =   1 \r\nOptimize=0\r\n\r\nNum Views = 4\nNum Views = 8\nNo Such Option = 3\njust words\n"
                        b"Random Depth Ratio = 0.004\nFilter Adjust = 0\nIgnore Mask Label = 7\n[Child]\n{\n\tEstimation Iters = 9\n}\n")
    o = optdense.load(p)
    assert o.loaded and o.unknown == 2
    assert (o.nMinViewsTrustPoint, o.nOptimize, o.nNumViews, o.bFilterAdjust, o.nIgnoreMaskLabel, o.nEstimationIters) == (1, 0, 8, 0, 7, 3)
    assert o.fRandomDepthRatio == float(np.float32(0.004))
    # values are read with `istream >> value` (OPTDENSE::update): leading number, rest ignored; a word leaves 0
    o.set("Min Views", "5 views"); o.set("Optim Angle", "1e1"); o.set("Max Views", "many")
    assert (o.nMinViews, o.fOptimAngle, o.nMaxViews) == (5, 10.0, 0)
    with pytest.raises(KeyError):
        o.set("min views", "3")                                              # titles are case-sensitive keys of the table
    # floats are written in the shortest form that reads back exactly
    o.fViewMinScoreRatio = 1.0 / 3.0
    o.save(p)
    assert optdense.load(p).fViewMinScoreRatio == o.fViewMinScoreRatio
    # the estimator / front subsets follow the table
    o.set("Estimation Geometric Iters", 1); o.set("NCC Threshold Keep", 0.5)
    pr = o.params(seed=9)
    assert (pr.nEstimationGeometricIters, pr.fNCCThresholdKeep, pr.seed, pr.fRandomDepthRatio) == (1, 0.5, 9, float(np.float32(0.004)))
    assert o.front_options().nNumViews == 8 and o.dense_options().nNumViews == 8 and o.dense_options().nMinViewsTrustPoint == 1


def test_driver_options_from_the_table(tmp_path):
    """include/OptDenseHIP.hpp: the table -> DenseDepthMapsHIP::Options (compiled here, no GPU call)."""
    import subprocess
    from openmvs_amd import build
    lib = build.build_host_lib("libmvsfront.so")
    src = tmp_path / "t.cpp"
    src.write_text('#include "OptDenseHIP.hpp"\n#include <cstdio>\nint main(int, char** v) { MVSFOptDense o; int unk = 0; if (mvsf_optdense_load(v[1], &o, &unk)) return 2;\n'
                   ' const MVS::DenseDepthMapsHIP::Options d = MVS::DenseOptionsFrom(o, 11);\n'
                   ' printf("%u %u %u %g %u %u %u %d %d %d %u %u %g\\n", d.nOptimize, d.nSpeckleSize, d.nIpolGapSize, d.fDepthDiffThreshold, d.nMinViewsFilter, d.nMinViewsFilterAdjust,\n'
                   '  d.nMinViewsFuse, (int)d.bFilterAdjust, (int)d.bEstimateColor, (int)d.bEstimateNormal, d.seed, d.nEstimationIters, d.fRandomSmoothBonus); return 0; }\n')
    ini = tmp_path / "d.ini"
    ini.write_text("Optimize = 5\nSpeckle Size = 50\nEstimate Normals = 2\nEstimate Colors = 1\nMin Views Fuse = 3\nEstimation Iters = 4\nFilter Adjust = 0\n")
    exe = str(tmp_path / "t")
    inc = os.path.join(HERE, "..", "include")
    pm = build.build_lib("libpmhip.so")                                      # (Options() takes its estimator defaults from pmhip_default_params; no device is touched)
    subprocess.check_call(["g++", "-std=c++17", "-I", inc, str(src), lib, pm, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    out = subprocess.check_output([exe, str(ini)]).decode().split()
    assert out == ["5", "50", "7", "0.01", "2", "1", "3", "0", "0", "1", "11", "4", "0.93"]


# ---- neighbour lists the scene carries ------------------------------------------------------------------------------------------------------------

def _write(path, text):
    open(path, "wb").write(text if isinstance(text, bytes) else text.encode())
    return str(path)


def test_view_neighbors_file_both_front_ends(tmp_path):
    cf, py = mvsfront.SceneFront(SCENE), mvsi.load(SCENE)
    assert all(len(cf.neighbors(i)) == 0 for i in range(4)) and all(len(im.view_scores) == 0 for im in py.images)      # archive version 6: no stored scores
    p = _write(tmp_path / "nb.txt", "# cam-id neighbours\r\n0 1 2 3\n1   0\t2\n\n2\n3 = 2 1\n\"0\" \"3\" 1\n")
    cf.load_view_neighbors(p); mvsi.load_view_neighbors(py, p)
    want = {0: [3, 1], 1: [0, 2], 2: [], 3: []}     # "2" alone: fewer than two words, skipped; "3 = 2 1": SML names the entry "3", its value "2 1" is image 2 with neighbour 1 ...
    want[2] = [1]                                   # ... (the reference's container does that; a plain list never has an '='); the quoted line re-lists image 0 (later line wins)
    for i in range(4):
        a, b = cf.neighbors(i), py.images[i].view_scores
        assert a.tobytes() == b.tobytes() and list(a["ID"]) == want[i], (i, a, b)
        if len(a):                                  # ViewScore{ID, 0, 1.f, FD2R(15.f), 0.5f, 3.f}, Scene.cpp:451
            assert set(a["points"]) == {0} and set(a["scale"]) == {1.0} and set(a["area"]) == {0.5} and set(a["score"]) == {3.0}
            assert a["angle"][0] == np.float32(15.0) * (np.float32(3.14159265358979323846) / np.float32(180.0))
    # SaveViewNeighbors: every image, best first; and it reads back to the same lists
    q, q2 = str(tmp_path / "out_c.txt"), str(tmp_path / "out_py.txt")
    cf.save_view_neighbors(q); mvsi.save_view_neighbors(py, q2)
    assert open(q, "rb").read() == open(q2, "rb").read() == b"0 3 1\n1 0 2\n2 1\n3\n"
    cf2, py2 = mvsfront.SceneFront(SCENE), mvsi.load(SCENE)
    cf2.load_view_neighbors(q); mvsi.load_view_neighbors(py2, q)
    assert [list(cf2.neighbors(i)["ID"]) for i in range(4)] == [[3, 1], [0, 2], [1], []] == [list(im.view_scores["ID"]) for im in py2.images]
    # an id outside the scene (the reference asserts) is refused and nothing is installed
    bad = _write(tmp_path / "bad.txt", "0 1\n1 9\n")
    cf3, py3 = mvsfront.SceneFront(SCENE), mvsi.load(SCENE)
    with pytest.raises(ValueError):
        cf3.load_view_neighbors(bad)
    with pytest.raises(ValueError):
        mvsi.load_view_neighbors(py3, bad)
    assert len(cf3.neighbors(0)) == 0 and len(py3.images[0].view_scores) == 0
    for junk in ("0 -1\n", "0 1x\n", "0 99999999999\n", "x 1\n"):
        with pytest.raises(ValueError):
            cf3.load_view_neighbors(_write(tmp_path / "j.txt", junk))
        with pytest.raises(ValueError):
            mvsi.load_view_neighbors(py3, str(tmp_path / "j.txt"))
    with pytest.raises(ValueError):
        cf3.load_view_neighbors(str(tmp_path / "does_not_exist.txt"))


def test_select_views_takes_the_list_the_scene_carries(tmp_path):
    """DepthMapsData::SelectViews (SceneDensify.cpp:278-281): a stored list replaces the scoring; FilterNeighborViews and the score cut of InitViews still apply; no seed points."""
    cf, py = mvsfront.SceneFront(SCENE), mvsi.load(SCENE)
    cams = views.Cameras(py)
    scored = views.select_views(py, cams, 0)
    assert scored is not None and len(scored[1]) > 0
    p = _write(tmp_path / "nb.txt", "0 2 1\n")
    cf.load_view_neighbors(p); mvsi.load_view_neighbors(py, p)
    a, b = cf.select_views(0), views.select_views(py, cams, 0)
    assert list(a[0]["ID"]) == [2, 1] == list(b[0]["ID"]) and a[0].tobytes() == b[0].tobytes()
    assert len(a[1]) == 0 and len(b[1]) == 0
    # image 1 has no list: scored as before, identically by both front ends
    a1, b1 = cf.select_views(1), views.select_views(py, cams, 1)
    assert list(a1[0]["ID"]) == list(b1[0]["ID"]) and len(a1[1]) == len(b1[1]) > 0
    # the score cut applies to a stored list too: score 3 >= max(3 * 0.03, fViewMinScore = 2), but not with fViewMinScore = 4; nNumViews cuts the length
    assert cf.select_views(0, mvsfront.default_options(fViewMinScore=4.0)) is None
    assert views.select_views(py, cams, 0, views.DenseOptions(fViewMinScore=4.0)) is None
    assert list(cf.select_views(0, mvsfront.default_options(nNumViews=1))[0]["ID"]) == [2] == list(views.select_views(py, cams, 0, views.DenseOptions(nNumViews=1))[0]["ID"])
    # no seed points: the depth map starts from random values in [0.1, 100] (InitViews, :418-427)
    d, n, dmin, dmax = cf.init_depth_map(0, a[1], (640, 480))
    d2, n2, dmin2, dmax2 = views.init_depth_map(py, views.Cameras(py, [(640, 480)] * 4), 0, b[1])
    assert not d.any() and not d2.any() and np.float32(dmin) == np.float32(dmin2) == np.float32(0.1) and dmax == dmax2 == 100.0


def test_archive_view_scores_are_the_neighbour_lists(tmp_path):
    """An archive of version 7 stores every image's view scores and depth statistics; Scene::LoadInterface makes them Image::neighbors / avgDepth (Scene.cpp:158-159)."""
    py = mvsi.load(SCENE)
    cams = views.Cameras(py)
    for i, im in enumerate(py.images):                     # what the reference's DensifyPointCloud saves in scene_dense.mvs: the lists it selected
        ok, nb, pts, avg = views.select_neighbor_views(py, cams, i)
        assert ok
        im.view_scores, im.avg_depth, im.min_depth, im.max_depth = nb, avg, 0.5 * avg, 2.0 * avg
    p = str(tmp_path / "dense.mvs")
    mvsi.save(p, py, version=7)
    cf, py7 = mvsfront.SceneFront(p), mvsi.load(p)
    for i in range(4):
        assert cf.neighbors(i).tobytes() == py.images[i].view_scores.tobytes() == py7.images[i].view_scores.tobytes()
        assert cf.image_depths(i) == (py7.images[i].min_depth, py7.images[i].avg_depth, py7.images[i].max_depth)
        a, b = cf.select_views(i), views.select_views(py7, views.Cameras(py7), i)
        assert a[0].tobytes() == b[0].tobytes() and len(a[1]) == 0 and len(b[1]) == 0 and a[2] == b[2] == py7.images[i].avg_depth
    # setting a list by hand, and a neighbour outside the scene
    nb = cf.neighbors(0)[:1].copy()
    cf.set_neighbors(1, nb)
    assert cf.neighbors(1).tobytes() == nb.tobytes()
    nb["ID"] = 77
    with pytest.raises(ValueError):
        cf.set_neighbors(1, nb)
    # a stored view score that names an image the archive does not have is a corrupt file
    raw = bytearray(open(p, "rb").read())
    first = py.images[0].view_scores[:1].tobytes()
    at = bytes(raw).find(first)
    assert at > 0
    raw[at:at + 4] = (1000).to_bytes(4, "little")
    bad = str(tmp_path / "bad.mvs"); open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        mvsfront.SceneFront(bad)


# ---- ignore masks ---------------------------------------------------------------------------------------------------------------------------------

def test_ignore_mask_lookup_and_import():
    """DepthEstimator::ImportIgnoreMask (DepthMap.cpp:296-323) and --mask-path (DensifyPointCloud.cpp:307-320): where the mask is looked for, and label -> process / ignore at the
    depth map's size through cv::resize(INTER_NEAREST)."""
    assert views.mask_file_name("images/00001.jpg") == "images/00001.mask.png"
    assert views.mask_file_name("images/a.b/00001") == "images/a.mask.png"               # Util::getFileFullName cuts at the LAST '.', wherever it is
    assert views.mask_file_name("noext") == "noext.mask.png"
    assert views.mask_file_name("images/00001.jpg", "masks/m1.png") == "masks/m1.png"
    assert views.mask_file_name("images/00001.jpg", "", "/data/masks") == "/data/masks/00001.mask.png" == views.mask_file_name("images/00001.jpg", "", "/data/masks/")
    with pytest.raises(ValueError):
        views.mask_file_name("images/00001.jpg", "masks/m1.png", "/data/masks")
    rng = np.random.default_rng(3)
    for (sh, sw), (w, h) in (((48, 64), (64, 48)), ((48, 64), (32, 24)), ((37, 53), (64, 48)), ((100, 70), (33, 47)), ((5, 7), (3, 2))):
        labels = rng.integers(0, 4, (sh, sw)).astype(np.uint16)
        labels[0, 0] = 65535
        got = views.import_ignore_mask(labels, (w, h), 2)
        want = np.zeros((h, w), np.uint8)
        ifx, ify = 1.0 / (w / sw), 1.0 / (h / sh)                                           # resizeNN, literally
        for y in range(h):
            for x in range(w):
                want[y, x] = labels[min(int(np.floor(y * ify)), sh - 1), min(int(np.floor(x * ifx)), sw - 1)] != 2
        assert got.dtype == np.uint8 and np.array_equal(got, want)
        assert np.array_equal(views.import_ignore_mask(labels, (w, h), 65535) == 0, views.import_ignore_mask(labels, (w, h), -1) == 0)   # (uint16_t)nIgnoreMaskLabel
    assert views.import_ignore_mask(np.full((4, 4), 7, np.uint8), (4, 4), 7).sum() == 0 and views.import_ignore_mask(np.full((4, 4), 7, np.uint8), (4, 4), 1).all()


def test_load_scene_reads_the_option_table_the_neighbour_file_and_the_masks(tmp_path):
    """densify.load_scene with what the command line hands the reference: --dense-config-file, --view-neighbors-file, --ignore-mask-label."""
    from openmvs_amd import densify
    ini = tmp_path / "dense.ini"; ini.write_text("Ignore Mask Label = 3\nMin Resolution = 320\nResolution Level = 1\n")
    nbf = tmp_path / "nb.txt"; nbf.write_text("0 1 2\n")
    opt = optdense.load(str(ini))
    py = mvsi.load(SCENE)
    seen = []

    def masks(p):
        seen.append(os.path.basename(p))
        if p.endswith("00000.mask.png") or "0000" not in p:
            raise OSError("no such mask")
        m = np.zeros((479, 640), np.uint16); m[:100] = 3
        return m

    sv = densify.load_scene(SCENE, opt=opt, view_neighbors_file=str(nbf), mask_loader=masks)
    assert (sv.width, sv.height) == (320, 240) and sv.mask_option and len(seen) == 4
    got = sorted(sv.masks)
    assert len(got) >= 1 and all(sv.masks[i].shape == (240, 320) for i in got)
    m = sv.masks[got[0]]
    assert not m[:50].any() and m[51:].all()                                   # rows 0..99 of 479 carry the label: rows 0..50 of 240 (nearest rule)
    assert list(sv.neighbors[0]) == [1, 2] and not sv.init_depth[0].any() and abs(sv.dmin[0] - 0.1) < 1e-6 and sv.dmax[0] == 100.0     # the listed neighbours; no seed points
    assert len(sv.neighbors[1]) >= 1 and sv.init_depth[1].any()               # the others are scored and seeded as before
    sv2 = densify.load_scene(SCENE)                                            # option off: nothing is looked up
    assert not sv2.mask_option and not sv2.masks
