"""Pins the readers of the two text files in front of the path to the REFERENCE'S OWN TEXT (oracle/_ref/libref_text.so = SML.cpp, ConfigTable.cpp, the OPTDENSE list
of DepthMap.cpp:50-115, Scene::LoadViewNeighbors / SaveViewNeighbors and Util::CommandLineToArgvA, cut verbatim by oracle/ref/build_ref.py):
openmvs_amd/csrc/opt_dense.cpp + sml_text.h + mvs_front.cpp (C ABI of include/optdense.h, mvsfront.h) and their numpy mirrors (optdense.py, mvsi.py)."""
import os

import numpy as np
import pytest

from openmvs_amd import mvsfront, mvsi, optdense
from oracle import pyref as pr

pytestmark = pytest.mark.skipif(not pr.text_available(), reason="oracle/_ref/libref_text.so not built (needs /root/reference)")
HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "data", "scene", "scene.mvs")


def _write(path, data):
    open(path, "wb").write(data if isinstance(data, bytes) else data.encode("latin-1"))
    return str(path)


def _values(o):
    return np.array([float(getattr(o, name)) for name, _, _, _ in optdense.table()])


CONFIGS = [
    b"",                                                                                 # an empty file is a valid configuration
    b"Min Views Trust Point = 1\nNum Views = 8\nRandom Depth Ratio = 0.004\nFilter Adjust = 0\nIgnore Mask Label = 7\nOptimize = 0\n",
    b"  Min Views Trust Point =   1 \r\nOptimize=0\r\n\r\nNum Views = 4\nNo Such Option = 3\njust words\nEstimation Iters =\t5\t\n",
    b"Min Views = 5 views\nOptim Angle = 1e1\nMax Views = many\nMin Angle = .5\nMax Angle = 7,5\nSpeckle Size = -3\nOptimizer Max Iters = -3\nFilter Adjust = true\nInit Sparse = 2\n"
    b"Add Corners = 1\nNCC Threshold Keep = 0x10\nRandom Iters = 4294967295\nRandom Max Scale = 4294967296\nInterpolate Gap Size = 007\nView Min Score = +3.5\nMin Area = 1e-50\n"
    b"Depth Diff Threshold = 1e50\nPairwise Mul = nan\nOptimizer Eps = inf\n",
    b"Estimation Iters = 4\n[Child]\n{\n\tEstimation Iters = 9\n\tNum Views = 3\n}\nNum Views = 6\n",      # a child section is not the table; what follows it at root level is
    b"Estimation Iters = 4",                                                              # no newline at the end of the file
    b"Num Views = 5\n[]\n{\nNum Views = 6\n}\nEstimation Iters = 9\n",                     # a section without a name is a parse error: not a valid configuration, but what stood in front of it counts
    b"Num Views = 5\n[ ]\n{\nNum Views = 6\n}\nEstimation Iters = 9\n",
    b"Num Views = 5\n}\nEstimation Iters = 9\n",
    b"\n\n\nEstimation Geometric Iters = 0\n\n\n",
    b"Estimation Iters = 7\n" + b"# a long comment line that pushes the file past the reader's 2048-byte chunk\n" * 60 + b"Num Views = 9\nRandom Smooth Bonus = 0.5\n",
    b"Random Angle1 Range = 20.0\nRandom Angle2 Range = 12.0 \nRandom Smooth Depth=0.03\nRandom Smooth Normal= 11\nEstimation Geometric Weight =0.2\nSubResolution levels = 3\n"
    b"Descriptor Min Magnitude Threshold = 0.12\nResolution Level = 0\nMax Resolution = 3840\nMin Resolution = 320\nMin Views Fuse = 3\nMin Views Filter = 1\n"
    b"Min Views Filter Adjust = 0\nPoint Inside ROI = 2\nRemove Dmaps = 1\nView Min Score Ratio = 0.3\nNormal Diff Threshold = 20\nEstimate Colors = 1\nEstimate Normals = 2\n",
]


@pytest.mark.parametrize("k", range(len(CONFIGS)))
def test_config_file_is_read_like_the_reference_reads_it(tmp_path, k):
    """OPTDENSE::init(); oConfig.Load(file); OPTDENSE::update() of the reference against mvsf_optdense_load: all 46 variables."""
    p = _write(tmp_path / "dense.ini", CONFIGS[k])
    valid, ref = pr.ref_optdense_load(p)
    o = optdense.load(p)
    mine = _values(o)
    assert valid == o.loaded and len(ref) == len(mine) == 46 and valid == (b"[]" not in CONFIGS[k])
    bad = [(optdense.table()[i][1], ref[i], mine[i]) for i in range(46) if not (ref[i] == mine[i] or (np.isnan(ref[i]) and np.isnan(mine[i])))]
    assert not bad, bad


def test_missing_file_and_the_template_the_reference_writes(tmp_path):
    valid, ref = pr.ref_optdense_load(str(tmp_path / "none.ini"), save_path=str(tmp_path / "ref_template.ini"))      # DensifyPointCloud.cpp:239-254
    o = optdense.load(str(tmp_path / "none.ini"))
    assert not valid and not o.loaded and np.array_equal(ref, _values(o)) and np.array_equal(ref, _values(optdense.defaults()))
    o.save(str(tmp_path / "my_template.ini"))
    ref_lines = sorted(open(tmp_path / "ref_template.ini").read().splitlines())      # the reference writes its hash map's order; the set of lines is what counts
    my_lines = sorted(open(tmp_path / "my_template.ini").read().splitlines())
    assert len(ref_lines) == 46
    # the reference keeps the default as the literal of its source ("2.0", "16.0"); ours is the shortest decimal of the value: equal as numbers, line by line
    for a, b in zip(ref_lines, my_lines):
        ta, va = a.split(" = "); tb, vb = b.split(" = ")
        assert ta == tb and float(va) == float(vb), (a, b)
    # either template read back by either reader gives the defaults
    for path in ("ref_template.ini", "my_template.ini"):
        v2, r2 = pr.ref_optdense_load(str(tmp_path / path))
        assert v2 and np.array_equal(r2, ref) and np.array_equal(_values(optdense.load(str(tmp_path / path))), ref)


SML_TEXTS = [
    "a = 1\nb=2\n  c  =  3  \n",
    "0 1 2 3\n1 0 2\n",
    "# comment\nx\n\n y z \n= 5\nk =\n",
    "a = b = c\n",
    "first\r\nsecond\r\nname = v\r\n",
    "a = 1\n[Sec]\n{\n b = 2\n}\nc = 3\n",
    "tail without newline",
    "a = 1\n[S1]\n{\n b = 2\n [S2]\n {\n  c = 3\n }\n d = 4\n}\ne = 5\n[S3]\n{\n}\nf = 6\n",
    "a = 1\n}\nb = 2\n",                                  # a stray '}' ends the root section
    "a = 1\n[S]\n{ b = 2\n",                              # a section that is never closed
    "a = 1\n[S]\nb = 2\n",                                # a section name without a body
    "a = 1\n[]\n{\nb = 2\n}\nc = 3\n",                     # a section without a name: the reader gives up there
    "a = 1\n[ ]\n{\nb = 2\n}\nc = 3\n",                    # (a blank is a name)
    "a = 1\n[S\n{\nb = 2\n}\nc = 3\n",                     # a name that is never closed
    "q = \"quoted value\"\n",
]


@pytest.mark.parametrize("k", range(len(SML_TEXTS)))
def test_sml_root_entries_are_the_reference_readers(tmp_path, k):
    """(name, value) of the root entries: sml_text.h (through the two loaders) and mvsi._sml_root_values against SML::Load.  Unnamed entries are filed by the reference as
    Item<number of entries so far>."""
    p = _write(tmp_path / "t.sml", SML_TEXTS[k])
    ok, ref = pr.ref_sml_root(p)
    mine = {}
    for name, value in mvsi._sml_root_values(SML_TEXTS[k]):
        mine[name or "Item%d" % len(mine)] = value
    assert ref == mine and ok == mvsi._sml_section(SML_TEXTS[k], 0, None)[0]


@pytest.mark.parametrize("line", ["0 1 2 3", "  7\t8  9 ", "\"0\" \"3\" 1", "a\"b c\"d e", "", "   ", "\"\"", "x \"unterminated y", "1\r2\n3"])
def test_words_of_a_line_are_command_line_to_argv(line):
    assert mvsi._split_words(line) == pr.ref_split_words(line)


NEIGHBOUR_FILES = [
    "0 1 2 3\n1 0 2 3\n2 1 3 0\n3 2 1 0\n",
    "# cam-id neighbours\r\n0 1 2 3\r\n1   0\t2\r\n\r\n2\r\n3 2\r\n",
    "0 3\n\n\n# 1 2 3\n2 \"1\" 0\n",
    "3 0 1 2",
    "#0 1\n # 1 2\n2 3 3 3\n",
    "1 = 2 3\n",                      # an '=' makes SML name the entry: the words of its VALUE are image 2 with neighbour 3
    "0 1 2 3\n" + "# filler filler filler filler filler filler filler filler filler filler filler\n" * 40 + "1 0\n2 0 1\n",
]


@pytest.mark.parametrize("k", range(len(NEIGHBOUR_FILES)))
def test_view_neighbours_file_is_read_like_the_reference_reads_it(tmp_path, k):
    p = _write(tmp_path / "nb.txt", NEIGHBOUR_FILES[k])
    ref, first = pr.ref_load_view_neighbors(p, 4, save_path=str(tmp_path / "ref_out.txt"))
    cf, py = mvsfront.SceneFront(SCENE), mvsi.load(SCENE)
    cf.load_view_neighbors(p); mvsi.load_view_neighbors(py, p)
    assert [list(cf.neighbors(i)["ID"]) for i in range(4)] == ref == [list(im.view_scores["ID"]) for im in py.images]
    made = next(cf.neighbors(i) for i in range(4) if len(cf.neighbors(i)))[0]
    assert (np.float32(made["points"]), made["scale"], made["angle"], made["area"], made["score"]) == tuple(first)      # ViewScore{nID, 0, 1.f, FD2R(15.f), 0.5f, 3.f}
    # SaveViewNeighbors: the same bytes from all three
    cf.save_view_neighbors(str(tmp_path / "c_out.txt")); mvsi.save_view_neighbors(py, str(tmp_path / "py_out.txt"))
    want = open(tmp_path / "ref_out.txt", "rb").read()
    assert want == open(tmp_path / "c_out.txt", "rb").read() == open(tmp_path / "py_out.txt", "rb").read()


def test_working_resolution_is_the_reference_rule():
    """--resolution-level / --min-resolution / --max-resolution -> the image size fed to the estimator: views.compute_max_resolution + views.resized_size against
    TImage::computeMaxResolution + Image::ResizeImage, over sizes incl. odd ones and halves that round to even."""
    from openmvs_amd import views
    rng = np.random.default_rng(8)
    cases = [(640, 479, 1, 640, 3200), (4000, 3000, 1, 640, 3200), (4000, 3000, 0, 640, 3200), (2000, 1000, 3, 640, 3200), (3840, 2160, 0, 640, 3840), (6000, 4000, 2, 640, 2560),
             (1001, 667, 0, 100, 500), (667, 1001, 0, 100, 500), (1000, 1000, 0, 10, 333), (5, 3, 1, 640, 3200)]
    for _ in range(400):
        cases.append((int(rng.integers(1, 9000)), int(rng.integers(1, 9000)), int(rng.integers(0, 6)), int(rng.integers(1, 2000)), int(rng.integers(1, 5000))))
    for W, H, level, mn, mx in cases:
        rw, rh, rl, rres, rscale = pr.ref_image_size(W, H, level, mn, mx)
        res, lv = views.compute_max_resolution(W, H, level, mn, mx)
        assert (res, lv) == (rres, rl), (W, H, level, mn, mx)
        assert views.resized_size(W, H, res) == (rw, rh), (W, H, level, mn, mx, res)


def test_gray_conversion_is_the_reference_helpers():
    """TImage::toGray as both paths call it: PatchMatch (normalised, views.to_gray) and SGM (normalised + sRGB -> linear, sgm_pipeline.to_gray_linear), against the
    reference's CONVERT helpers compiled verbatim -- every byte value in every channel, and random pixels."""
    from openmvs_amd import sgm_pipeline, views
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (64, 256, 3)).astype(np.uint8)
    for v in range(256):
        bgr[0, v] = (v, v, v); bgr[1, v] = (v, 0, 0); bgr[2, v] = (0, v, 0); bgr[3, v] = (0, 0, v)
    assert np.array_equal(views.to_gray(bgr[..., ::-1]), pr.ref_to_gray_bgr(bgr))
    assert np.array_equal(sgm_pipeline.to_gray_linear(bgr), pr.ref_to_gray_bgr(bgr, srgb=True))
