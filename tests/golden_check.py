"""Comparison of full-size results with the digests of tests/golden/make_fullsize_golden.py (sequential CPU oracle at BASELINE's sizes)."""
import hashlib
import json
import os
import zlib

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bad_rows(a, crcs):
    a = np.ascontiguousarray(a)
    return [y for y in range(a.shape[0]) if zlib.crc32(a[y].tobytes()) != crcs[y]]


def check_maps(got, gold, what):
    """got = (depth, normal, conf); gold = one entry of the golden file.  Bit-exact (SHA-256); names the differing rows when the golden has row CRCs."""
    for a, key in zip(got, ("depth", "normal", "conf")):
        if sha(a) != gold[key]:
            rows = bad_rows(a, gold[key + "_rows"]) if key + "_rows" in gold else None
            extra = ""
            if rows is not None:
                extra = "; %d of %d rows differ, first %s" % (len(rows), a.shape[0], rows[:8])
                if key == "depth" and rows:
                    st = gold["depth_sample_step"]
                    smp = np.asarray(gold["depth_sample"], np.float32).reshape(a[::st, ::st].shape)
                    d = np.abs(a[::st, ::st] - smp)
                    extra += "; strided sample: max |diff| %.3g, %d of %d differ" % (float(d.max()), int((d > 0).sum()), d.size)
            raise AssertionError("%s: %s map differs from the sequential oracle's golden digest (valid here %d, golden %d)%s"
                                 % (what, key, int((got[0] > 0).sum()), gold["valid"], extra))
