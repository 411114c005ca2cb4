"""profiles/traffic.json from the counter passes of tools/pmc/run_pmc.sh:
    python tools/pmc/make_traffic.py <dir with pmc_*_per_kernel.txt and *_run.json> [kernel name substring]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch (L2 <-> fabric requests; Infinity-Cache hits are counted, MI355X_MICROARCH.md
"HBM").  The guide's gfx950 note (FETCH_SIZE = 1/2 of the bytes of a wide streaming read) is calibrated for 16-B-per-lane loads only; this kernel reads
4 B per lane, so both the raw figure and the doubled one are recorded and `bytes_per_launch` uses the raw one.  The file carries a digest of the kernel
sources it was measured on; bench.py reports the figure as offline and drops it when the tree's kernels differ."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "pm_sweep"


def kernel_digest():
    h = hashlib.sha256()
    for f in ("pm_kernels.hip", "pm_band.hip", "pm_math.h"):   # the measured kernel (pm_sweep2_kernel, pm_band.hip) and every device function it uses; not the host engine
        h.update(open(os.path.join(ROOT, "openmvs_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def table(name):
    out = {}; cur = None
    for line in open(os.path.join(d, "pmc_%s_per_kernel.txt" % name)):
        m = re.match(r"(\S.*) dispatches (\d+)", line)
        if m:
            cur = m.group(1); out[cur] = {"dispatches": int(m.group(2))}; continue
        m = re.match(r"\s+(\S+)\s+sum (\S+)\s+per-dispatch (\S+)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


def per_launch(tab, counter):
    n = sum(v["dispatches"] for k, v in tab.items() if kern in k); s = sum(v.get(counter, 0.0) for k, v in tab.items() if kern in k)
    return s / max(1, n), n


fetch, n = per_launch(table("fetch"), "FETCH_SIZE"); write, _ = per_launch(table("write"), "WRITE_SIZE")
run = json.load(open(os.path.join(d, "unprofiled_run.json")))
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, stand-alone C++ workload tools/pmc/pmc_workload.cpp (%d views %dx%d, %d geometric round(s), one stream)"
                 % (run["views"], run["w"], run["h"], run["geo_iters"]),
       "kernel": kern, "dispatches": n, "kernel_digest": kernel_digest(),
       "fetch_bytes_per_launch": round(fetch * 1024), "write_bytes_per_launch": round(write * 1024),
       "bytes_per_launch": round((fetch + write) * 1024), "bytes_per_launch_fetch_doubled": round((2 * fetch + write) * 1024),
       "algorithmic_bytes_per_launch": run["algorithmic_bytes_per_launch"], "avg_launch_us_unprofiled": run["avg_launch_us"]}
try:
    sq1 = table("sq1"); sq2 = table("sq2")
    def tot(tab, c): return sum(v.get(c, 0.0) for k, v in tab.items() if kern in k)
    waves = tot(sq1, "SQ_WAVES")
    res["sq"] = {"waves_per_launch": round(waves / n, 1), "valu_insts_per_wave": round(tot(sq1, "SQ_INSTS_VALU") / waves), "salu_insts_per_wave": round(tot(sq1, "SQ_INSTS_SALU") / waves),
                 "lds_insts_per_wave": round(tot(sq1, "SQ_INSTS_LDS") / waves), "smem_insts_per_wave": round(tot(sq2, "SQ_INSTS_SMEM") / waves), "vmem_rd_insts_per_wave": round(tot(sq2, "SQ_INSTS_VMEM_RD") / waves),
                 "wave_quadcycles_per_wave": round(tot(sq1, "SQ_WAVE_CYCLES") / waves), "frac_active_valu": round(tot(sq1, "SQ_ACTIVE_INST_VALU") / tot(sq1, "SQ_WAVE_CYCLES"), 4),
                 "frac_active_any": round(tot(sq2, "SQ_ACTIVE_INST_ANY") / tot(sq1, "SQ_WAVE_CYCLES"), 4), "frac_wait_any": round(tot(sq2, "SQ_WAIT_ANY") / tot(sq1, "SQ_WAVE_CYCLES"), 4),
                 "frac_wait_inst_any": round(tot(sq1, "SQ_WAIT_INST_ANY") / tot(sq1, "SQ_WAVE_CYCLES"), 4), "lds_bank_conflict_per_lds_active": round(tot(sq2, "SQ_LDS_BANK_CONFLICT") / max(1.0, tot(sq2, "SQ_ACTIVE_INST_LDS")), 3),
                 "valu_lane_utilisation": round(tot(sq2, "SQ_THREAD_CYCLES_VALU") / max(1.0, 64.0 * tot(sq1, "SQ_ACTIVE_INST_VALU")), 4)}
except Exception as ex:
    res["sq"] = {"error": str(ex)}
json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
