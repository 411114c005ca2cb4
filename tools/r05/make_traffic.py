"""profiles/traffic.json from the counter passes of tools/r05/pmc_bench.sh -- taken on the BENCHMARK'S OWN workload (100 views of 1920x1080, two view groups, photometric
pass + 2 geometric rounds; the stand-alone C++ program makes one bench.py step's engine calls, in one view group: see pmc_bench.sh):
    python tools/r05/make_traffic.py <dir with pmc_*_per_kernel.txt and unprofiled_run.json>
Per step the same kernels run on the same pixels as in the benchmark; the benchmark splits them over twice as many launches (two groups), so per-launch figures are the per-step
sums divided by the benchmark's 43 126 launches (BENCH_LAUNCHES), and rates are formed by bench.py with its own wall time.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch (L2 <-> fabric requests; Infinity-Cache hits are counted, MI355X_MICROARCH.md "HBM").  The tap rows load
16 bytes per lane, for which the guide's gfx950 note prescribes FETCH_SIZE x 2; both the raw and the doubled figure are recorded.  Per kernel family (pm_sweep2, pm_sweep_widen,
pm_sweep_wide, pm_init) and for all sweep launches together; the SQ sums give the share of the SIMDs' cycles in which a VALU instruction executes during the sweeps of the
UNPROFILED run (counters: what was executed; unprofiled run: how long it took).  The file carries a digest of the kernel sources it was measured on; bench.py drops it when
the tree's kernels differ."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = sys.argv[1]
FAMILIES = ("pm_sweep2_kernel", "pm_sweep_widen_kernel", "pm_sweep_wide_kernel", "pm_init_kernel")
SIMDS, CLOCK_HZ = 1024, 2.4e9
BENCH_LAUNCHES = 43126      # sweep launches of one benchmark step (two groups; bench.py's roofline.launches / steps)


def kernel_digest():
    h = hashlib.sha256()
    for f in ("pm_kernels.hip", "pm_band.hip", "pm_wide_n.hip", "pm_math.h"):
        h.update(open(os.path.join(ROOT, "openmvs_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def table(name):
    out = {}; cur = None
    path = os.path.join(d, "pmc_%s_per_kernel.txt" % name)
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"(\S.*) dispatches (\d+)", line)
        if m:
            cur = m.group(1); out[cur] = {"dispatches": int(m.group(2))}; continue
        m = re.match(r"\s+(\S+)\s+sum (\S+)\s+per-dispatch (\S+)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


def fam(tab, family, counter=None):
    rows = [v for k, v in tab.items() if family in k]
    if counter is None:
        return sum(v["dispatches"] for v in rows)
    return sum(v.get(counter, 0.0) for v in rows)


fetch, write, sq1, sq2 = table("fetch"), table("write"), table("sq1"), table("sq2")
run = json.load(open(os.path.join(d, "unprofiled_run.json")))
res = {"source": "rocprofv3 --pmc, one counter group per pass, on the benchmark's own workload through tools/pmc/pmc_workload.cpp (%d views %dx%d, photometric pass + %d geometric rounds, "
                 "one view group with the per-launch kernel threshold doubled: per step the benchmark's kernels on the benchmark's pixels; tools/r05/pmc_bench.sh)" % (run["views"], run["w"], run["h"], run["geo_iters"]),
       "kernel_digest": kernel_digest(),
       "unprofiled": {"seconds_per_step": run["seconds"], "mpix_s": run["mpix_s"], "sweep_launches": run["sweep_launches"], "avg_launch_us": run["avg_launch_us"],
                      "algorithmic_bytes_per_launch": run["algorithmic_bytes_per_launch"], "sweep_wall_s": run.get("sweep_wall_s")},
       "families": {}}
tot = {"dispatches": 0, "fetch": 0.0, "write": 0.0, "valu_active": 0.0, "wave_cycles": 0.0, "waves": 0.0, "valu_insts": 0.0}
for f in FAMILIES:
    n = fam(fetch, f) or fam(sq1, f)
    if not n:
        continue
    fb, wb = fam(fetch, f, "FETCH_SIZE") * 1024, fam(write, f, "WRITE_SIZE") * 1024
    waves = fam(sq1, f, "SQ_WAVES")
    row = {"dispatches": n, "fetch_bytes_per_launch_raw": round(fb / n), "write_bytes_per_launch": round(wb / max(1, fam(write, f))),
           "fabric_bytes_per_launch": round((2 * fb + wb) / n)}
    if waves:
        wc = fam(sq1, f, "SQ_WAVE_CYCLES")
        row.update({"waves_per_launch": round(waves / fam(sq1, f), 1), "valu_insts_per_wave": round(fam(sq1, f, "SQ_INSTS_VALU") / waves), "salu_insts_per_wave": round(fam(sq1, f, "SQ_INSTS_SALU") / waves),
                    "lds_insts_per_wave": round(fam(sq1, f, "SQ_INSTS_LDS") / waves), "wave_cycles_per_wave": round(4 * wc / waves),
                    "frac_active_valu": round(fam(sq1, f, "SQ_ACTIVE_INST_VALU") / wc, 4), "frac_wait_inst_any": round(fam(sq1, f, "SQ_WAIT_INST_ANY") / wc, 4)})
        if fam(sq2, f):
            row.update({"vmem_rd_insts_per_wave": round(fam(sq2, f, "SQ_INSTS_VMEM_RD") / max(1.0, waves)), "frac_wait_any": round(fam(sq2, f, "SQ_WAIT_ANY") / wc, 4),
                        "frac_active_any": round(fam(sq2, f, "SQ_ACTIVE_INST_ANY") / wc, 4),
                        "valu_lane_utilisation": round(fam(sq2, f, "SQ_THREAD_CYCLES_VALU") / max(1.0, 64.0 * fam(sq1, f, "SQ_ACTIVE_INST_VALU")), 4)})
    res["families"][f] = row
    if "sweep" in f:
        tot["dispatches"] += n; tot["fetch"] += fb; tot["write"] += wb
        tot["valu_active"] += 4 * fam(sq1, f, "SQ_ACTIVE_INST_VALU"); tot["wave_cycles"] += 4 * fam(sq1, f, "SQ_WAVE_CYCLES"); tot["waves"] += waves; tot["valu_insts"] += fam(sq1, f, "SQ_INSTS_VALU")
n = BENCH_LAUNCHES
fabric = 2 * tot["fetch"] + tot["write"]
alg = run["algorithmic_bytes_per_launch"] * run["sweep_launches"] / BENCH_LAUNCHES      # algorithmic bytes of a step / the benchmark's launches
wall = run.get("sweep_wall_s") or run["seconds"]
res["sweeps"] = {"dispatches_profiled": tot["dispatches"], "dispatches": BENCH_LAUNCHES, "fetch_bytes_per_launch_raw": round(tot["fetch"] / n), "write_bytes_per_launch": round(tot["write"] / n),
                 "fabric_bytes_per_launch": round(fabric / n), "algorithmic_bytes_per_launch": alg, "over_algorithmic": round(fabric / n / alg, 2),
                 "fabric_bytes_per_step": round(fabric),
                 "correction": "FETCH_SIZE x 2 (16-byte-per-lane loads on gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported"}
if tot["valu_active"]:
    res["valu"] = {"valu_active_cycles_per_step": round(tot["valu_active"]), "wave_visits_per_step": round(tot["waves"]), "valu_insts_per_wave_visit": round(tot["valu_insts"] / max(1.0, tot["waves"])),
                   "note": "SQ_ACTIVE_INST_VALU (x 4 cycles) summed over every sweep launch of one benchmark step; bench.py divides by 1024 SIMDs x 2.4 GHz x its own wall time of the passes"}
if "valu" not in res and len(sys.argv) > 2:      # no SQ pass of this run finished: carry the previous round's SQ block along, labelled (bench.py extrapolates from it and says so)
    old = json.load(open(sys.argv[2]))
    if "sq" in old:
        res["sq_round4"] = dict(old["sq"], source=old.get("source"), kernel_digest=old.get("kernel_digest"))
notes = []
if not write:
    notes.append("the WRITE_SIZE pass hung twice (rocprofv3 --pmc at start-up, 90 s each): fabric bytes are reads only; round 4 measured writes at 1.7 % of the reads")
if "valu" not in res:
    notes.append("both SQ passes hung: no SQ sums of this workload")
if notes:
    res["notes"] = notes
json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
