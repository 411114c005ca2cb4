"""profiles/traffic.json from the counter passes over a SLICE of the benchmark (tools/pmc/pmc_slice.cpp, tools/r06/final.sh): the benchmark's 100 views of 1920x1080 resident, the maps
of the benchmark after its photometric pass, then ONE level-0 sweep of the photometric kernels ("photo") and ONE of the geometric kernels ("geo"), one view group -- about 3 000
dispatches per pass, which rocprofv3 --pmc finishes in seconds (the whole schedule, 43 126 dispatches, hung in 6 of 7 passes in round 5).
    python tools/r06/make_traffic.py <dir with pmc_<group>_<photo|geo>_per_kernel.txt and slice_unprofiled.jsonl>
A benchmark step is 3 photometric sweeps on each of three pyramid levels (1, 1/4, 1/16 of the pixels) and 2 geometric sweeps on level 0: per-step sums are
photo x 3 x (1 + 1/4 + 1/16) + geo x 2 -- the same kernels on the same scene, the coarse levels priced by their pixel share -- and per-launch figures divide by the benchmark's 43 126
launches (two view groups).  FETCH_SIZE / WRITE_SIZE: KiB per dispatch, L2 <-> fabric requests tallied at 64 B; the tap rows' runs of 16-byte entries travel as 128-byte requests
(profiles/r06_call1/fetch_calib_*: FETCH_SIZE = 1/2 of the bytes for the runs pattern as for a coalesced stream, and 4x the bytes for isolated 16-byte gathers), so the doubled
figure is the byte count for this access pattern; Infinity-Cache hits are counted (MI355X_MICROARCH.md "HBM").  The SQ sums are live counters of BOTH kernel families of the slice."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = sys.argv[1]
FAMILIES = ("pm_sweep2_kernel", "pm_sweep_widen_kernel", "pm_init_kernel")
BENCH_LAUNCHES = 43126      # sweep launches of one benchmark step (two view groups; bench.py's roofline.launches / steps)
PHOTO_SWEEPS = 3 * (1 + 0.25 + 0.0625)
GEO_SWEEPS = 2.0
INIT_PHOTO = 1 + 0.25 + 0.0625


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
    rows = [v for k, v in tab.items() if family in k or family[3:] in k]      # (the aggregator keeps the last 40 characters of a kernel name)
    if counter is None:
        return sum(v["dispatches"] for v in rows)
    return sum(v.get(counter, 0.0) for v in rows)


T = {m: {g: table("%s_%s" % (g, m)) for g in ("fetch", "write", "sq1", "sq2", "tcc", "tcp", "grbm", "ta1", "ta2")} for m in ("photo", "geo")}
runs = [json.loads(l) for l in open(os.path.join(d, "slice_unprofiled.jsonl")) if l.strip().startswith("{")]
res = {"source": "rocprofv3 --pmc, one counter group per pass, over a slice of the benchmark through tools/pmc/pmc_slice.cpp: 100 views 1920x1080 resident with the benchmark's maps, one "
                 "level-0 sweep of the photometric and one of the geometric kernels, one view group (tools/r06/final.sh); per-step figures = photo x 3 x (1 + 1/4 + 1/16) + geo x 2",
       "kernel_digest": kernel_digest(), "slice_unprofiled": [r for r in runs if r.get("views") == 100 and r.get("groups") in (1, 2)], "families": {}}
step = {"fetch": 0.0, "write": 0.0, "valu_active": 0.0, "valu_insts": 0.0, "waves": 0.0, "wave_cycles": 0.0, "wait_inst": 0.0, "wait_any": 0.0, "hit": 0.0, "miss": 0.0, "profiled": 0}
for f in FAMILIES:
    row = {}
    for mode, weight in (("photo", INIT_PHOTO if "init" in f else PHOTO_SWEEPS), ("geo", GEO_SWEEPS)):
        t = T[mode]
        n = fam(t["fetch"], f) or fam(t["sq1"], f)
        if not n:
            continue
        fb, wb = fam(t["fetch"], f, "FETCH_SIZE") * 1024, fam(t["write"], f, "WRITE_SIZE") * 1024
        waves, wc = fam(t["sq1"], f, "SQ_WAVES"), fam(t["sq1"], f, "SQ_WAVE_CYCLES")
        r = {"dispatches": n, "fetch_bytes_per_launch_raw": round(fb / n), "write_bytes_per_launch": round(wb / n), "fabric_bytes_per_launch": round((2 * fb + wb) / n)}
        if waves:
            r.update({"waves_per_launch": round(waves / n, 1), "valu_insts_per_wave": round(fam(t["sq1"], f, "SQ_INSTS_VALU") / waves), "wave_cycles_per_wave": round(4 * wc / waves),
                      "frac_active_valu": round(fam(t["sq1"], f, "SQ_ACTIVE_INST_VALU") / wc, 4), "frac_wait_inst_any": round(fam(t["sq1"], f, "SQ_WAIT_INST_ANY") / wc, 4),
                      "frac_wait_any": round(fam(t["sq1"], f, "SQ_WAIT_ANY") / wc, 4), "frac_active_any": round(fam(t["sq1"], f, "SQ_ACTIVE_INST_ANY") / wc, 4)})
            gui = fam(t["grbm"], f, "GRBM_GUI_ACTIVE")
            if gui:   # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of one XCD = / 8; 1 024 SIMDs run for that long
                r["valu_busy_of_simd_cycles"] = round(4 * fam(t["sq1"], f, "SQ_ACTIVE_INST_VALU") / (1024 * gui / 8), 4)
        if fam(t["sq2"], f):
            r.update({"vmem_rd_insts_per_wave": round(fam(t["sq2"], f, "SQ_INSTS_VMEM_RD") / max(1.0, waves)), "lds_insts_per_wave": round(fam(t["sq2"], f, "SQ_INSTS_LDS") / max(1.0, waves)),
                      "valu_lane_utilisation": round(fam(t["sq2"], f, "SQ_THREAD_CYCLES_VALU") / max(1.0, 64.0 * fam(t["sq1"], f, "SQ_ACTIVE_INST_VALU")), 4)})
        hit, miss = fam(t["tcc"], f, "TCC_HIT_sum"), fam(t["tcc"], f, "TCC_MISS_sum")
        if hit + miss:
            r["l2_hit_rate"] = round(hit / (hit + miss), 4)
        acc, l1miss = fam(t["tcp"], f, "TCP_TOTAL_CACHE_ACCESSES_sum"), fam(t["tcp"], f, "TCP_TCC_READ_REQ_sum")
        if acc:
            r["l1_hit_rate"] = round(1.0 - l1miss / acc, 4)
        # the texture-address unit (one per CU): busy cycles of all 256 over the cycles the kernels ran, and what it was busy with.  A wave-load whose lanes are scattered
        # costs it 64 cycles whatever the width (tools/probes/l1_gather.hip, profiles/r06_call10): the tap rows' 16-byte gathers are that case.
        ta_busy, ta_waves = fam(t["ta1"], f, "TA_TA_BUSY_sum"), fam(t["ta1"], f, "TA_TOTAL_WAVEFRONTS_sum")
        gui = fam(t["grbm"], f, "GRBM_GUI_ACTIVE")
        if ta_busy and gui:
            r.update({"ta_busy_frac": round(ta_busy / (256 * gui / 8), 4), "ta_wavefronts_per_wave": round(ta_waves / max(1.0, waves), 1),
                      "ta_buffer_wavefronts_per_wave": round(fam(t["ta1"], f, "TA_BUFFER_WAVEFRONTS_sum") / max(1.0, waves), 1),
                      "ta_flat_wavefronts_per_wave": round(fam(t["ta1"], f, "TA_FLAT_WAVEFRONTS_sum") / max(1.0, waves), 1),
                      "ta_busy_cycles_per_wavefront": round(ta_busy / max(1.0, ta_waves), 1)})
        if fam(t["ta2"], f, "TA_BUFFER_TOTAL_CYCLES_sum"):
            r.update({"ta_buffer_cycles_per_wave": round(fam(t["ta2"], f, "TA_BUFFER_TOTAL_CYCLES_sum") / max(1.0, waves)),
                      "ta_addr_stalled_by_tc_frac": round(fam(t["ta2"], f, "TA_ADDR_STALLED_BY_TC_CYCLES_sum") / max(1.0, 256 * gui / 8), 4) if gui else None,
                      "ta_data_stalled_by_tc_frac": round(fam(t["ta2"], f, "TA_DATA_STALLED_BY_TC_CYCLES_sum") / max(1.0, 256 * gui / 8), 4) if gui else None,
                      "td_busy_frac": round(fam(t["ta2"], f, "TD_TD_BUSY_sum") / max(1.0, 256 * gui / 8), 4) if gui else None})
        row[mode] = r
        if "sweep" in f:
            step["vmem_rd"] = step.get("vmem_rd", 0.0) + weight * fam(t["sq2"], f, "SQ_INSTS_VMEM_RD")
            step["ta_busy"] = step.get("ta_busy", 0.0) + weight * ta_busy; step["gui"] = step.get("gui", 0.0) + weight * gui
            step["fetch"] += weight * fb; step["write"] += weight * wb; step["profiled"] += n
            step["valu_active"] += weight * 4 * fam(t["sq1"], f, "SQ_ACTIVE_INST_VALU"); step["valu_insts"] += weight * fam(t["sq1"], f, "SQ_INSTS_VALU")
            step["waves"] += weight * waves; step["wave_cycles"] += weight * 4 * wc
            step["wait_inst"] += weight * 4 * fam(t["sq1"], f, "SQ_WAIT_INST_ANY"); step["wait_any"] += weight * 4 * fam(t["sq1"], f, "SQ_WAIT_ANY")
            step["hit"] += weight * hit; step["miss"] += weight * miss
    if row:
        # (bench.py reads dispatches / fabric_bytes_per_launch / frac_* of the family at the top level: the photometric slice stands for the family there)
        top = dict(row.get("photo") or row.get("geo")); top.update({"photo": row.get("photo"), "geo": row.get("geo")})
        res["families"][f] = top
photo1 = [r for r in runs if r.get("views") == 100 and r.get("mode") == "photo" and r.get("groups") == 1]
geo1 = [r for r in runs if r.get("views") == 100 and r.get("mode") == "geo" and r.get("groups") == 1]
alg_step = (photo1[0]["algorithmic_bytes_per_launch"] * photo1[0]["sweep_launches"] * PHOTO_SWEEPS + geo1[0]["algorithmic_bytes_per_launch"] * geo1[0]["sweep_launches"] * GEO_SWEEPS) if photo1 and geo1 else 0.0
fabric = 2 * step["fetch"] + step["write"]
res["sweeps"] = {"dispatches_profiled": step["profiled"], "dispatches": BENCH_LAUNCHES, "fetch_bytes_per_launch_raw": round(step["fetch"] / BENCH_LAUNCHES),
                 "write_bytes_per_launch": round(step["write"] / BENCH_LAUNCHES), "fabric_bytes_per_launch": 2 * round(step["fetch"] / BENCH_LAUNCHES) + round(step["write"] / BENCH_LAUNCHES),
                 "algorithmic_bytes_per_launch": round(alg_step / BENCH_LAUNCHES, 1), "over_algorithmic": round(fabric / max(1.0, alg_step), 2),
                 "fabric_bytes_per_step": round(fabric), "l2_hit_rate": round(step["hit"] / max(1.0, step["hit"] + step["miss"]), 4),
                 "correction": "FETCH_SIZE x 2: the tap rows' runs of 16-byte entries travel as 128-byte requests tallied at 64 B (calibrated on this access pattern: profiles/r06_call1/fetch_calib_*); "
                               "WRITE_SIZE as reported; per step = photo x 3 x (1 + 1/4 + 1/16) + geo x 2 of the slice"}
res["valu"] = {"valu_active_cycles_per_step": round(step["valu_active"]), "wave_visits_per_step": round(step["waves"]), "valu_insts_per_wave_visit": round(step["valu_insts"] / max(1.0, step["waves"])),
               "cycles_per_valu_inst": round(step["valu_active"] / max(1.0, step["valu_insts"]), 2),
               "wave_cycle_shares": {"issuing": round(1.0 - (step["wait_inst"] + step["wait_any"]) / max(1.0, step["wave_cycles"]), 4), "issue_stall": round(step["wait_inst"] / max(1.0, step["wave_cycles"]), 4),
                                     "parked_on_waitcnt": round(step["wait_any"] / max(1.0, step["wave_cycles"]), 4)}}
# the bound the sweeps run against: vector-memory wave-loads through the texture-address units, 64 cycles each when scattered (probe), one unit per CU
res["gather"] = {"vmem_rd_wave_loads_per_step": round(step.get("vmem_rd", 0.0)), "vmem_rd_wave_loads_per_wave_visit": round(step.get("vmem_rd", 0.0) / max(1.0, step["waves"]), 1),
                 "cycles_per_scattered_wave_load": 64.5, "texture_address_units": 256,
                 "ta_busy_frac_of_the_slice": round(step.get("ta_busy", 0.0) / max(1.0, 256 * step.get("gui", 0.0) / 8), 4) if step.get("gui") else None,
                 "note": "cycles_per_scattered_wave_load: tools/probes/l1_gather.hip (profiles/r06_call10/l1_gather.log): 64.5 cycles at 2.4 GHz for a wave-load whose quads of lanes touch four different 64-byte blocks, at 4, 8 or 16 bytes per lane; 17 when every quad stays inside one block"}
json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({"sweeps": res["sweeps"], "valu": res["valu"], "gather": res["gather"], "families": {k: {q: v.get(q) for q in ("dispatches", "frac_active_valu", "valu_busy_of_simd_cycles", "ta_busy_frac", "vmem_rd_insts_per_wave", "ta_busy_cycles_per_wavefront", "l2_hit_rate", "l1_hit_rate", "valu_insts_per_wave")} for k, v in res["families"].items()}}, indent=1))
