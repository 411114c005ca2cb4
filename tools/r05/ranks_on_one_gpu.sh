#!/bin/bash
# (round 5: tools/r04/two_ranks_one_gpu.sh with the needed-only image distribution on the 3-rank run)
# The N-rank path of bench.py on ONE GPU (gloo, ranks share the device): the sharded schedule -- compact per-rank scenes, broadcast of the images, neighbour-only exchange of the
# previous-round maps, per-rank prints -- run on real hardware and compared map by map (sha-1 of every view's final depth) with the single-process run of the same scene.
set -u
OUT=gpurun_out/r05_ranks_on_one_gpu; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OPENMVS_AMD_BENCH_DIGESTS=1
ARGS="--views 12 --width 640 --height 360 --steps 1 --warmup 0 --no-extras --no-cpu-baseline"
timeout 120 python bench.py --gpus 1 $ARGS > "$OUT/one_rank.json" 2> "$OUT/one_rank.err"; echo "1 rank rc $?"
for n in 2 3; do
  # 2 ranks: one broadcast of the image set; 3 ranks: --images needed (rank 0 sends every rank only the views it holds)
  EXTRA=""; [ $n = 3 ] && EXTRA="--images needed"
  OPENMVS_AMD_DIST_BACKEND=gloo timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29540 + n)) bench.py --gpus $n $ARGS $EXTRA > "$OUT/${n}_ranks.json" 2> "$OUT/${n}_ranks.err"; echo "$n ranks rc $?"
done
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r05_ranks_on_one_gpu/one_rank.json").read().strip().splitlines()[-1])
for n in (2, 3):
    try:
        t = json.loads(open("gpurun_out/r05_ranks_on_one_gpu/%d_ranks.json" % n).read().strip().splitlines()[-1])
        same = o["config"]["depth_digests"] == t["config"]["depth_digests"]
        print("%d ranks on one GPU (gloo): %d views, every depth map equals the single-process run: %s; per rank: %s" % (n, len(t["config"]["depth_digests"]), same, [(r["rank"], r["views_per_gpu"], r["foreign_views_held"], r["kernel"], r["exchange_ms_per_step"], r["wait_for_own_estimate_ms_per_step"], r["scene_mb"]) for r in t["config"]["ranks"]]), "; images:", t["config"]["images"])
    except Exception as ex:
        print(n, "ranks: no result:", ex)
PY
for f in "$OUT"/*.err; do tail -2 "$f"; done | tail -12
