#!/bin/bash
# round 5, call 35: + two steps of measured-time rebalancing; contiguous bands fitted to measured per-rank times (fitted_bounds) at configs[3] / [4], 8 and 4 ranks
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run35; mkdir -p $O
for c in cfg4 cfg5; do for g in 8; do
  BAND_LAYOUTS=bands,balanced,fitted,rebalanced,rebalanced2 timeout 900 python tools/band_timing.py $g $c > $O/band${g}_$c.json 2> $O/band${g}_$c.err
  python - $O/band${g}_$c.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["workload"], "G", d["G"], "single", d["single_gpu_step_us"])
for k in ("bands","balanced","fitted","rebalanced","rebalanced2"):
    print("  ", k.ljust(9), d[k]["graph_us"], "max", d[k]["graph_max_us"])
print("   balanced_bounds", d["balanced_bounds"]); print("   fitted_bounds  ", d["fitted_bounds"], d["fitted_model_F_a_b"])
PY
done; done
