#!/bin/bash
# round 5, call 46: predicted scaling at the metric's configuration again (call 44's world-1 measurement was a cold-box outlier: 168 us)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_e
python tools/band_timing.py 1 cfg2 > /dev/null 2>&1
python tools/predict_scaling.py cfg2 > gpurun_out/r5_e/predicted_scaling_cfg2.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/r5_e/predicted_scaling_cfg2.json'));print(d['single_gpu_step_us'],d['multi_step_compute_world1_us'],d['collective_floor'])
for r in d['table']: print(r['G'],r['layout'],r['max_rank_compute_us'],r['predicted_step_us_overlap'],r['predicted_Msplats_per_s_overlap'],r['predicted_speedup_overlap'],r.get('predicted_speedup_overlap_with_link_estimate'))"
