#!/bin/bash
# round 5, call 44: the files of the final profile set that the owner default changes at the metric's configuration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5_e; mkdir -p $out
for g in 2 4 8; do python tools/band_timing.py $g cfg2 >> $out/band_timing_cfg2.jsonl 2>/dev/null; done
python tools/predict_scaling.py cfg2 > $out/predicted_scaling_cfg2.json 2>/dev/null
BAND_TRACE=1 BAND_TRACE_LAYOUT=cyclic rocprofv3 --kernel-trace --stats -d $out/ks -o b --output-format csv -- python tools/band_timing.py 8 cfg2 > /dev/null 2>&1
cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/band_kernel_stats_cfg2_cyclic_rank3.csv; rm -rf $out/ks
for e in overlap auto; do BENCH_FORCE_DIST=1 BENCH_EXCHANGE=$e python bench.py --gpus 1 --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' > $out/bench_forced_dist_world1_$e.json; done
BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_2ranks_gloo_one_gpu.json
ls -la $out
