#!/bin/bash
# Developer tool (GPU box): every measurement the round's profiles/ files come from, into gpurun_out/$1/ (default r3).
# Counter passes (--pmc) run on their own with --kernel-trace only; nothing here combines them with other trace domains.
tag=${1:-r6}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench.json 2> $out/bench.err
# (the counters measured by that very run; there is no committed fallback any more)
cp gpurun_out/traffic_fine_kernel.json $out/traffic_fine_kernel.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $out/ks -o b --output-format csv -- python bench.py --mode eager --no-cpu-baseline --no-traffic > /dev/null 2>&1
cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv; rm -rf $out/ks
python tools/step_timeline.py graph > $out/timeline.txt 2>&1
# the causal multi-GPU step at world size 1 (RCCL path forced): its kernels and gaps
BENCH_FORCE_DIST=1 python tools/step_timeline.py graph > $out/timeline_forced_dist_world1.txt 2>&1
python tools/pmc_kernels.py > $out/pmc_sq.txt 2>&1
for c in cfg3 cfg4 cfg5; do python tools/bench_large.py $c 2>/dev/null | grep '^{' >> $out/bench_large.jsonl; done
for c in cfg4 cfg5; do DSS_BENCH_MORTON=1 python tools/bench_large.py $c 2>/dev/null | grep '^{' >> $out/bench_large_morton.jsonl; done
for c in cfg4 cfg5; do
  # the driver's entry point, counters and rocprofv3 averages inside the line; the kernel statistics from the timed region
  # only (--timed-only: bench.py's own pair-counting torch kernels stay out of the trace)
  python bench.py --workload $c --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
  cp gpurun_out/traffic_fine_kernel_$c.json $out/traffic_fine_kernel_$c.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $out/ks -o b --output-format csv -- python bench.py --workload $c --timed-only --mode eager --no-cpu-baseline --no-traffic > /dev/null 2>&1
  cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/kernel_stats_$c.csv; rm -rf $out/ks
done
python tools/fetch_large.py cfg4 > $out/fetch_cfg4.txt 2>&1
python tools/pmc_large.py cfg4 > $out/pmc_sq_cfg4.txt 2>&1
python tools/fine_timing.py cfg2 > $out/fine_timing_cfg2.txt 2>&1
python tools/fine_timing.py cfg4 > $out/fine_timing_cfg4.txt 2>&1
mkdir -p build_ab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o build_ab/valu_rate 2>/dev/null; build_ab/valu_rate > $out/valu_rate.txt 2>&1
for g in 2 4 8; do python tools/band_timing.py $g cfg2 >> $out/band_timing_cfg2.jsonl 2>/dev/null; done
python tools/predict_scaling.py cfg2 > $out/predicted_scaling_cfg2.json 2>/dev/null
BAND_GRADIENT=owner python tools/predict_scaling.py cfg2 > $out/predicted_scaling_cfg2_owner_form.json 2>/dev/null
# the step an UNMODIFIED training loop runs (loss on the gathered image on every rank): per-rank compute at 8 ranks
BAND_LOSS=replicated BAND_LAYOUTS=cyclic python tools/band_timing.py 8 cfg2 2>/dev/null | tail -1 > $out/band_timing_cfg2_replicated_loss.json
for c in cfg4 cfg5; do PREDICT_FLOOR_FROM=$out/predicted_scaling_cfg2.json python tools/predict_scaling.py $c > $out/predicted_scaling_$c.json 2>/dev/null; done
# one emulated rank of the 8-rank step under the kernel trace (metric's configuration: cyclic bands; configs[3]: contiguous)
BAND_TRACE=1 BAND_TRACE_LAYOUT=cyclic rocprofv3 --kernel-trace --stats -d $out/ks -o b --output-format csv -- python tools/band_timing.py 8 cfg2 > /dev/null 2>&1
cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/band_kernel_stats_cfg2_cyclic_rank3.csv; rm -rf $out/ks
BAND_TRACE=1 BAND_TRACE_LAYOUT=bands rocprofv3 --kernel-trace --stats -d $out/ks -o b --output-format csv -- python tools/band_timing.py 8 cfg4 > /dev/null 2>&1
cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/band_kernel_stats_cfg4_bands_rank3.csv; rm -rf $out/ks
BAND_TRACE=1 BAND_TRACE_LAYOUT=bands rocprofv3 --kernel-trace --stats -d $out/ks -o b --output-format csv -- python tools/band_timing.py 8 cfg5 > /dev/null 2>&1
cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/band_kernel_stats_cfg5_bands_rank3.csv; rm -rf $out/ks
for w in headline cfg3 cfg4 cfg5; do python tools/window_stats.py $w 2>/dev/null >> $out/backward_window_stats.jsonl; done
{ python tools/band_fused_timing.py 8 cyclic 3; TIMING_REBUILD=0 python tools/band_fused_timing.py 8 balanced 3; } > $out/band_gather_stamps.txt 2>&1
python tools/setup_timing.py > $out/setup_timing.txt 2>&1
for e in overlap auto; do BENCH_FORCE_DIST=1 BENCH_EXCHANGE=$e python bench.py --gpus 1 --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' > $out/bench_forced_dist_world1_$e.json; done
python tools/knn_timing.py > $out/knn_timing.json 2>/dev/null
python tools/knn_graph_timing.py > $out/knn_graph_timing.txt 2>/dev/null
{ python tools/knn_chain_timeline.py; python tools/knn_chain_timeline.py view; } > $out/knn_chain_timeline.txt 2>/dev/null
for c in cfg2 cfg3 cfg4 cfg5; do python tools/gather_split.py $c >> $out/gather_split.jsonl 2>/dev/null; done
BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_2ranks_gloo_one_gpu.json
# the clustered state of the reference's training loop (tests/golden/trained_cloud_cfg3.npz): kNN without / with the skip structure,
# fine-pass phases, and the kernels of one step under the trace
python tools/clustered_timing.py default@knn=3 default > $out/clustered_timing.txt 2>&1
python tools/fine_timing.py trained > $out/fine_timing_trained.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/ks -o b --output-format csv -- python tools/clustered_timing.py default > /dev/null 2>&1
cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/kernel_stats_trained_cloud.csv; rm -rf $out/ks
[ -n "$COLLECT_REF_LOOP" ] && REF_LEGS=${REF_LEGS:-class} python tools/train_mvr_ref.py $out/ref 40 > $out/train_mvr_ref.log 2>&1
rm -rf gpurun_out/libdss_hip_timing.so gpurun_out/fine_timing.npy gpurun_out/traffic_cfg4 gpurun_out/traffic_cfg5 $out/ref/train_mvr_ref_prof $out/ref/*.pt $out/ref/*.png gpurun_out/timeline gpurun_out/pmc_sq gpurun_out/pmc_large gpurun_out/fetch_large gpurun_out/traffic
ls -la $out
