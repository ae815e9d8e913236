#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_point_order.py -x -q -m gpu > $O/pytest_order.txt 2>&1
echo "pytest rc $?" >> $O/pytest_order.txt
for c in cfg4 cfg5; do
  for k in 0 16; do
    BENCH_ORDER_REFRESH=$k timeout 600 python tools/bench_large.py $c 2>$O/err_$c_$k.txt | grep '^{' >> $O/bench_large_order.jsonl
  done
done
BENCH_ORDER_REFRESH=16 timeout 600 python tools/bench_large.py cfg3 2>/dev/null | grep '^{' >> $O/bench_large_order.jsonl
cd /tmp
BENCH_ORDER_REFRESH=16 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/ks -o b --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_large.py cfg4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats_cfg4_order16.csv; rm -rf $O/ks
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-traffic 2> $O/forced.err | grep '^{' > $O/bench_forced_dist_world1.json
