#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_point_order.py -x -q -m gpu > $O/pytest_order.txt 2>&1
echo "pytest rc $?" >> $O/pytest_order.txt
for c in cfg4 cfg5; do
  for k in 0 16; do
    BENCH_ORDER_REFRESH=$k timeout 600 python tools/bench_large.py $c 2>$O/err_${c}_$k.txt | grep '^{' >> $O/bench_large_order.jsonl
  done
done
cd /tmp
for c in cfg4 cfg5; do
BENCH_ORDER_REFRESH=16 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/ks -o b --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_large.py $c > /dev/null 2>&1
cp $(find $GRAFT_REPO_ROOT/$O/ks -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_${c}_order16.csv; rm -rf $GRAFT_REPO_ROOT/$O/ks
done
