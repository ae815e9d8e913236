#!/bin/bash
# round 5, call 37: gather -- trips that lie inside every window of the wavefront walk running offsets without row test / clamp / mask,
# column masks folded into dx2: parity + same-run A/B (headline, configs[2..4])
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run37; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_named_configs.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for rep in 1 2 3; do for lib in new gbase; do
  if [ $lib = gbase ]; then export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_gbase.so; else unset DSS_HIP_LIBRARY; fi
  timeout 300 python bench.py --timed-only --no-cpu-baseline --no-traffic > $O/bench_head_${lib}_$rep.json 2>/dev/null
  if [ $rep != 3 ]; then for c in cfg3 cfg4 cfg5; do timeout 600 python bench.py --workload $c --timed-only --no-cpu-baseline --no-traffic > $O/bench_${c}_${lib}_$rep.json 2>/dev/null; done; fi
done; done
for lib in new gbase; do
  if [ $lib = gbase ]; then export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_gbase.so; else unset DSS_HIP_LIBRARY; fi
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$lib -o b --output-format csv -- python bench.py --timed-only --mode eager --no-cpu-baseline --no-traffic > /dev/null 2>&1
  cp $(find /tmp/ks_$lib -name '*kernel_stats.csv' | head -1) $O/kernel_stats_head_$lib.csv
done
unset DSS_HIP_LIBRARY
tail -n 3 $O/pytest.txt; for f in $O/bench_*.json; do echo -n "$(basename $f) "; cut -c1-110 $f; done; grep -h render_backward $O/kernel_stats_head_*.csv | cut -c1-50,280-400
