#!/bin/bash
# round 5, call 24: one-column-slot occupancy path of the gather (configs 3-5), kNN build A/B test; parity + same-run A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run24; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_knn.py tests/test_gpu_named_configs.py tests/test_gpu_raster.py tests/test_gpu_point_order.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for rep in 1 2; do for lib in default one0 one8 one10; do
  if [ $lib = default ]; then unset DSS_HIP_LIBRARY; else export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_$lib.so; fi
  for c in cfg4 cfg5; do timeout 600 python bench.py --workload $c --timed-only --no-cpu-baseline --no-traffic > $O/bench_${c}_${lib}_$rep.json 2> $O/bench_${c}_${lib}_$rep.err; done
done; done
for lib in default one0; do
  if [ $lib = default ]; then unset DSS_HIP_LIBRARY; else export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_$lib.so; fi
  for c in cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks_${c}_$lib -o b --output-format csv -- python bench.py --workload $c --timed-only --mode eager --no-cpu-baseline --no-traffic > /dev/null 2>&1
  cp $(find /tmp/ks_${c}_$lib -name '*kernel_stats.csv' | head -1) $O/kernel_stats_${c}_$lib.csv
  done
done
unset DSS_HIP_LIBRARY
tail -n 3 $O/pytest.txt; for f in $O/bench_*.json; do echo -n "$(basename $f) "; cut -c1-120 $f; done; grep -h render_backward $O/kernel_stats_*.csv | cut -c1-60,300-420
