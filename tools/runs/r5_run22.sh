#!/bin/bash
# round 5, call 22: candidate records in the order's positions (cell-ordered path): parity + configs[3]/[4] lines, same-run A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run22; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_named_configs.py tests/test_gpu_point_order.py tests/test_gpu_raster.py -x -q -m gpu -k "named or order or large_inputs or band_outputs or lean or far_camera or overflow" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for rep in 1 2; do for lib in new r5pre; do
  if [ $lib = r5pre ]; then export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_r5pre.so; else unset DSS_HIP_LIBRARY; fi
  for c in cfg4 cfg5; do timeout 600 python bench.py --workload $c --timed-only --no-cpu-baseline --no-traffic > $O/bench_${c}_${lib}_$rep.json 2> $O/bench_${c}_${lib}_$rep.err; done
done; done
unset DSS_HIP_LIBRARY
for c in cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o b --output-format csv -- python bench.py --workload $c --timed-only --mode eager --no-cpu-baseline --no-traffic > /dev/null 2>&1
  cp $(find /tmp/ks_$c -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$c.csv
done
tail -n 3 $O/pytest.txt; for f in $O/bench_*.json; do echo -n "$f "; cat $f; done
