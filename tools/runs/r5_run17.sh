#!/bin/bash
# round 5, call 17: setup_bin_kernel with every returning operation ahead of the first store: parity, stamps, same-run A/B of the step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run17; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py tests/test_gpu_point_order.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
timeout 600 python tools/setup_timing.py > $O/setup_timing.txt 2>&1
for rep in 1 2; do for lib in new r5pre; do
  if [ $lib = r5pre ]; then export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_r5pre.so; else unset DSS_HIP_LIBRARY; fi
  timeout 300 python bench.py --timed-only --no-cpu-baseline --no-traffic > $O/bench_${lib}_$rep.json 2> $O/bench_${lib}_$rep.err
  timeout 300 python bench.py --timed-only --mode eager --no-cpu-baseline --no-traffic > $O/bench_eager_${lib}_$rep.json 2>> $O/bench_${lib}_$rep.err
done; done
unset DSS_HIP_LIBRARY
tail -3 $O/pytest.txt; tail -12 $O/setup_timing.txt; cat $O/bench_new_*.json $O/bench_r5pre_*.json $O/bench_eager_*.json
