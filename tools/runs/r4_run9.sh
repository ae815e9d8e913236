#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_named_configs.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2; do
  for w in cfg4 cfg5; do
    for sw in 1 0; do
      echo "== $w two_slot_always=$sw" >> $O/ab.txt
      BENCH_BACKWARD_SWEEP=$sw timeout 600 python bench.py --workload $w --timed-only --mode eager >> $O/ab.txt 2>&1
    done
  done
done
for sw in 1 0; do
  echo "== cfg3 two_slot_always=$sw" >> $O/ab_large.txt
  BENCH_BACKWARD_SWEEP=$sw timeout 600 python tools/bench_large.py cfg3 >> $O/ab_large.txt 2>&1
  echo "== cfg4 two_slot_always=$sw" >> $O/ab_large.txt
  BENCH_BACKWARD_SWEEP=$sw timeout 600 python tools/bench_large.py cfg4 >> $O/ab_large.txt 2>&1
done
for i in 1 2; do BENCH_BACKWARD_FUSED=4 timeout 300 python bench.py --timed-only --steps 200 >> $O/ab_cfg2.txt 2>&1; done
timeout 1300 python -m pytest tests/test_gpu_reference_loop.py -x -q -m gpu -s -k "converges" > $O/pytest_ref5000.txt 2>&1
echo "pytest rc $?" >> $O/pytest_ref5000.txt
timeout 600 python tools/fetch_calibration.py > $O/fetch_calibration.txt 2>&1
