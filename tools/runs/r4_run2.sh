#!/bin/bash
# GPU box, round 4 run 2: single-launch backward -- parity, then A/B of the three modes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "backward or radius or render or cfg2 or bench" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2; do
  for f in 1 2 3; do
    echo "== fused $f" >> $O/ab.txt
    BENCH_BACKWARD_FUSED=$f timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
for f in 2 3; do
  BENCH_BACKWARD_FUSED=$f timeout 600 python tools/step_timeline.py graph > $O/timeline_f$f.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_setup.py tests/test_gpu_named_configs.py tests/test_gpu_training.py tests/test_gpu_model.py -x -q -m gpu > $O/pytest2.txt 2>&1
echo "pytest rc $?" >> $O/pytest2.txt
