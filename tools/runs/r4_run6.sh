#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_setup.py tests/test_gpu_raster.py -x -q -m gpu -k "fused or graphed or renderer or backward or radius or render" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2; do
  for f in 1 4; do
    echo "== fused $f" >> $O/ab.txt
    BENCH_BACKWARD_FUSED=$f timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
timeout 600 python tools/fused_timing.py 4 > $O/fused_timing_4.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python tools/api_profile.py > $O/api_profile.txt 2>&1
