#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run44; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_knn.py tests/test_gpu_setup.py tests/test_gpu_losses.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
DSS_TEST_BACKWARD_FUSED=1 timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "backward" > $O/pytest_opt1.txt 2>&1
echo "pytest rc $?" >> $O/pytest_opt1.txt
for o in 1 4; do BENCH_BACKWARD_FUSED=$o timeout 300 python bench.py --timed-only 2>/dev/null | grep '^{' >> $O/ab.json; done
timeout 600 python tools/band_timing.py 8 cfg2 > $O/band_timing.json 2>/dev/null
