#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run42; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 1200 python -m pytest tests/test_gpu_setup.py tests/test_gpu_reference_loop.py tests/test_gpu_training.py tests/test_gpu_model.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
