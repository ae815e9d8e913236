#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run48; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rccl_world1.py tests/test_gpu_raster.py -x -q -m gpu -k "rccl or bench or rank" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
cp gpurun_out/bench_forced_dist_world1.json $O/ 2>/dev/null
BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-traffic 2>$O/gloo2.err | grep '^{' > $O/bench_2ranks_gloo.json
