#!/bin/bash
# round 5, call 43: owner gradient exchange as the default everywhere: RCCL world-1 tests, two-rank / self-launch tests, forced-dist line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run43; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_rccl_world1.py -x -q -m gpu -k "two_rank or rccl or world1 or launch or owner" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for g in owner bucket owner bucket; do BENCH_GRADIENT=$g BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --timed-only --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' | cut -c1-140; done | tee $O/forced.txt
tail -n 3 $O/pytest.txt | cut -c1-300
