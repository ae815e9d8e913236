#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_setup.py -x -q -m gpu -k "fused or graphed or renderer" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 900 python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 900 python bench.py --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python tools/band_timing.py 8 cfg4 > $O/band_timing_cfg4.json 2> $O/band_timing_cfg4.err
timeout 900 python tools/band_timing.py 8 cfg2 > $O/band_timing_cfg2.json 2> $O/band_timing_cfg2.err
timeout 1300 python -m pytest tests/test_gpu_reference_loop.py -x -q -m gpu -s -k "converges" > $O/pytest_ref5000.txt 2>&1
echo "pytest rc $?" >> $O/pytest_ref5000.txt
