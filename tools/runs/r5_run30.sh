#!/bin/bash
# round 5, call 30: binning chunks without a live record leave early (band ranks): parity, per-rank timing at configs[3] / [4] (G = 8)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run30; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_named_configs.py tests/test_gpu_point_order.py tests/test_gpu_raster.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for c in cfg4 cfg5; do
  BAND_LAYOUTS=bands timeout 900 python tools/band_timing.py 8 $c > $O/band8_$c.json 2> $O/band8_$c.err
  timeout 600 python bench.py --workload $c --timed-only --no-cpu-baseline --no-traffic > $O/bench_$c.json 2>/dev/null
done
tail -n 3 $O/pytest.txt; cat $O/band8_*.json | cut -c1-900; cat $O/bench_cfg*.json | cut -c1-160
