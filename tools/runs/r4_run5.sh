#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run5; mkdir -p $O
for i in 1 2; do
  for f in 1 4; do
    echo "== fused $f" >> $O/ab.txt
    BENCH_BACKWARD_FUSED=$f timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
rm -rf gpurun_out/timeline
BENCH_BACKWARD_FUSED=4 timeout 600 python tools/step_timeline.py graph > $O/timeline_f4.txt 2>&1
timeout 600 python tools/fused_timing.py 4 > $O/fused_timing_4.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
echo "pytest rc $?" >> $O/pytest_all.txt
