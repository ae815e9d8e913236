#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run38; mkdir -p $O
for c in cfg4 cfg5; do
  timeout 900 python bench.py --workload $c --no-cpu-baseline 2>$O/bench_$c.err | grep '^{' > $O/bench_$c.json
done
