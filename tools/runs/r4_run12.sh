#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run12; mkdir -p $O
for i in 1 2; do
  for w in cfg4 cfg5; do
    for t in 4 2; do
      echo "== $w tpw=$t" >> $O/ab.txt
      BENCH_BACKWARD_TPW=$t timeout 600 python bench.py --workload $w --timed-only --mode eager >> $O/ab.txt 2>&1
    done
  done
done
for t in 0 4 2 1; do
  echo "== cfg3 tpw=$t" >> $O/ab_cfg3.txt
  BENCH_BACKWARD_TPW=$t timeout 600 python tools/bench_large.py cfg3 2>/dev/null | grep '^{' | cut -c1-600 >> $O/ab_cfg3.txt
done
