#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run32; mkdir -p $O
for c in cfg4 cfg5; do
  for lib in r4c lw5 lw4; do
    echo "== $c $lib" >> $O/ab.txt
    DSS_HIP_LIBRARY=$PWD/build_ab/libdss_$lib.so timeout 600 python tools/bench_large.py $c 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('ms_per_step_eager','Msplats_per_s','backward_gather_ms','backward_total_ms','fine_kernel_ms')})" >> $O/ab.txt
  done
done
