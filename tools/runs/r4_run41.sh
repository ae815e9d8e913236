#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run41; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err
DSS_AMD_ENGINE_THREAD=0 timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/bench_env0.json 2>> $O/bench.err
