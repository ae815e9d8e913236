#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run49; mkdir -p $O
timeout 600 python tools/train_mvr_cfg3.py > $O/train_mvr_cfg3.json 2> $O/err.txt
DSS_AMD_ENGINE_THREAD=1 timeout 600 python tools/train_mvr_cfg3.py > $O/train_mvr_cfg3_engine_thread.json 2>> $O/err.txt
