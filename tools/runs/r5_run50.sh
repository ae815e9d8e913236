#!/bin/bash
# round 5, call 50: fetch bytes per kernel at configs[3] again (same name truncation as the SQ tools)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_e
python tools/fetch_large.py cfg4 > gpurun_out/r5_e/fetch_cfg4.txt 2>&1
grep -h "render_backward\|fine_kernel" gpurun_out/r5_e/fetch_cfg4.txt | cut -c1-200
rm -rf gpurun_out/fetch_large
