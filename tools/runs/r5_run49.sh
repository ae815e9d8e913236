#!/bin/bash
# round 5, call 49: the SQ counter tables again (the tools cut kernel names to their LAST 60 characters: the gather's name, one
# template parameter longer since round 5, lost its "dss::" and was filtered out of r5_a / r5_c / r5_d_pmc_sq*.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_e
python tools/pmc_kernels.py > gpurun_out/r5_e/pmc_sq.txt 2>&1
python tools/pmc_large.py cfg4 > gpurun_out/r5_e/pmc_sq_cfg4.txt 2>&1
python tools/pmc_large.py cfg5 > gpurun_out/r5_e/pmc_sq_cfg5.txt 2>&1
grep -h "render_backward\|fine_kernel" gpurun_out/r5_e/pmc_sq*.txt | cut -c1-220
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_large
