#!/bin/bash
# round 5, call 18: the round's profile set (tools/collect_profiles.sh r5_a)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 2400 bash tools/collect_profiles.sh r5_a > gpurun_out/r5_a_collect.log 2>&1
tail -40 gpurun_out/r5_a_collect.log
