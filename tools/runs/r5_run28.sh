#!/bin/bash
# (record: the one-launch kNN build and its option / tools were dropped after this call, profiles/r5_b_knn_one_launch_build_experiment.txt)
# round 5, call 28: the one-launch kNN build with two processes on one GPU (wrong h seen in the two-rank test): failure rate with
# workgroup-scope waits vs agent-scope release / acquire fences at the tag hand-offs, and what the fences cost
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run28; mkdir -p $O
for lib in default; do
  if [ $lib = default ]; then unset DSS_HIP_LIBRARY; else export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_$lib.so; fi
  echo "== $lib, 1 process"; timeout 300 python tools/knn_two_process_stress.py 1 2000
  echo "== $lib, 2 processes"; timeout 300 python tools/knn_two_process_stress.py 2 2000
  echo "== $lib, 4 processes"; timeout 300 python tools/knn_two_process_stress.py 4 1000
  timeout 300 python tools/knn_timing.py > $O/knn_timing_$lib.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/knn_timing_$lib.json'))['32k']; print({k:(round(v['kth7_chain_graph_ms']*1e3,1) if isinstance(v,dict) else v) for k,v in d.items()})"
done 2>&1 | grep -v amdgpu.ids | tee $O/stress.txt
