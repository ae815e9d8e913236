#!/bin/bash
# round 5, call 21: BAND gather A/B -- wait for the medians first, then whole tasks (one filter, one record read) vs the two-phase form
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run21; mkdir -p $O
export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_waitfirst.so
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "band or cyclic or two_rank" > $O/pytest_wf.txt 2>&1
echo "pytest rc $?" >> $O/pytest_wf.txt
unset DSS_HIP_LIBRARY
for rep in 1 2; do for lib in new wf; do
  if [ $lib = wf ]; then export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_waitfirst.so; else unset DSS_HIP_LIBRARY; fi
  for tpw in 0 2; do
    BAND_TPW=$tpw BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_${lib}_tpw${tpw}_$rep.json 2> $O/band8_${lib}_$rep.err
  done
done; done
unset DSS_HIP_LIBRARY
tail -n 3 $O/pytest_wf.txt; for f in $O/band8_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("balanced","cyclic"):
    if k in d: print("  ",k,"graph_us",d[k]["graph_us"])
PY
done
