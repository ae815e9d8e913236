#!/bin/bash
# round 5, call 6: BAND gather with the share's records in LDS and the whole window in one trip (RB 14 vs 8 A/B)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "band or cyclic or rank or backward" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for lib in new rb8; do
  if [ $lib = rb8 ]; then export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_rb8.so; else unset DSS_HIP_LIBRARY; fi
  for tpw in 0 1 4; do
    BAND_TPW=$tpw BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_${lib}_tpw$tpw.json 2> $O/band8_${lib}_tpw$tpw.err
  done
done
unset DSS_HIP_LIBRARY
timeout 300 python tools/band_fused_timing.py 8 balanced 3 > $O/stamps_balanced.txt 2>&1
tail -3 $O/pytest.txt; for f in $O/band8_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("balanced","cyclic"):
    if k in d: print("  ",k,"graph_us",d[k]["graph_us"])
PY
done; tail -14 $O/stamps_balanced.txt
