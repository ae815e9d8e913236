#!/bin/bash
# round 5, call 40: (large path: rs-aware dense plane in full layout) owner gradient exchange in the bench step: two-rank test, RCCL world-1 tests, per-rank timing owner vs bucket
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run40; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_rccl_world1.py -x -q -m gpu -k "two_rank or rccl or world1 or launch or owner" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for g in owner bucket; do
  BAND_GRADIENT=$g BAND_LAYOUTS=balanced,cyclic timeout 600 python tools/band_timing.py 8 cfg2 > $O/band8_cfg2_$g.json 2>/dev/null
  BAND_GRADIENT=$g BAND_LAYOUTS=bands timeout 900 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4_$g.json 2>/dev/null
  BAND_GRADIENT=$g BAND_LAYOUTS=bands timeout 900 python tools/band_timing.py 8 cfg5 > $O/band8_cfg5_$g.json 2>/dev/null
done
tail -n 4 $O/pytest.txt | cut -c1-300
python - $O <<'PY'
import json,sys,os
O=sys.argv[1]
for c in ("cfg2","cfg4","cfg5"):
    for g in ("owner","bucket"):
        d=json.load(open(os.path.join(O,"band8_%s_%s.json"%(c,g))))
        print(c,g,{k:v["graph_us"] for k,v in d.items() if isinstance(v,dict) and "graph_us" in v})
PY
