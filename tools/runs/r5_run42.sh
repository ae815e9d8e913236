#!/bin/bash
# round 5, call 42: owner mode on the short-list path with the dense full-image plane from fb_prep: parity, per-rank timing owner vs bucket at 8 x configs[1]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run42; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "owner_mode or two_rank" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for rep in 1 2; do for g in owner bucket; do
  for G in 8 4 2; do BAND_GRADIENT=$g BAND_LAYOUTS=balanced,cyclic timeout 600 python tools/band_timing.py $G cfg2 > $O/band${G}_cfg2_${g}_$rep.json 2>/dev/null; done
done; done
tail -n 3 $O/pytest.txt | cut -c1-300
python - $O <<'PY'
import json,sys,os
O=sys.argv[1]
for G in (8,4,2):
  for g in ("owner","bucket"):
    for rep in (1,2):
        d=json.load(open(os.path.join(O,"band%d_cfg2_%s_%d.json"%(G,g,rep))))
        print(G,g,rep,{k:(min(v["graph_us"]),max(v["graph_us"])) for k,v in d.items() if isinstance(v,dict) and "graph_us" in v})
PY
