#!/bin/bash
# round 5, call 10: per-dispatch durations of the fine kernel on a row band at configs[3] (why 0.28 ms for an eighth of the load?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run10; mkdir -p $O
BAND_TRACE=1 BAND_TRACE_LAYOUT=balanced timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_b -o t --output-format csv -- python tools/band_timing.py 8 cfg4 > $O/trace_balanced.log 2>&1
f=$(find /tmp/prof_b -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/fine_dispatches.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
for r in rows:
    n=r.get('Kernel_Name','')
    if 'fine_kernel' in n or 'setup_cell' in n or 'bin_sorted' in n or 'render_backward_kernel' in n:
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000.0
        print('%-40s grid %s wg %s  %.1f us'%(n[:40], r.get('Grid_Size_X',r.get('Grid_Size')), r.get('Workgroup_Size_X',r.get('Workgroup_Size')), d))
PY
head -60 $O/fine_dispatches.txt
