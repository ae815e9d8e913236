cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/clprof -o cl --output-format csv -- python $GRAFT_REPO_ROOT/tools/clustered_timing.py default > /dev/null 2>&1
python - <<'P'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/clprof/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    if 'knn' in r['Name']: print('%-64s calls %5s avg %9.1f us %5s%%'%(r['Name'].replace('void ','')[:64],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
P
rm -f $GRAFT_REPO_ROOT/gpurun_out/clprof/*/*kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/clprof/*kernel_trace.csv
