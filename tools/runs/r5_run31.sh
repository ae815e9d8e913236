#!/bin/bash
# round 5, call 31: kernel split of one rank (rank 3 of 8, contiguous bands) at configs[3] / [4]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run31; mkdir -p $O
for c in cfg4 cfg5; do
  BAND_TRACE=1 BAND_TRACE_LAYOUT=bands BAND_TRACE_RANK=3 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o b --output-format csv -- python tools/band_timing.py 8 $c > /dev/null 2>&1
  cp $(find /tmp/ks_$c -name '*kernel_stats.csv' | head -1) $O/band_kernel_stats_${c}_bands_rank3.csv
  head -16 $O/band_kernel_stats_${c}_bands_rank3.csv | cut -c1-90,200-330
done
