#!/bin/bash
# (record: the one-launch kNN build and its option / tools were dropped after this call, profiles/r5_b_knn_one_launch_build_experiment.txt)
# round 5, call 23: kNN grid build in one launch: parity + chain timing A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_setup.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
timeout 600 python tools/knn_timing.py > $O/knn_timing.json 2> $O/knn_timing.err
rocprofv3 --kernel-trace --stats -d /tmp/ks -o b --output-format csv -- python tools/knn_timing.py > /dev/null 2>&1
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats_knn.csv
tail -n 5 $O/pytest.txt; cat $O/knn_timing.json; tail -3 $O/knn_timing.err; grep -i knn $O/kernel_stats_knn.csv | cut -c1-200
