#!/bin/bash
# round 5, call 14: folded exchange + measured choice, named configs on the benched scene, predicted scaling tables
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run14; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rccl_world1.py tests/test_gpu_named_configs.py tests/test_gpu_raster.py -x -q -m gpu -k "rccl or forced or named or two_rank or launches" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
timeout 1200 python tools/predict_scaling.py cfg2 > $O/predicted_scaling_cfg2.json 2> $O/predict_cfg2.err
mkdir -p profiles; cp $O/predicted_scaling_cfg2.json profiles/r5_a_predicted_scaling_cfg2.json
timeout 1200 python tools/predict_scaling.py cfg4 > $O/predicted_scaling_cfg4.json 2> $O/predict_cfg4.err
cp gpurun_out/bench_forced_dist_world1_*.json $O/ 2>/dev/null
tail -3 $O/pytest.txt; cat $O/predicted_scaling_cfg2.json; echo; cat $O/predicted_scaling_cfg4.json; tail -3 $O/predict_cfg2.err $O/predict_cfg4.err
