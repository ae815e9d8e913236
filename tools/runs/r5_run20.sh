#!/bin/bash
# round 5, call 20: every launch form of dss_render_backward (DSS_OPT_BACKWARD_FUSED 1 = round-3 sequence, 5 = three launches) under
# the backward tests, with the round-5 band / large-path changes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run20; mkdir -p $O
for form in 1 5; do
  DSS_TEST_BACKWARD_FUSED=$form timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py tests/test_gpu_point_order.py tests/test_gpu_training.py tests/test_gpu_model.py -x -q -m gpu > $O/pytest_form$form.txt 2>&1
  echo "form $form rc $?" >> $O/pytest_form$form.txt
done
tail -3 $O/pytest_form1.txt $O/pytest_form5.txt
