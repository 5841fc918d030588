#!/bin/bash
# round 6, GPU call 16: bisect of the train-step failure of call 15 (bn_bwd finalize fold vs batched stride-2 class weights)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call16; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "batchnorm" > $O/pytest_bn.log 2>&1; echo "bn hook rc=$?"; tail -3 $O/pytest_bn.log
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -k "shape_sweep or conv_backward" > $O/pytest_a.log 2>&1; echo "batched rc=$?"; tail -3 $O/pytest_a.log
VR_S2W_LOOP=1 timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -k "shape_sweep" > $O/pytest_b.log 2>&1; echo "loop rc=$?"; tail -3 $O/pytest_b.log
