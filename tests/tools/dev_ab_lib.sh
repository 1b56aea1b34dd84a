#!/bin/bash
# development aid: quick_time.py of one configuration with the shipped library and with a development build
#   tests/tools/dev_ab_lib.sh cfg4 robot_mpcs_amd/csrc/librmpc_hip_wpe2.so
cfg=$1; lib=$2
mkdir -p gpurun_out
RMPC_ALLOW_STALE=1 python tests/tools/quick_time.py $cfg > gpurun_out/abl_base.log 2>&1 &&
RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/$lib python tests/tools/quick_time.py $cfg > gpurun_out/abl_dev.log 2>&1
cat gpurun_out/abl_base.log gpurun_out/abl_dev.log
