#!/bin/bash
# development aid: quick one-unit build of a subset of the kernel variants into csrc/librmpc_hip_dev.so
#   scripts/dev_build.sh 0x4 [-DRMPC_STAMPS ...]     (bit i = variant i of RMPC_VARIANTS: 0 point robot, 2 panda, 5 boxer + slack,
#                                                      6 .. 9 chains n = 2, 4, 5, 6)
# use with RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so
set -e
cd "$(dirname "$0")/../robot_mpcs_amd/csrc"
mask=${1:-0x3f}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRMPC_SOURCE_HASH='"dev"' -DRMPC_DEV_VARIANTS=$mask "$@" \
  -o librmpc_hip_dev.so rmpc_kernels.hip
ls -la librmpc_hip_dev.so
