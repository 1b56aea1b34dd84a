#!/bin/bash
# phase stamps of k_fused and k_fused_arm for the round's profile file (library built with: scripts/dev_build.sh 0x25 -DRMPC_STAMPS,
# moved to csrc/librmpc_hip_stamps.so)       usage: scripts/stamps_round.sh > profiles/rNN_fused_phase_stamps.txt
export RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_stamps.so
echo "# scripts/fused_stamps.py, library built with scripts/dev_build.sh 0x25 -DRMPC_STAMPS (s_memtime around the phases of k_fused; cycles per wavefront pass)"
echo "## generated views (default): the step lengths are formed inside the sweep call, the step column is empty"
python scripts/fused_stamps.py cfg2 4096 && python scripts/fused_stamps.py cfg2 128 || exit 1
echo "## runtime tables (RMPC_NO_SPEC=1; the boxer always)"
RMPC_NO_SPEC=1 python scripts/fused_stamps.py cfg2 4096 && RMPC_NO_SPEC=1 python scripts/fused_stamps.py cfg2 128 || exit 1
python scripts/fused_stamps.py cfg3 4096 && python scripts/fused_stamps.py cfg3 128
echo "## k_fused_arm (tests/tools/dev_arm_fused_stamps.py: cycles per instance pass; the sweep call's sections from s_memtime stamps inside it)"
python tests/tools/dev_arm_fused_stamps.py cfg4 1024 && python tests/tools/dev_arm_fused_stamps.py cfg4 64
