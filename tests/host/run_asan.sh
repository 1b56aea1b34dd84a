#!/bin/bash
# Builds tests/host/host_sweep.cpp for x86 with AddressSanitizer + UBSan and runs tests/host/check_host_sweeps.py under it
# (the compile takes ~7 minutes: the whole device file is instantiated for the host).
#   bash tests/host/run_asan.sh > profiles/rNN_host_device_functions_asan_ubsan.log 2>&1
set -e
cd "$(dirname "$0")/../.."
OBJ=tests/host/host_sweep_asan.o
OUT=tests/host/libhost_sweep_asan.so
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
if [ ! -f $OBJ ] || [ tests/host/host_sweep.cpp -nt $OBJ ] || [ robot_mpcs_amd/csrc/rmpc_kernels.hip -nt $OBJ ]; then
  /opt/rocm/bin/hipcc -x hip --cuda-host-only -std=c++17 -O1 -g -fno-omit-frame-pointer $SAN -ferror-limit=0 -fPIC \
    -DRMPC_SOURCE_HASH='"host"' -DRMPC_DEV_VARIANTS=0x25 -Irobot_mpcs_amd/csrc -Iinclude -c tests/host/host_sweep.cpp -o $OBJ
fi
# (the symbol the registration code refers to: the device code bundle of this file -- there is none)
FAT=$(nm -u $OBJ | awk '/__hip_fatbin_/{print $2}' | head -1)
echo "const char ${FAT:-rmpc_host_no_fatbin}[64] = {0};" > tests/host/fatbin_stub.c
gcc -c -fPIC tests/host/fatbin_stub.c -o tests/host/fatbin_stub.o
/opt/rocm/bin/hipcc $SAN -shared -Wl,-Bsymbolic $OBJ tests/host/fatbin_stub.o -o $OUT
echo "# built $OUT (hipcc --cuda-host-only -O1 -g $SAN)"
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$ASAN" ] || ASAN=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan*x86_64*.so" | head -1)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python tests/host/check_host_sweeps.py $OUT
