# development aid (round 4): iterates after 1 .. 16 iterations with two development libraries (dev / dev2) compared: the first iteration at which a changed recursion gives another step
export RMPC_ALLOW_STALE=1
mkdir -p gpurun_out
for l in dev dev2; do RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_$l.so timeout -k 10 200 python tests/tools/dev_ab_iter.py gpurun_out/abit_$l.npz ${1:-cfg3} ${2:-4} 2>&1 | grep -v amdgpu; done
python - <<'P'
import numpy as np
a=np.load('gpurun_out/abit_dev.npz'); b=np.load('gpurun_out/abit_dev2.npz')
np.set_printoptions(precision=3, linewidth=220, suppress=False)
for it in (1,2,3,5,8,12,16):
    za, zb = a[f'z{it}'], b[f'z{it}']
    d=np.abs(za-zb)
    dm=d.reshape(d.shape[0],-1).max(axis=1)
    print('iter', it, 'max diff', d.max(), 'instances over 1e-9:', np.flatnonzero(dm>1e-9)[:20], 'flags differ', np.flatnonzero(a[f'f{it}']!=b[f'f{it}'])[:20])
    if it==1:
        i=np.unravel_index(d.argmax(), d.shape); print(' at', i)
        print(' per-variable max diff over stages (instance 0):', d[0].max(axis=0))
        print(' per-stage max diff (instance 0):', d[0].max(axis=1))
P
