#!/usr/bin/env python3
"""Development aid (round 3, review item 2): the arm's generated view, which round 2 dropped because "every build came
out wrong in a different way".  Rebuilds that view in a development library and looks for the first difference.

  python tests/tools/dev_arm_view.py build [extra hipcc flags]
      (no GPU needed) writes rmpc_spec_gen.hpp with an additional view of cfg4 (SpecArm), builds
      robot_mpcs_amd/csrc/librmpc_hip_dev.so with the arm's kernels only, restores the committed header
  RMPC_ALLOW_STALE=1 RMPC_LIB_PATH=$PWD/robot_mpcs_amd/csrc/librmpc_hip_dev.so python tests/tools/dev_arm_view.py run
      (GPU) first sweep of the view against the runtime tables entry by entry (rmpc_debug_sweep), then whole solves:
      runtime tables once, the view three times (the LDS / scratch / workspace poisoned before the second and third)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "robot_mpcs_amd", "csrc")


def build(extra):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_specs as g
    g.SPEC_CONFIGS.append(("SpecArm", "cfg4"))
    text = g.render()
    committed = open(g.HEADER).read()
    try:
        with open(g.HEADER, "w") as f:
            f.write(text)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               '-DRMPC_SOURCE_HASH="dev"', "-DRMPC_DEV_VARIANTS=0x4"] + extra + ["-o", "librmpc_hip_dev.so", "rmpc_kernels.hip"]
        subprocess.check_call(cmd, cwd=CSRC)
    finally:
        with open(g.HEADER, "w") as f:
            f.write(committed)
    print("built", os.path.join(CSRC, "librmpc_hip_dev.so"))


def first_diff(name, a, b):
    if np.array_equal(a, b):
        print(f"  {name}: identical")
        return
    d = np.abs(a - b)
    bad = ~np.isclose(a, b, rtol=1e-12, atol=1e-12) | np.isnan(a) != np.isnan(b)
    idx = np.argwhere(bad)
    print(f"  {name}: max abs diff {np.nanmax(d):.3e}, entries beyond 1e-12: {len(idx)} of {a.size}"
          + (f", first at (instance, stage, ...) = {tuple(idx[0])}: {a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}" if len(idx) else ""))
    if len(idx):
        stages = np.unique(idx[:, 1])
        print(f"    stages with differences: {stages.tolist()}; instances: {len(np.unique(idx[:, 0]))}")


def run():
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import make_scenario
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    sc = make_scenario("cfg4", B=B, seed=3)
    os.environ["RMPC_NO_SPEC"] = "1"
    s_rt = Solver(sc.desc, max_batch=B)
    del os.environ["RMPC_NO_SPEC"]
    s_v = Solver(sc.desc, max_batch=B)
    print("views:", repr(s_rt.spec_name()), repr(s_v.spec_name()))
    a = s_rt.debug_sweep(sc.xinit, sc.x0, sc.params)
    for rep in range(2):
        b = s_v.debug_sweep(sc.xinit, sc.x0, sc.params)
        print(f"first sweep, view (run {rep}) vs runtime tables:")
        for k in ("f", "g", "rc", "q0", "q1", "Q"):
            first_diff(k, a[k], b[k])
    r = s_rt.solve(sc.xinit, sc.x0, sc.params)
    print("runtime tables: flags", np.unique(r["exitflag"], return_counts=True), "iters mean", r["iters"].mean())
    prev = None
    for rep in range(3):
        if rep:
            s_v.poison_lds()
        v = s_v.solve(sc.xinit, sc.x0, sc.params)
        same = np.array_equal(v["exitflag"], r["exitflag"]) and np.array_equal(v["iters"], r["iters"])
        print(f"view run {rep}: flags", np.unique(v["exitflag"], return_counts=True), "iters mean", v["iters"].mean(),
              "| flags+iters equal to runtime tables:", same, "| max |z - z_rt|", float(np.nanmax(np.abs(v["z"] - r["z"]))),
              "| bitwise equal to previous view run:", None if prev is None else bool(np.array_equal(prev["z"], v["z"])))
        prev = v
    s_rt.close(); s_v.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run()
