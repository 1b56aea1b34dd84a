#!/usr/bin/env python3
"""Comparison of two tests/tools/ab_dump.py outputs: exit flags and iteration counts must be identical, floating-point
results bitwise identical or (with --tol T) within T -- the fast in-LDS recursion of the fused kernel contracts its
multiply-adds differently from the pass kernels' generic path, a rounding-level difference."""
import sys

import numpy as np

args = [a for a in sys.argv[1:] if not a.startswith("--")]
tol = float(sys.argv[sys.argv.index("--tol") + 1]) if "--tol" in sys.argv else 0.0
args = [a for a in args if a != str(tol) and a != sys.argv[sys.argv.index("--tol") + 1]] if "--tol" in sys.argv else args
a, b = np.load(args[0]), np.load(args[1])
bad = 0
worst = 0.0
for k in a.files:
    if np.array_equal(a[k], b[k]):
        continue
    d = float(np.abs(a[k].astype(float) - b[k].astype(float)).max())
    worst = max(worst, d)
    integer = a[k].dtype.kind in "iu"
    if integer or d > tol:
        bad += 1
        print("DIFF", k, "max abs", d, "(integer result)" if integer else "")
print("all bit-identical" if worst == 0.0 else (f"flags and iteration counts identical, max abs difference {worst:.3e}" if bad == 0 else f"{bad} arrays differ"))
sys.exit(1 if bad else 0)
