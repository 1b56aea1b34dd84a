#!/usr/bin/env python3
"""Bitwise comparison of two scripts/ab_dump.py outputs."""
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    same = np.array_equal(a[k], b[k])
    if not same:
        bad += 1
        d = np.abs(a[k].astype(float) - b[k].astype(float)).max()
        print("DIFF", k, "max abs", d)
print("all bit-identical" if bad == 0 else f"{bad} arrays differ")
sys.exit(1 if bad else 0)
