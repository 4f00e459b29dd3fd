#!/usr/bin/env python3
"""Which accumulation order does v_mfma_f64_16x16x4_f64 use?  Compares the device's cosine matrix
(roman_debug_cosine -> k_cos) bit for bit with the oracle's stated-order cosine, and with a few alternative
hypotheses computed here with exact rational arithmetic.  Run on the GPU box: python tools/gpu_mfma_order.py"""
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from roman_amd import _abi
from roman_amd.runtime import Context
from oracle import oracle as orc
from conftest import ulp_diff


def fma(a, b, c):
    """correctly rounded a*b+c via exact rationals"""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def dot_h(a, b, order, fused=True):
    acc = 0.0
    for k in order:
        acc = fma(a[k], b[k], acc) if fused else (acc + a[k] * b[k])
    return acc


def orders(d):
    o = {}
    o["stated t-outer g-inner"] = [k0 + 4 * g + t for k0 in range(0, d, 16) for t in range(4) for g in range(4) if k0 + 4 * g + t < d]
    o["t-outer g-inner reversed g"] = [k0 + 4 * g + t for k0 in range(0, d, 16) for t in range(4) for g in (3, 2, 1, 0) if k0 + 4 * g + t < d]
    o["sequential"] = list(range(d))
    return o


def main():
    ctx = Context(0)
    rng = np.random.default_rng(7)
    for d in (4, 8, 16, 37, 64, 512):
        n1, n2 = 24, 20
        P = _abi.RomanParams.default(); P.cos_feature_dim = d
        D1 = rng.standard_normal((n1, 3 + d)); D2 = rng.standard_normal((n2, 3 + d))
        got = ctx.debug_cosine(P, D1, D2)
        ref = np.array([[orc.cosine(D1[i, 3:], D2[j, 3:]) for j in range(n2)] for i in range(n1)])
        u = ulp_diff(got, ref)
        print(f"d={d:4d}: device vs oracle stated order: max ulp {u.max()}, mismatches {(u > 0).sum()} / {u.size}")
        if u.max() > 0 and d <= 64:
            # try the hypotheses on the raw dot (norms via the oracle's stated order: recompute cos from the parts)
            for name, od in orders(d).items():
                for fused in (True, False):
                    bad = 0
                    for i in range(6):
                        for j in range(6):
                            a, b = D1[i, 3:], D2[j, 3:]
                            dot = dot_h(a, b, od, fused)
                            na = orc.cosine(a, a); nb = orc.cosine(b, b)   # == 1 up to rounding; not the norm: skip normalisation
                            c = dot / (np.sqrt(dot_h(a, a, od, True)) * np.sqrt(dot_h(b, b, od, True)))
                            bad += (c != got[i, j])
                    print(f"      hypothesis {name:32s} fused={fused}: {bad}/36 differ")
    ctx.close()


if __name__ == "__main__":
    main()
