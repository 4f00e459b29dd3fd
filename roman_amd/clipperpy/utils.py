"""clipperpy.utils"""
import numpy as np


def create_all_to_all(n1, n2):
    """clipperpy.utils.create_all_to_all ([REF roman/align/object_registration.py:41],
    [REF roman/align/dist_reg_with_pruning.py:72]): (n1*n2, 2) int32, row i*n2+j = (i, j).
    Pure index arithmetic (the C ABI twin is roman_create_all_to_all)."""
    n1, n2 = int(n1), int(n2)
    A = np.empty((n1 * n2, 2), dtype=np.int32)
    if n1 * n2:
        A[:, 0] = np.repeat(np.arange(n1, dtype=np.int32), n2)
        A[:, 1] = np.tile(np.arange(n2, dtype=np.int32), n1)
    return A
