"""Host-side float32 matrix product with Blender's mathutils rounding (float products, double accumulate).

Only used for the `take_m_with` side effect of the operator (operators/icp_align.py:123-127): 4x4 products of
other objects' matrices with the per-iteration new_mat the engine recorded.  Not on the hot path.
"""
import numpy as np


def mat4_mul(a, b):
    a = np.asarray(a, dtype=np.float32).reshape(4, 4)
    b = np.asarray(b, dtype=np.float32).reshape(4, 4)
    out = np.empty((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            acc = 0.0
            for k in range(4):
                acc += float(np.float32(a[i, k] * b[k, j]))
            out[i, j] = np.float32(acc)
    return out
