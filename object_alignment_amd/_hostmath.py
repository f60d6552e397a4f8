"""Host-side float32 matrix product with Blender's mathutils rounding (float products, double accumulate).

Used for the `take_m_with` side effect of the operators (operators/icp_align.py:123-127) and for the landmark
operator's point bookkeeping (operators/align_pick_points.py:184): a handful of 4x4 products.  Not on the hot path.
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


def mat4_mul_vec3(m, v):
    """Matrix @ Vector (4x4 . 3-vector, w = 1): per row, float32 products accumulated in a double, cast to float32
    (same rounding as m4_mul_v3 in csrc/oa_kernels.hpp)."""
    m = np.asarray(m, dtype=np.float32).reshape(4, 4)
    v = np.asarray(v, dtype=np.float32).reshape(3)
    out = np.empty(3, np.float32)
    for i in range(3):
        acc = 0.0
        for k in range(3):
            acc += float(np.float32(m[i, k] * v[k]))
        acc += float(np.float32(m[i, 3] * np.float32(1.0)))
        out[i] = np.float32(acc)
    return out


def mat4_inverted(m):
    """Matrix.inverted(): adjugate / determinant in double in the fixed operation order of m4_inverted
    (csrc/oa_kernels.hpp), rounded to float32.  Raises ValueError for a singular matrix, as mathutils does."""
    a = [float(x) for x in np.asarray(m, dtype=np.float32).reshape(16)]
    s0 = a[0] * a[5] - a[4] * a[1]; s1 = a[0] * a[6] - a[4] * a[2]; s2 = a[0] * a[7] - a[4] * a[3]
    s3 = a[1] * a[6] - a[5] * a[2]; s4 = a[1] * a[7] - a[5] * a[3]; s5 = a[2] * a[7] - a[6] * a[3]
    c5 = a[10] * a[15] - a[14] * a[11]; c4 = a[9] * a[15] - a[13] * a[11]; c3 = a[9] * a[14] - a[13] * a[10]
    c2 = a[8] * a[15] - a[12] * a[11]; c1 = a[8] * a[14] - a[12] * a[10]; c0 = a[8] * a[13] - a[12] * a[9]
    det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0
    if det == 0.0:
        raise ValueError("Matrix.invert(ed): matrix does not have an inverse")
    b = [
        ((a[5] * c5 - a[6] * c4) + a[7] * c3) / det, ((-a[1] * c5 + a[2] * c4) - a[3] * c3) / det,
        ((a[13] * s5 - a[14] * s4) + a[15] * s3) / det, ((-a[9] * s5 + a[10] * s4) - a[11] * s3) / det,
        ((-a[4] * c5 + a[6] * c2) - a[7] * c1) / det, ((a[0] * c5 - a[2] * c2) + a[3] * c1) / det,
        ((-a[12] * s5 + a[14] * s2) - a[15] * s1) / det, ((a[8] * s5 - a[10] * s2) + a[11] * s1) / det,
        ((a[4] * c4 - a[5] * c2) + a[7] * c0) / det, ((-a[0] * c4 + a[1] * c2) - a[3] * c0) / det,
        ((a[12] * s4 - a[13] * s2) + a[15] * s0) / det, ((-a[8] * s4 + a[9] * s2) - a[11] * s0) / det,
        ((-a[4] * c3 + a[5] * c1) - a[6] * c0) / det, ((a[0] * c3 - a[1] * c1) + a[2] * c0) / det,
        ((-a[12] * s3 + a[13] * s1) - a[14] * s0) / det, ((a[8] * s3 - a[9] * s1) + a[10] * s0) / det,
    ]
    return np.array(b, dtype=np.float64).astype(np.float32).reshape(4, 4)
