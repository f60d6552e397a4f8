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


def _det2(a, b, c, d):
    return np.float32(np.float32(a * d) - np.float32(b * c))


def _det3(a1, a2, a3, b1, b2, b3, c1, c2, c3):
    t = np.float32(np.float32(a1 * _det2(b2, b3, c2, c3)) - np.float32(b1 * _det2(a2, a3, c2, c3)))
    return np.float32(t + np.float32(c1 * _det2(a2, a3, b2, b3)))


def mat4_inverted(m):
    """Matrix.inverted() as Blender computes it (mathutils matrix_invert_internal: float determinant_m4, float adjoint_m4_m4,
    element / det -- every operation rounded to float32, left to right), the rule of m4_inverted in csrc/oa_kernels.hpp and of
    the oracle.  Raises ValueError for a singular matrix, as mathutils does."""
    A = np.asarray(m, dtype=np.float32).reshape(4, 4)
    with np.errstate(all="ignore"):
        # Blender's column-major m[i][j] = element (row j, column i)
        (a1, b1, c1, d1), (a2, b2, c2, d2), (a3, b3, c3, d3), (a4, b4, c4, d4) = (tuple(np.float32(A[j, i]) for j in range(4)) for i in range(4))
        det = np.float32(np.float32(np.float32(np.float32(a1 * _det3(b2, b3, b4, c2, c3, c4, d2, d3, d4)) -
                                               np.float32(b1 * _det3(a2, a3, a4, c2, c3, c4, d2, d3, d4))) +
                                    np.float32(c1 * _det3(a2, a3, a4, b2, b3, b4, d2, d3, d4))) -
                         np.float32(d1 * _det3(a2, a3, a4, b2, b3, b4, c2, c3, c4)))
        if det == 0.0:
            raise ValueError("Matrix.invert(ed): matrix does not have an inverse")
        R = np.empty((4, 4), np.float32)
        R[0, 0] = _det3(b2, b3, b4, c2, c3, c4, d2, d3, d4)
        R[1, 0] = -_det3(a2, a3, a4, c2, c3, c4, d2, d3, d4)
        R[2, 0] = _det3(a2, a3, a4, b2, b3, b4, d2, d3, d4)
        R[3, 0] = -_det3(a2, a3, a4, b2, b3, b4, c2, c3, c4)
        R[0, 1] = -_det3(b1, b3, b4, c1, c3, c4, d1, d3, d4)
        R[1, 1] = _det3(a1, a3, a4, c1, c3, c4, d1, d3, d4)
        R[2, 1] = -_det3(a1, a3, a4, b1, b3, b4, d1, d3, d4)
        R[3, 1] = _det3(a1, a3, a4, b1, b3, b4, c1, c3, c4)
        R[0, 2] = _det3(b1, b2, b4, c1, c2, c4, d1, d2, d4)
        R[1, 2] = -_det3(a1, a2, a4, c1, c2, c4, d1, d2, d4)
        R[2, 2] = _det3(a1, a2, a4, b1, b2, b4, d1, d2, d4)
        R[3, 2] = -_det3(a1, a2, a4, b1, b2, b4, c1, c2, c4)
        R[0, 3] = -_det3(b1, b2, b3, c1, c2, c3, d1, d2, d3)
        R[1, 3] = _det3(a1, a2, a3, c1, c2, c3, d1, d2, d3)
        R[2, 3] = -_det3(a1, a2, a3, b1, b2, b3, d1, d2, d3)
        R[3, 3] = _det3(a1, a2, a3, b1, b2, b3, c1, c2, c3)
        return (R / det).T.astype(np.float32).copy()          # out[row j][column i] = R[i][j] / det
