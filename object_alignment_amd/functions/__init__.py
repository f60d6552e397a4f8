from .general import affine_matrix_from_points, calc_target_matrix, make_pairs, GpuBVH, AlignObject  # noqa: F401
