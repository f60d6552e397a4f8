"""Synthetic meshes / clouds for BASELINE.json's configs.

Pure numpy, deterministic (PCG64 seeds).  No reference asset exists for any of
these; they are the build's own inputs, shared by tests, the golden generator
and bench.py so that every leg sees byte-identical data.
"""
from __future__ import annotations

import math

import numpy as np


def rotation_from_rotvec(rv) -> np.ndarray:
    """3x3 rotation (float64) from a rotation vector (Rodrigues)."""
    rv = np.asarray(rv, dtype=np.float64)
    th = float(np.linalg.norm(rv))
    if th == 0.0:
        return np.identity(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.identity(3) + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K)


def rigid4(R=None, t=None, dtype=np.float32) -> np.ndarray:
    M = np.identity(4, dtype=np.float64)
    if R is not None:
        M[:3, :3] = R
    if t is not None:
        M[:3, 3] = t
    return M.astype(dtype)


def rot_z(deg: float) -> np.ndarray:
    a = math.radians(deg)
    return np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], dtype=np.float64)


def icosphere(subdivisions: int = 4, radius: float = 1.0) -> np.ndarray:
    """Vertices of a subdivided icosahedron (10*4**s + 2 points; s=4 -> 2562), float32."""
    return icosphere_mesh(subdivisions, radius)[0]


def icosphere_mesh(subdivisions: int = 4, radius: float = 1.0):
    """(vertices float32 [n,3], triangles int32 [20*4**s, 3]) of a subdivided icosahedron."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(v, dtype=np.float64) / math.sqrt(1 + t * t) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
             (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache = {}
        new_faces = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                m /= np.linalg.norm(m)
                cache[key] = len(verts)
                verts.append(m)
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            new_faces += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = new_faces
    return (np.array(verts) * radius).astype(np.float32), np.array(faces, dtype=np.int32)


def bumpy_icosphere_mesh(subdivisions: int = 4):
    """bumpy_icosphere() with its triangles."""
    v, f = icosphere_mesh(subdivisions)
    v = v.astype(np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (v * _bunny_radius(v)[:, None]).astype(np.float32), f


def lattice_surface_mesh(nu: int, nv: int):
    """Triangulated (nu x nv) lat-long grid of the bunny surface: (vertices [nu*nv,3], triangles [2(nu-1)(nv-1),3])."""
    th = np.linspace(0.02, math.pi - 0.02, nu)
    ph = np.linspace(0.0, 2.0 * math.pi, nv, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing="ij")
    u = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], axis=-1).reshape(-1, 3)
    verts = (u * _bunny_radius(u)[:, None]).astype(np.float32)
    i, j = np.meshgrid(np.arange(nu - 1), np.arange(nv), indexing="ij")
    a = (i * nv + j).ravel()
    b = (i * nv + (j + 1) % nv).ravel()
    c = ((i + 1) * nv + j).ravel()
    d = ((i + 1) * nv + (j + 1) % nv).ravel()
    tris = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)]).astype(np.int32)   # outward winding
    return verts, tris


def cubed_surface_mesh(n: int):
    """The bunny surface over a cube-sphere: six n x n patches of two triangles per quad, no poles -- triangles of nearly uniform
    size and aspect (the lat-long lattice_surface_mesh has fans of thin triangles at both poles).  Patches do not share vertex
    indices along the cube's edges (the seams coincide geometrically).  (6 (n+1)^2 vertices, 12 n^2 triangles)."""
    g = np.tan(np.linspace(-math.pi / 4, math.pi / 4, n + 1))          # equal-angle spacing
    A, B = np.meshgrid(g, g, indexing="ij")
    one = np.ones_like(A)
    faces = [(one, A, B), (-one, B, A), (B, one, A), (A, -one, B), (A, B, one), (B, A, -one)]
    verts, tris = [], []
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).ravel(); b = a + 1; c = a + (n + 1); d = c + 1
    quad = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)])
    for k, (x, y, z) in enumerate(faces):
        u = np.stack([x.ravel(), y.ravel(), z.ravel()], 1)
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        verts.append(u * _bunny_radius(u)[:, None])
        tris.append(quad + k * (n + 1) ** 2)
    return np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.int32)


def _bunny_radius(u: np.ndarray) -> np.ndarray:
    theta = np.arccos(np.clip(u[:, 2], -1.0, 1.0))
    phi = np.arctan2(u[:, 1], u[:, 0])
    return 1.0 + 0.30 * np.sin(3 * theta) * np.cos(2 * phi) + 0.15 * np.cos(5 * theta + 1.0) + 0.10 * np.sin(4 * phi)


def fibonacci_dirs(n: int, offset: float = 0.0) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    z = 1.0 - 2.0 * (k + 0.5 + 0.5 * offset) / (n + 0.5)
    z = np.clip(z, -1.0, 1.0)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    ga = math.pi * (3.0 - math.sqrt(5.0))
    ph = ga * (k + offset)
    return np.stack([r * np.cos(ph), r * np.sin(ph), z], axis=1)


def bunny_surface(n: int, offset: float = 0.0) -> np.ndarray:
    """'Synthetic bunny' (SURVEY.md 8d C2): a bumpy star-shaped surface sampled on a Fibonacci lattice."""
    u = fibonacci_dirs(n, offset)
    return (u * _bunny_radius(u)[:, None]).astype(np.float32)


def bunny_surface_with_normals(n: int, offset: float = 0.0):
    """bunny_surface() plus outward unit normals (central differences of the (theta, phi) parametrisation)."""
    u = fibonacci_dirs(n, offset)
    theta = np.arccos(np.clip(u[:, 2], -1.0, 1.0))
    phi = np.arctan2(u[:, 1], u[:, 0])

    def P(t, p):
        d = np.stack([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)], axis=1)
        return d * _bunny_radius(d)[:, None]

    e = 1e-5
    pt = (P(theta + e, phi) - P(theta - e, phi)) / (2 * e)
    pp = (P(theta, phi + e) - P(theta, phi - e)) / (2 * e)
    nrm = np.cross(pt, pp)
    nrm /= np.maximum(1e-30, np.linalg.norm(nrm, axis=1, keepdims=True))
    nrm *= np.sign(np.sum(nrm * u, axis=1, keepdims=True) + 1e-30)
    return (u * _bunny_radius(u)[:, None]).astype(np.float32), nrm.astype(np.float32)


def bumpy_icosphere(subdivisions: int = 4) -> np.ndarray:
    """Icosphere with the bunny radius modulation: breaks the icosahedral symmetry."""
    v = icosphere(subdivisions).astype(np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (v * _bunny_radius(v)[:, None]).astype(np.float32)


def c1_icospheres():
    """Config 1: two 2562-vertex icospheres, the align one rotated 15 degrees about z."""
    tgt = icosphere(4)
    src = tgt.copy()
    mx_align = rigid4(rot_z(15.0))
    mx_base = np.identity(4, dtype=np.float32)
    return src, tgt, mx_align, mx_base


def c2_bunny_pair(n: int = 100_000, seed: int = 100):
    """Config 2: n<->n synthetic bunny pair; source is an independent lattice moved by a small rigid motion."""
    del seed  # lattice is deterministic; kept for signature symmetry
    tgt = bunny_surface(n, 0.0)
    srcw = bunny_surface(n, 0.5).astype(np.float64)
    R = rotation_from_rotvec([0.10, -0.07, 0.12])
    t = np.array([0.05, -0.03, 0.02])
    src = ((srcw - t) @ R).astype(np.float32)            # R^-1 (x - t), row-vector form
    return src, tgt, np.identity(4, dtype=np.float32), np.identity(4, dtype=np.float32)


def c3_random_pair(n: int = 1_000_000, seed: int = 1234, sigma: float | None = None, n_target: int | None = None):
    """Config 3/4: n<->n uniform clouds in [-1,1]^3, source = R^-1(Q_perm + noise - t).

    sigma defaults to 5% of the mean point spacing (V/N)^(1/3).
    """
    rng = np.random.default_rng(seed)
    nt = n if n_target is None else n_target
    tgt = rng.uniform(-1.0, 1.0, size=(nt, 3)).astype(np.float32)
    spacing = (8.0 / nt) ** (1.0 / 3.0)
    if sigma is None:
        sigma = 0.05 * spacing
    perm = rng.permutation(nt)
    if n <= nt:
        base = tgt[perm[:n]].astype(np.float64)
    else:
        reps = -(-n // nt)
        base = np.concatenate([tgt[rng.permutation(nt)] for _ in range(reps)])[:n].astype(np.float64)
    noise = rng.normal(0.0, sigma, size=(n, 3))
    R = rotation_from_rotvec([0.010, -0.020, 0.015])
    t = np.array([0.004, -0.003, 0.002])
    src = ((base + noise - t) @ R).astype(np.float32)
    return src, tgt, np.identity(4, dtype=np.float32), np.identity(4, dtype=np.float32)
