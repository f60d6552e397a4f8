#!/usr/bin/env python3
"""GPU box: how many cell-list records must a grid search of the triangle surface look at, per query?

VERDICT r2 item 3: k_tri_search_grid scans ~165 records per query in the first iteration of the bench's surface case
(1M points, 1.96M triangles) for 12.5 closest-point evaluations.  This script derives the floor for that number on this
mesh and this grid: a search that answers with the TRUE nearest triangle at distance d has to look at every record of every
cell that intersects the open ball B(q, d) -- a triangle listed there could be closer, and nothing cheaper than its record
says it is not.  With an oracle for d (the device's own exact answer) that count is a lower bound for ANY scan order, seed
or ring schedule over the same cell lists; what the kernel scans on top of it is what seeds that are not the answer
(reach > d) and whole-cell granularity of rows cost.

The grid is rebuilt here the way build_tri_grid() builds it (cell edge = OA_TRI_CELL x mean triangle bounding-box diagonal)
and checked against the library's own cell and entry counts (oa_get_stat).
Usage: OA_GRID_STATS=1 python tools/tri_lower_bound.py [sample]      (OA_GRID_STATS=1 also prints the kernel's own totals)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine

sample = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tgt, tris = synth.lattice_surface_mesh(700, 1400)
src = synth.bunny_surface(1_000_000, offset=0.37)
mxa = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
eye = np.identity(4, dtype=np.float32)

# ---- the grid, as build_tri_grid() sizes it
P = tgt.astype(np.float64)[tris]                                   # (T, 3 corners, 3)
tlo, thi = P.min(axis=1), P.max(axis=1)
diag = np.sqrt(((thi - tlo) ** 2).sum(axis=1))
lo, hi = tgt.min(axis=0).astype(np.float64), tgt.max(axis=0).astype(np.float64)
ext = hi - lo
h = float(os.environ.get("OA_TRI_CELL", 1.25)) * diag.sum() / len(tris)
h = max(h, ext.max() / 1023.0)
while True:
    n = np.minimum(np.maximum(np.floor(ext / h).astype(np.int64) + 1, 1), 1024)
    if n.prod() > (1 << 24):
        h *= 1.3
        continue
    break
inv_h = 1.0 / h
cell = lambda v: np.clip(np.floor((v - lo) * inv_h).astype(np.int64), 0, n - 1)
clo, chi = cell(tlo), cell(thi)
span = chi - clo + 1
counts = np.zeros(int(n.prod()), np.int64)
for dz in range(int(span[:, 2].max())):
    for dy in range(int(span[:, 1].max())):
        for dx in range(int(span[:, 0].max())):
            m = (span[:, 0] > dx) & (span[:, 1] > dy) & (span[:, 2] > dz)
            if m.any():
                c = clo[m] + np.array([dx, dy, dz])
                np.add.at(counts, (c[:, 2] * n[1] + c[:, 1]) * n[0] + c[:, 0], 1)
print("grid: cell edge %.6g, %d x %d x %d = %d cells, %d cell-list entries (%.2f per triangle), %.1f per occupied cell" % (
    h, n[0], n[1], n[2], n.prod(), counts.sum(), counts.sum() / len(tris), counts.sum() / max(1, (counts > 0).sum())))


def floor_for(q, d, label):
    """mean over the sample of: records in cells intersecting the open ball B(q, d)"""
    tot = np.zeros(len(q))
    cells = np.zeros(len(q))
    a, b = cell(q - d[:, None]), cell(q + d[:, None])
    w = int((b - a).max()) + 1
    for dz in range(w):
        for dy in range(w):
            for dx in range(w):
                c = a + np.array([dx, dy, dz])
                ok = (c <= b).all(axis=1)
                blo = lo + c * h
                gap = np.maximum(np.maximum(blo - q, q - (blo + h)), 0.0)
                ok &= (gap ** 2).sum(axis=1) < d ** 2
                idx = (c[:, 2] * n[1] + c[:, 1]) * n[0] + c[:, 0]
                tot += np.where(ok, counts[np.where(ok, idx, 0)], 0)
                cells += ok
    print("%-34s true distance %.3g cells (mean), cells met by the ball %.1f, RECORDS in them %.1f (median %.0f, 90th pct %.0f)" % (
        label, (d / h).mean(), cells.mean(), tot.mean(), np.median(tot), np.percentile(tot, 90)))
    return tot.mean()


rng = np.random.default_rng(1)
pick = rng.choice(len(src), size=sample, replace=False)
with IcpEngine(0) as e:
    e.set_target_mesh(tgt, tris)
    e.set_source(src, stride=1)
    assert int(e.stat("tri_grid_cells")) == int(n.prod()) and int(e.stat("tri_grid_entries")) == int(counts.sum()), \
        (e.stat("tri_grid_cells"), n.prod(), e.stat("tri_grid_entries"), counts.sum())
    pose = mxa
    for it in (0, 1, 2, 4, 29):
        e.set_matrices(mxa, eye)
        e.reset_seeds()
        if it:
            e.run(iters=it, thresh=0.05, early_exit=False)          # (with OA_GRID_STATS=1 the kernel prints its own totals per launch)
        pose = e.matrix_world()
        e.set_matrices(pose, eye)
        _, d2, _ = e.nn_search()
        q = (src[pick].astype(np.float64) @ pose[:3, :3].astype(np.float64).T) + pose[:3, 3].astype(np.float64)
        floor_for(q, np.sqrt(d2[pick].astype(np.float64)), "pose before iteration %d:" % it)
