"""The library's own LSD argsort (object_alignment_amd/csrc/oa_sort.hpp) behind the Morton sorts of the index builds: it must
return the permutation rocprim's stable sort returns -- then slot order, tree leaves and every sum downstream are bitwise the
same.  Checked directly (tools/sort_bench.exe: eleven sizes, keys with ties) and end to end (whole loops with the sort forced
on for every size, OA_SORT_LSD_MIN=1, against rocprim everywhere, OA_SORT_LSD=0; the switches are read once per process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import json, sys, hashlib
import numpy as np
sys.path.insert(0, %r)
from object_alignment_amd import synth
from object_alignment_amd.engine import IcpEngine
out = {}
def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
with IcpEngine(0) as e:
    for name, n in (("c1", 0), ("bunny_3001", 3001), ("bunny_70k", 70001), ("bunny_150k", 150000)):
        if n == 0:
            s, t, a, b = synth.c1_icospheres()
        else:
            s, t, a, b = synth.c2_bunny_pair(n)
        for mode in ("auto", "bvh"):
            e.set_search_mode(mode)
            e.set_target(t); e.set_source(s, stride=1); e.set_matrices(a, b)
            r = e.run(iters=6, thresh=0.5, target_d=0.01, use_target=True, early_exit=False)
            out[name + ":" + mode] = digest(r.step_M, r.step_K, r.matrix_world)
    tv, tt = synth.bumpy_icosphere_mesh(6)                                   # 82k triangles: the triangle tree's sort
    pts = synth.bunny_surface(90_000, offset=0.37)
    pose = synth.rigid4(synth.rotation_from_rotvec([0.02, -0.015, 0.025]), [0.01, -0.008, 0.012])
    e.set_search_mode("auto")
    e.set_target_mesh(tv, tt); e.set_source(pts, stride=1); e.set_matrices(pose, np.identity(4, dtype=np.float32))
    r = e.run(iters=4, thresh=0.05, target_d=0.01, use_target=True, early_exit=False)
    out["surface"] = digest(r.step_M, r.step_K, r.matrix_world)
print(json.dumps(out))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, "-c", _CHILD], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_loops_do_not_depend_on_which_sort_ordered_the_slots():
    """Source slots, vertex tree and triangle tree sorted by the LSD argsort at EVERY size (from 2562 keys: one tile and a
    bit) against rocprim at every size: per-iteration matrices, pair counts and the final float32 matrix bitwise equal."""
    ours = _run({"OA_SORT_LSD": "1", "OA_SORT_LSD_MIN": "1", "OA_SORT_LSD_MAX": str(1 << 28)})
    theirs = _run({"OA_SORT_LSD": "0"})
    default = _run({})
    assert ours.keys() == theirs.keys() and len(ours) == 9
    assert ours == theirs, {k: (ours[k][:12], theirs[k][:12]) for k in ours if ours[k] != theirs[k]}
    assert default == theirs
    # ... nor on how the builds' small results reach the host (kernels storing into mapped host memory / device buffer + copy)
    assert _run({"OA_MAPPED_RESULTS": "0"}) == theirs


def test_sort_bench_permutations():
    """tools/sort_bench.exe (built by __graft_entry__.build()): 1 ... 10M keys with ties -- the same permutation as rocprim's
    merge sort and Onesweep, keys ascending along it, inputs untouched."""
    exe = os.path.join(ROOT, "tools", "sort_bench.exe")
    if not os.path.exists(exe):
        pytest.skip("tools/sort_bench.exe not built")
    p = subprocess.run([exe], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if "same permutation" in ln]
    assert len(lines) >= 10, p.stdout[-2000:]
    for ln in lines:
        assert "NO" not in ln and "oa_sort yes" in ln and "untouched yes" in ln, ln
