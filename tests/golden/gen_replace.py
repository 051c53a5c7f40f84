"""
G10: golden vectors for `replace_interpolated_weights` (xugrid/regrid/unstructured.py:17-57).

Runs ONLY in the build container (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python -B /root/repo/tests/golden/gen_replace.py

unstructured.py itself cannot be imported (top-level `import xarray`), so the function is lifted out of the file
with `ast` at run time and executed as plain Python (`numba` -> xugrid.constants.NoOpNumba, as the reference
does itself when numba is absent).  Only DATA is written (inputs + the weights the reference leaves behind).

One subtlety: the function squares coordinate differences with `** 2`.  numba lowers that to a multiplication,
plain Python calls libm's pow (not always the correctly rounded square: 162 of 200 000 random doubles differ by one
ulp).  The golden coordinates therefore lie on a dyadic lattice (multiples of 2^-10 below 2^12): every difference
has at most 22 significant bits, its square is exact, and the reference's result is the same under numba and
without it.  Everything after the squares (sqrt, /, *, +) is correctly rounded IEEE arithmetic either way.

Cases
  a  "hot path": faces = Voronoi-like cells with trailing -1 fill, the last n_map vertices are substitutes whose
     map entries (q, r) are ordinary vertices, sometimes present in the cell, sometimes not; face_index -1 rows
  b  q or r repeated inside a face, q == r, p itself twice in a face, zero and negative weights
  c  chains: map entries that are substitutes themselves (a later slot receives weight and is then processed
     in turn, an earlier slot keeps what it received) -- the sequential semantics of the loop
"""
import ast
import os
import sys
import types
import warnings

import numpy as np

warnings.simplefilter("ignore")
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

pkg = types.ModuleType("xugrid")
pkg.__path__ = [f"{REF}/xugrid"]
sys.modules["xugrid"] = pkg
from xugrid.constants import NoOpNumba as numba  # noqa: E402

src = open(f"{REF}/xugrid/regrid/unstructured.py").read()
fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "replace_interpolated_weights")
ns = {"numba": numba, "np": np}
exec(compile(ast.Module(body=[fn], type_ignores=[]), "replace_interpolated_weights", "exec"), ns)
reference = ns["replace_interpolated_weights"]


def dyadic(rng, n):
    return rng.integers(0, 1 << 22, size=(n, 2)).astype(np.float64) / 1024.0


def case(rng, n_vertex, n_map, n_face, m, n_rows, chains, repeats):
    vertices = dyadic(rng, n_vertex)
    threshold = n_vertex - n_map
    hi = n_vertex if chains else threshold
    node_map = rng.integers(0, hi, size=(n_map, 2)).astype(np.int64)
    if repeats:
        node_map[::5, 1] = node_map[::5, 0]  # q == r
    faces = np.full((n_face, m), -1, dtype=np.int64)
    for f in range(n_face):
        k = int(rng.integers(3, m + 1))
        ids = rng.integers(0, n_vertex, size=k)
        # make substitutes and their neighbours meet in the same cell often
        for j in range(k):
            if rng.random() < 0.35:
                ids[j] = rng.integers(threshold, n_vertex)
        for j in range(k):
            if ids[j] >= threshold and rng.random() < 0.7:
                q, r = node_map[ids[j] - threshold]
                ids[(j + 1) % k] = q
                if rng.random() < 0.6:
                    ids[(j - 1) % k] = r
        if repeats and k >= 5 and rng.random() < 0.5:
            ids[3] = ids[1]  # a node twice in a face (q / r / p repeated)
        faces[f, :k] = ids
    face_index = rng.integers(0, n_face, size=n_rows).astype(np.int64)
    weights = rng.random((n_rows, m))
    weights[faces[face_index] < 0] = 0.0
    weights[rng.random((n_rows, m)) < 0.15] = 0.0
    if repeats:
        weights[rng.random((n_rows, m)) < 0.05] *= -1.0
    outside = rng.random(n_rows) < 0.1
    face_index[outside] = -1  # point outside every cell: all-zero weights (faces[-1] is read, nothing happens)
    weights[outside] = 0.0
    out = weights.copy()
    reference(vertices, faces, face_index, out, node_map, threshold)
    return dict(vertices=vertices, faces=faces, face_index=face_index, weights_in=weights, node_to_node_map=node_map,
                threshold=np.int64(threshold), weights_out=out)


def main():
    rng = np.random.default_rng(10)
    out = {}
    for tag, kw in {
        "a": dict(n_vertex=400, n_map=60, n_face=150, m=9, n_rows=3000, chains=False, repeats=False),
        "b": dict(n_vertex=120, n_map=40, n_face=80, m=7, n_rows=3000, chains=False, repeats=True),
        "c": dict(n_vertex=90, n_map=45, n_face=60, m=8, n_rows=3000, chains=True, repeats=True),
    }.items():
        for k, v in case(rng, **kw).items():
            out[f"{tag}_{k}"] = v
        changed = int((out[f"{tag}_weights_out"] != out[f"{tag}_weights_in"]).sum())
        print(tag, "entries changed by the reference:", changed)
    np.savez_compressed(f"{OUT}/g10_replace.npz", **out)


if __name__ == "__main__":
    main()
