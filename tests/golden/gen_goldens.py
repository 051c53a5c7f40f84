"""
Generate the golden input/output vectors that pin oracle/ against the reference.

Runs ONLY in the build container (needs /root/reference, which does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python -B /root/repo/tests/golden/gen_goldens.py

It imports the importable parts of xugrid 0.15.3 from /root/reference (pure-Python fallback,
numba absent -> xugrid.constants.NoOpNumba) and lifts `make_regrid` and
`CentroidLocatorRegridder._regrid` out of regridder.py with `ast` at run time (the module
itself needs xarray).  Only DATA (inputs + the reference's outputs) is written to
tests/golden/*.npz; no reference source is stored in this repository.

Sets:
  G1 reducers   -- every method of reduce.ABSOLUTE/RELATIVE_OVERLAP_METHODS on crafted + random rows
  G2 apply      -- make_regrid(f)._regrid on a random CSR, K=3, f32 and f64 sources
  G3 csr        -- MatrixCSR.from_triplet / to_coo
  G4 rectilinear overlap -- overlap_1d x utils.broadcast (the quad x quad clip must reproduce it)
  G5 area / centroids on triangle, quad and mixed meshes
  G6 voronoi_topology full outputs on seeded Delaunay meshes
  G8 elevation_nl arrays (separate script: gen_elevation_nl.py, needs h5py from /opt/conda)
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
from xugrid.core.sparse import MatrixCOO, MatrixCSR, row_slice  # noqa: E402
from xugrid.regrid import overlap_1d, reduce, utils  # noqa: E402
from xugrid.ugrid import connectivity, voronoi  # noqa: E402

src = open(f"{REF}/xugrid/regrid/regridder.py").read()
tree = ast.parse(src)
fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "make_regrid")
ns = {"numba": numba, "np": np, "row_slice": row_slice, "FloatArray": np.ndarray, "MatrixCSR": MatrixCSR}
exec(compile(ast.Module(body=[fn], type_ignores=[]), "make_regrid", "exec"), ns)
make_regrid = ns["make_regrid"]
cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CentroidLocatorRegridder")
coo_fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_regrid")
coo_fn.decorator_list = []
ns2 = {"numba": numba, "np": np, "FloatArray": np.ndarray, "MatrixCOO": MatrixCOO}
exec(compile(ast.Module(body=[coo_fn], type_ignores=[]), "coo_regrid", "exec"), ns2)
coo_regrid = ns2["_regrid"]

METHODS = dict(reduce.ABSOLUTE_OVERLAP_METHODS)
METHODS.update(reduce.RELATIVE_OVERLAP_METHODS)
METHODS["p33.3"] = reduce.create_percentile_method(33.3)
METHODS["p0"] = reduce.create_percentile_method(0)
METHODS["p100"] = reduce.create_percentile_method(100)


def ragged(rows):
    off = np.zeros(len(rows) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r) for r in rows])
    flat = np.concatenate([np.asarray(r, dtype=np.float64) for r in rows]) if rows else np.zeros(0)
    return flat, off


def g1_reducers():
    rng = np.random.default_rng(101)
    vals, wts = [], []

    def add(v, w):
        vals.append(np.asarray(v, dtype=np.float64))
        wts.append(np.asarray(w, dtype=np.float64))

    # crafted (same spirit as tests/test_regrid/test_reduce.py)
    add([0, 1, 2, np.nan], [0.5] * 4)
    add([np.nan, 2, 1, 0], [0.5] * 4)
    add([np.nan] * 4, [0.5] * 4)
    add([0, 1, 2, np.nan], [0.0] * 4)
    add([1, 1, 3, 3], [1.0, 1.0, 1.0, 1.0])  # mode tie -> larger value
    add([1, 2, 3], [0.5, 2.0, 0.5])
    add([3, 2, 1], [2.0, 2.0, 2.0])  # max_overlap tie
    add([-1, 2, 3], [1, 1, 1])  # geometric mean of negatives
    add([0, 2, 8], [1, 1, 1])
    add([5.0], [2.0])
    add([np.nan, 7.0], [3.0, 0.0])
    add([np.nan, 7.0, 7.0, 2.0, 2.0, 2.0], [3.0, 1.0, 1.0, 0.5, 0.5, 0.5])
    add([2.0, 4.0], [0.0, 1.0])
    add([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], [1] * 10)
    add([10, 9, 8, 7, 6, 5, 4, 3, 2, 1], np.linspace(0.1, 1, 10))
    # random rows, length 1..64, 20 % NaN, integer-valued ties in half of them
    for i in range(300):
        n = int(rng.integers(1, 65))
        if i % 2:
            v = rng.integers(0, 6, n).astype(np.float64)
        else:
            v = rng.normal(size=n) * 10.0 ** rng.integers(-3, 4)
        if i % 3 == 0:
            v = np.abs(v)
        v[rng.random(n) < 0.2] = np.nan
        w = rng.random(n)
        if i % 7 == 0:
            w[rng.random(n) < 0.3] = 0.0
        add(v, w)
    vflat, off = ragged(vals)
    wflat, _ = ragged(wts)
    out = {"values": vflat, "weights": wflat, "offsets": off}
    for name, f in METHODS.items():
        res = np.empty(len(vals))
        for i, (v, w) in enumerate(zip(vals, wts)):
            wcopy = w.copy()
            res[i] = f(v.copy(), wcopy, np.empty(len(v)))
            assert np.array_equal(wcopy, w)
        out["out_" + name] = res
    np.savez_compressed(f"{OUT}/g1_reducers.npz", **out)
    print("G1", len(vals), "rows x", len(METHODS), "methods")


def g2_apply():
    rng = np.random.default_rng(202)
    T, S, K = 2000, 1500, 3
    lens = rng.integers(0, 41, T)
    lens[rng.random(T) < 0.05] = 0
    indptr = np.zeros(T + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(lens)
    nnz = int(indptr[-1])
    indices = rng.integers(0, S, nnz).astype(np.int64)
    data = rng.random(nnz)
    data[rng.random(nnz) < 0.02] = 0.0
    A = MatrixCSR(data, indices, indptr, T, S, nnz)
    src64 = rng.normal(size=(K, S)) * 5
    src64[0] = np.abs(src64[0])  # geometric/harmonic get a positive layer
    src64[1] = np.round(src64[1])  # ties for mode
    src64[rng.random((K, S)) < 0.1] = np.nan
    src32 = src64.astype(np.float32)
    out = {"data": data, "indices": indices, "indptr": indptr, "T": T, "S": S, "src64": src64, "src32": src32}
    for name, f in METHODS.items():
        rg = make_regrid(f)
        out["out64_" + name] = rg(src64, A, T)
        out["out32_" + name] = rg(src32, A, T)
    # COO scatter (CentroidLocatorRegridder._regrid)
    rows = np.sort(rng.choice(T, 900, replace=False)).astype(np.int64)
    cols = rng.integers(0, S, rows.size).astype(np.int64)
    C = MatrixCOO(np.ones(rows.size), rows, cols, T, S, rows.size)
    out["coo_row"], out["coo_col"] = rows, cols
    out["coo_out64"] = coo_regrid(src64, C, T)
    np.savez_compressed(f"{OUT}/g2_apply.npz", **out)
    print("G2 nnz", nnz)


def g3_csr():
    rng = np.random.default_rng(303)
    n, m = 50, 40
    row = np.sort(rng.integers(0, n, 300)).astype(np.int64)
    col = rng.integers(0, m, 300).astype(np.int64)
    dat = rng.random(300)
    A = MatrixCSR.from_triplet(row, col, dat, n=n, m=m)
    B = A.to_coo()
    # the known answer of tests/test_sparse.py: 5 rows x 2 entries -> [0,2,4,6,8,10]
    row2 = np.repeat(np.arange(5), 2)
    A2 = MatrixCSR.from_triplet(row2, np.arange(10) % 3, np.ones(10))
    np.savez_compressed(
        f"{OUT}/g3_csr.npz", row=row, col=col, data=dat, n=n, m=m, indptr=A.indptr, indices=A.indices,
        coo_row=B.row, coo_col=B.col, small_row=row2, small_indptr=A2.indptr, small_n=A2.n, small_m=A2.m,
    )
    print("G3")


def rectilinear(xb_s, yb_s, xb_t, yb_t):
    """Triplets of structured overlap: overlap_1d per axis, broadcast to 2D linear indices.
    bounds arrays are (n, 2) ascending.  Face id = row-major (y, x)."""
    sy, ty, wy = overlap_1d.overlap_1d(yb_s, yb_t)
    sx, tx, wx = overlap_1d.overlap_1d(xb_s, xb_t)
    s, t, w = utils.broadcast(
        (len(yb_s), len(xb_s)), (len(yb_t), len(xb_t)), (sy, sx), (ty, tx), (wy, wx)
    )
    order = np.lexsort((s, t))
    return s[order], t[order], w[order]


def bounds_from_edges(e):
    return np.column_stack((e[:-1], e[1:]))


def g4_rectilinear():
    out = {}
    # 3x3 (50 m) -> 4x4 (50 m, shifted 25 m): 36 pairs of 625 m2 (tests/test_regrid/test_structured.py:108-204)
    ea = np.array([25.0, 75.0, 125.0, 175.0])
    eb = np.array([0.0, 50.0, 100.0, 150.0, 200.0])
    s, t, w = rectilinear(bounds_from_edges(ea), bounds_from_edges(ea), bounds_from_edges(eb), bounds_from_edges(eb))
    out.update(a_xe_s=ea, a_ye_s=ea, a_xe_t=eb, a_ye_t=eb, a_src=s, a_tgt=t, a_w=w)
    # random non-equidistant 40x30 -> 25x35, partially overlapping extents
    rng = np.random.default_rng(404)
    xs = np.cumsum(rng.uniform(0.5, 2.0, 41)) + 3.0
    ys = np.cumsum(rng.uniform(0.5, 2.0, 31)) - 7.0
    xt = np.cumsum(rng.uniform(0.8, 3.0, 26)) + 1.0
    yt = np.cumsum(rng.uniform(0.3, 1.8, 36)) - 9.0
    s, t, w = rectilinear(bounds_from_edges(xs), bounds_from_edges(ys), bounds_from_edges(xt), bounds_from_edges(yt))
    out.update(b_xe_s=xs, b_ye_s=ys, b_xe_t=xt, b_ye_t=yt, b_src=s, b_tgt=t, b_w=w)
    # large coordinates (RD-like, 1e5..6e5) to exercise the relative-precision requirement
    xs = 100000.0 + np.cumsum(rng.uniform(50.0, 200.0, 33))
    ys = 450000.0 + np.cumsum(rng.uniform(50.0, 200.0, 29))
    xt = 100500.0 + np.cumsum(rng.uniform(30.0, 400.0, 21))
    yt = 450300.0 + np.cumsum(rng.uniform(30.0, 400.0, 18))
    s, t, w = rectilinear(bounds_from_edges(xs), bounds_from_edges(ys), bounds_from_edges(xt), bounds_from_edges(yt))
    out.update(c_xe_s=xs, c_ye_s=ys, c_xe_t=xt, c_ye_t=yt, c_src=s, c_tgt=t, c_w=w)
    np.savez_compressed(f"{OUT}/g4_rectilinear.npz", **out)
    print("G4", [out[k].size for k in ("a_w", "b_w", "c_w")])


def random_mesh(rng, n_pts, mixed):
    from scipy.spatial import Delaunay

    pts = rng.random((n_pts, 2)) * [7.0, 3.0] + [120000.0, 480000.0]
    tri = Delaunay(pts).simplices.astype(np.int64)
    if not mixed:
        return pts, tri
    faces = np.full((tri.shape[0], 5), -1, dtype=np.int64)
    faces[:, :3] = tri
    return pts, faces


def g5_geometry():
    rng = np.random.default_rng(505)
    out = {}
    pts, tri = random_mesh(rng, 200, False)
    out.update(tri_xy=pts, tri_faces=tri, tri_area=connectivity.area(tri, pts[:, 0], pts[:, 1]),
               tri_centroids=connectivity.centroids(tri, pts[:, 0], pts[:, 1]))
    # quads: a sheared lattice
    nx, ny = 9, 7
    gx, gy = np.meshgrid(np.arange(nx + 1.0), np.arange(ny + 1.0))
    qx = (gx + 0.3 * gy + 0.05 * rng.normal(size=gx.shape)).ravel()
    qy = (gy + 0.1 * gx + 0.05 * rng.normal(size=gx.shape)).ravel()
    idx = np.arange((nx + 1) * (ny + 1)).reshape(ny + 1, nx + 1)
    quads = np.column_stack([idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()]).astype(np.int64)
    qxy = np.column_stack([qx, qy])
    out.update(quad_xy=qxy, quad_faces=quads, quad_area=connectivity.area(quads, qx, qy),
               quad_centroids=connectivity.centroids(quads, qx, qy))
    # mixed: triangles padded with -1 next to quads and a pentagon
    mxy = np.array([[0, 0], [1, 0], [2, 0], [0, 1], [1, 1], [2, 1], [0, 2], [1, 2], [2.5, 1.6], [1.5, 2.6]], dtype=float) * 100 + 5000
    mf = np.array([[0, 1, 4, 3, -1], [1, 2, 5, 4, -1], [3, 4, 7, -1, -1], [3, 7, 6, -1, -1], [4, 5, 8, 9, 7]], dtype=np.int64)
    out.update(mix_xy=mxy, mix_faces=mf, mix_area=connectivity.area(mf, mxy[:, 0], mxy[:, 1]),
               mix_centroids=connectivity.centroids(mf, mxy[:, 0], mxy[:, 1]))
    np.savez_compressed(f"{OUT}/g5_geometry.npz", **out)
    print("G5")


def voronoi_inputs(pts, faces):
    """The exact call made by UnstructuredGrid2d.barycentric (regrid/unstructured.py:151-165)."""
    n_node = pts.shape[0]
    nfc = connectivity.invert_dense_to_sparse(faces)
    if nfc.shape[0] < n_node:
        nfc.resize((n_node, nfc.shape[1]))
    enc, fec = connectivity.edge_connectivity(faces)
    efc = connectivity.invert_dense(fec)
    cen = connectivity.centroids(faces, pts[:, 0], pts[:, 1])
    return nfc, enc, efc, cen


def g6_voronoi():
    from scipy.spatial import Delaunay

    out = {}
    for tag, n, seed in (("a", 50, 1), ("b", 500, 2), ("c", 5000, 3)):
        rng = np.random.default_rng(600 + seed)
        pts = rng.random((n, 2)) * [10.0, 6.0] + [1000.0, 2000.0]
        faces = Delaunay(pts).simplices.astype(np.int64)
        # Delaunay of all points uses every node
        ccw = connectivity.counterclockwise(faces, pts)
        nfc, enc, efc, cen = voronoi_inputs(pts, ccw)
        v, f, fi, nmap = voronoi.voronoi_topology(
            nfc, pts, cen, edge_face_connectivity=efc, edge_node_connectivity=enc,
            add_exterior=True, add_vertices=True, skip_concave=True,
        )
        out.update({f"{tag}_xy": pts, f"{tag}_faces": ccw, f"{tag}_centroids": cen, f"{tag}_enc": enc, f"{tag}_efc": efc,
                    f"{tag}_vor_vertices": v, f"{tag}_vor_faces": f, f"{tag}_vor_face_i": fi, f"{tag}_vor_nmap": nmap})
        print("G6", tag, faces.shape, v.shape, f.shape, nmap.shape)
    np.savez_compressed(f"{OUT}/g6_voronoi.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6"]
    for w in which:
        {"g1": g1_reducers, "g2": g2_apply, "g3": g3_csr, "g4": g4_rectilinear, "g5": g5_geometry, "g6": g6_voronoi}[w]()
