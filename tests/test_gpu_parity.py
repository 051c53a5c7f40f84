"""
GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Bar: indices bit-exact, overlap areas bit-exact (identical
operation order, no FMA contraction on either side), regridded values bit-exact except
geometric_mean (device exp/log vs libm: rtol 1e-13; north_star tolerance is 1e-10).
"""
import numpy as np
import pytest

from conftest import same_or_nan
from xugrid_amd import meshgen

pytestmark = pytest.mark.gpu

RTOL_GEOMETRIC = 1e-13
LONG_ROW = 32  # rows with more entries are reduced cooperatively (wave / block, fixed tree order): ~1e-15 instead of bit-exact
RTOL_LONG = 1e-13


def assert_apply_equal(got, exp, indptr, name=""):
    """bit-exact for rows reduced sequentially, RTOL_LONG for block-reduced long rows."""
    long_rows = np.diff(indptr) > LONG_ROW
    short = ~long_rows
    assert same_or_nan(got[:, short], exp[:, short]).all(), (name, int((~same_or_nan(got[:, short], exp[:, short])).sum()))
    if long_rows.any():
        # harmonic means of values of both signs cancel catastrophically (sum w / v near zero): there a different
        # summation order shows up to 1e-10 (the north star's tolerance) instead of a few ulp
        rtol = 1e-9 if name == "harmonic_mean" else RTOL_LONG
        # (sums of values of both signs may cancel: absolute floor of a few ulp of the largest result)
        finite = exp[:, long_rows][np.isfinite(exp[:, long_rows])]
        atol = 1e-13 * (np.abs(finite).max() if finite.size else 1.0)
        np.testing.assert_allclose(got[:, long_rows], exp[:, long_rows], rtol=rtol, atol=atol, equal_nan=True, err_msg=name)


def gpu_triplets(hip, sxy, sf, txy, tf, relative=False, fill=-1):
    E = hip.engine
    ms, mt = E.DeviceMesh(sxy, sf, fill), E.DeviceMesh(txy, tf, fill)
    csr = ms.overlap(mt, relative)
    data, idx, indptr = csr.download()
    q = np.repeat(np.arange(csr.n), np.diff(indptr))
    return csr, q, idx, data, indptr


def assert_overlap_parity(hip, oracle, sxy, sf, txy, tf, relative=False, fill=-1):
    tree = oracle.CellTree2d(sxy, sf, fill)
    oq, os_, oa = tree.intersect_faces(txy, tf, fill)
    if relative:
        sf_norm = np.where(np.asarray(sf) == fill, -1, sf)
        oa = oa / oracle.area(sxy, sf_norm)[os_]
    csr, q, idx, data, indptr = gpu_triplets(hip, sxy, sf, txy, tf, relative, fill)
    assert csr.n == np.asarray(tf).shape[0] and csr.m == np.asarray(sf).shape[0]
    assert np.array_equal(q, oq), "target indices differ"
    assert np.array_equal(idx, os_), "source indices differ"
    assert np.array_equal(data, oa), "areas not bit-exact (max rel %.3g)" % (np.abs(data - oa) / oa).max()
    # rows sorted by source index
    for t in np.nonzero(np.diff(indptr) > 1)[0][:2000]:
        row = idx[indptr[t]:indptr[t + 1]]
        assert (np.diff(row) > 0).all()
    return csr, (data, idx, indptr)


def test_overlap_triangles_delaunay(hip, oracle):
    sxy, sf = meshgen.triangle_mesh(3000, 0)
    txy, tf = meshgen.triangle_mesh(3000, 1, 30.0, 0.7)
    assert_overlap_parity(hip, oracle, sxy, sf, txy, tf)
    assert_overlap_parity(hip, oracle, sxy, sf, txy, tf, relative=True)
    assert_overlap_parity(hip, oracle, txy, tf, sxy, sf)  # fine target partly outside the source hull


def test_overlap_random_face_numbering(hip, oracle):
    """Shuffled face numbering on both sides: the engine sorts the target into its spatial query
    order internally (coherent numberings skip that step) -- results are in the caller's numbering."""
    rng = np.random.default_rng(42)
    sxy, sf = meshgen.triangle_mesh(3000, 0)
    txy, tf = meshgen.triangle_mesh(3000, 1, 30.0, 0.7)
    sf, tf = sf[rng.permutation(sf.shape[0])], tf[rng.permutation(tf.shape[0])]
    csr, (data, idx, indptr) = assert_overlap_parity(hip, oracle, sxy, sf, txy, tf)
    v = meshgen.smooth_field(oracle.centroids(sxy, sf), 3, nan_fraction=0.03)[None, :]
    assert_apply_equal(csr.apply(v, 0), oracle.regrid_csr("mean", v, data, idx, indptr, csr.n), indptr)
    assert_overlap_parity(hip, oracle, sxy, sf, txy, tf, relative=True)


def test_overlap_clockwise_and_fill_values(hip, oracle):
    sxy, sf = meshgen.triangle_mesh(800, 2)
    txy, tf = meshgen.quad_mesh(np.linspace(0.05, 0.95, 23), np.linspace(0.1, 0.9, 17))
    sf_cw = sf[:, ::-1].copy()
    assert_overlap_parity(hip, oracle, sxy, sf_cw, txy, tf)
    # triangles padded to 5 columns with a non-default fill value, quads padded likewise
    sf5 = np.full((sf.shape[0], 5), -999, dtype=np.int64)
    sf5[:, :3] = sf
    tf5 = np.full((tf.shape[0], 5), -999, dtype=np.int64)
    tf5[:, :4] = tf[:, ::-1]
    assert_overlap_parity(hip, oracle, sxy, sf5, txy, tf5, fill=-999)
    assert_overlap_parity(hip, oracle, sxy, sf5, txy, tf5, relative=True, fill=-999)


def test_overlap_quadrilateral_targets_on_triangles(hip, oracle, monkeypatch, xr_option):
    """The shape of the reference's unstructured -> raster regridding: quadrilateral targets against a triangle source go
    through the flag / compaction clip with a four-vertex subject (k_clip_quad_tri).  A raster whose cell lines pass through
    source vertices (a lattice-split triangulation: exact contacts), a rotated and sheared quad mesh, clockwise quads,
    and a target that mixes quads with triangles (fill slot); bit-equal to the oracle and to the slot-loop kernel
    (XR_CLIP_QUAD=0), absolute and relative."""
    sxy, sf = meshgen.triangle_mesh(2500, 5, delaunay=False)
    lo, hi = sxy.min(), sxy.max()
    rxy, rf = meshgen.quad_mesh(np.linspace(lo, hi, 41), np.linspace(lo, hi, 33))
    ang = np.deg2rad(17.0)
    shear = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang) + 0.3, np.cos(ang)]])
    qxy = (rxy - rxy.mean(0)) @ shear.T * 0.8 + rxy.mean(0)
    mixed = rf.copy()
    mixed[::3, 3] = -1  # every third cell: the triangle of its first three corners
    dxy, df = meshgen.triangle_mesh(2500, 6)  # a Delaunay source for the general-position cases
    cases = [(sxy, sf, rxy, rf), (dxy, df, qxy, rf), (dxy, df, rxy, rf[:, ::-1].copy()), (sxy, sf, rxy, mixed), (dxy, df, qxy, mixed)]
    for src_xy, src_f, txy, tf in cases:
        results = []
        for flag in ("1", "0"):
            xr_option("clip_quad", flag)
            csr, _ = assert_overlap_parity(hip, oracle, src_xy, src_f, txy, tf)
            results.append(csr.download())
            assert_overlap_parity(hip, oracle, src_xy, src_f, txy, tf, relative=True)
        assert all(np.array_equal(a, b) for a, b in zip(*results))
    xr_option("clip_quad", None)


def test_overlap_polygons_up_to_hexagons(hip, oracle):
    """mixed 3..6-gons (voronoi cells of a Delaunay mesh) against triangles: MAXV = 16 clip kernel."""
    from xugrid_amd import connectivity as C, voronoi

    xy, faces = meshgen.triangle_mesh(400, 5)
    cen = xy[faces].mean(axis=1)
    nfc = C.invert_dense_to_sparse(faces, n_rows=len(xy))
    v, cells, _, _ = voronoi.voronoi_topology(nfc, xy, cen)  # interior cells only, ragged
    assert cells.shape[1] > 4
    txy, tf = meshgen.triangle_mesh(500, 6, 10.0, 0.8)
    assert_overlap_parity(hip, oracle, v, cells, txy, tf)
    assert_overlap_parity(hip, oracle, txy, tf, v, cells)


def test_polygon_meshes_flat_vertex_blocks(hip, oracle):
    """Meshes with more than 4 nodes per face keep their vertex blocks FLAT with offsets (xr_geom.h): one 32-gon among
    60k triangles must not cost n_face * 32 * 16 bytes three times over.  Overlap (both roles, caller-coherent and shuffled
    numbering), locate and the network-edge lengths agree with the oracle bit for bit on such a mesh."""
    E = hip.engine
    xy, faces = meshgen.triangle_mesh(30_000, 21)
    F = faces.shape[0]
    # a regular 32-gon far outside the unit square, appended as one more face of a 32-wide connectivity table
    ang = 2 * np.pi * np.arange(32) / 32
    gon = np.column_stack([3.0 + 0.5 * np.cos(ang), 0.5 + 0.5 * np.sin(ang)])
    pxy = np.vstack([xy, gon])
    pf = np.full((F + 1, 32), -1, dtype=np.int64)
    pf[:F, :3] = faces
    pf[F] = np.arange(32) + xy.shape[0]
    txy, tf = meshgen.triangle_mesh(20_000, 22, 15.0, 0.9)
    txy = np.vstack([txy, [[2.4, 0.0], [3.6, 0.1], [3.1, 1.1]]])  # + one triangle over the 32-gon
    tf = np.vstack([tf, [[txy.shape[0] - 3, txy.shape[0] - 2, txy.shape[0] - 1]]])

    assert_overlap_parity(hip, oracle, pxy, pf, txy, tf)          # polygon mesh as the tree
    csr, (data, idx, indptr) = assert_overlap_parity(hip, oracle, txy, tf, pxy, pf)   # ... and as the query
    assert indptr[F + 1] - indptr[F] >= 1                         # the 32-gon meets its triangle
    perm = np.random.default_rng(3).permutation(F + 1)            # incoherent numbering: Morton query order, flat too
    assert_overlap_parity(hip, oracle, txy, tf, pxy, pf[perm])
    assert_overlap_parity(hip, oracle, pxy, pf[perm], txy, tf, relative=True)

    mesh = E.DeviceMesh(pxy, pf)
    pts = np.vstack([np.random.default_rng(4).random((5000, 2)), gon.mean(axis=0)[None, :] + [[0.0, 0.0], [0.2, 0.1]]])
    tree = oracle.CellTree2d(pxy, pf)
    assert np.array_equal(mesh.locate_points(pts), tree.locate_points(pts))
    from test_gpu_network import device_vs_oracle

    edges = np.random.default_rng(5).random((3000, 2, 2)) * [3.8, 1.0]
    device_vs_oracle(oracle, pxy, pf, edges)

    # memory: raw table (4 B per slot) + flat blocks; dense blocks alone would be (F + 1) * 32 * 16 B per copy
    mesh.build_index()
    E.DeviceMesh(txy, tf).overlap(mesh)   # (also prepares `mesh` as a query: caller-order blocks)
    dense_copy = (F + 1) * 32 * 16
    assert mesh.device_bytes() < 0.6 * dense_copy, (mesh.device_bytes(), dense_copy)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_overlap_rectilinear_goldens(hip, golden, tag):
    """HIP clip of quads == the reference's separable structured overlap (golden G4)."""
    g = golden("g4_rectilinear.npz")
    sxy, sf = meshgen.quad_mesh(g[tag + "_xe_s"], g[tag + "_ye_s"])
    txy, tf = meshgen.quad_mesh(g[tag + "_xe_t"], g[tag + "_ye_t"])
    _, q, idx, data, _ = gpu_triplets(hip, sxy, sf, txy, tf)
    assert np.array_equal(q, g[tag + "_tgt"]) and np.array_equal(idx, g[tag + "_src"])
    np.testing.assert_allclose(data, g[tag + "_w"], rtol=1e-12)


def test_overlap_long_rows_and_big_queries(hip, oracle):
    """coarse target over a fine source (rows of thousands of entries -> LDS bitmap rank kernel,
    wave-per-face search) and the reverse (rows of 1-2 entries)."""
    sxy, sf = meshgen.triangle_mesh(40000, 3)
    txy, tf = meshgen.quad_mesh(np.linspace(-0.1, 1.1, 7), np.linspace(0.0, 1.0, 5))
    csr, (data, idx, indptr) = assert_overlap_parity(hip, oracle, sxy, sf, txy, tf)
    assert np.diff(indptr).max() > 2000
    assert_overlap_parity(hip, oracle, txy, tf, sxy, sf, relative=True)
    # rows of a few hundred to ~2000 candidates: block-per-row all-pairs rank kernel
    for nx, ny in ((21, 17), (12, 9)):
        txy, tf = meshgen.quad_mesh(np.linspace(-0.05, 1.05, nx), np.linspace(0.02, 0.97, ny))
        csr, (data, idx, indptr) = assert_overlap_parity(hip, oracle, sxy, sf, txy, tf, relative=(nx == 12))
        assert 128 < np.diff(indptr).max() < 2048


def test_overlap_queue_regrow_path(hip, oracle, monkeypatch, xr_option):
    """big query faces whose candidates do not fit the pair queue's margin: the queue is regrown and the
    pending faces are filled by the second launch (option "queue_margin" is a test hook)."""
    xr_option("queue_margin", 1)
    sxy, sf = meshgen.triangle_mesh(30000, 3)
    txy, tf = meshgen.quad_mesh(np.linspace(-0.1, 1.1, 6), np.linspace(0.0, 1.0, 4))
    csr, (data, idx, indptr) = assert_overlap_parity(hip, oracle, sxy, sf, txy, tf)
    assert np.diff(indptr).max() > 2000 and hip.engine.DeviceMesh  # rows far beyond 16 * T queue entries
    # mixed: a fine target with a few huge faces appended
    fxy, ff = meshgen.triangle_mesh(4000, 4, 10.0, 0.9)
    big = np.array([[0.02, 0.02], [0.98, 0.03], [0.97, 0.99], [0.03, 0.96]])
    mxy = np.vstack([fxy, big])
    n0 = fxy.shape[0]
    mf = np.vstack([ff, [[n0, n0 + 1, n0 + 2], [n0, n0 + 2, n0 + 3]]])
    assert_overlap_parity(hip, oracle, sxy, sf, mxy, mf)


def test_overlap_fused_matches_general_chain(hip, monkeypatch, xr_option):
    """triangle x triangle pairs take the single-round-trip pipeline (persistent clip, look-back assembly, big faces on a
    side stream); XR_OVERLAP_FUSED=0 sends the same meshes through the general chain (search -> clip -> scan -> row_fill):
    identical CSR, absolute and relative weights."""
    sxy, sf = meshgen.triangle_mesh(40000, 11)
    txy, tf = meshgen.triangle_mesh(30000, 12, 25.0, 0.8)
    results = {}
    for mode in ("1", "0"):
        xr_option("overlap_fused", mode)
        for relative in (False, True):
            results[mode, relative] = gpu_triplets(hip, sxy, sf, txy, tf, relative)[2:]
    for relative in (False, True):
        a, b = results["1", relative], results["0", relative]
        assert a[0].size > 100000
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_overlap_fused_pipeline_quads_and_mixed_meshes(hip, oracle, monkeypatch, xr_option):
    """Dense meshes of up to four nodes per face -- quadrilaterals, mixed triangle / quadrilateral meshes with -1 in the fourth slot
    (the flexible-mesh case), a raster's quads against triangles -- take the one-round-trip pipeline of xr_overlap_fused.h with
    the register / LDS clip of k_clip_small inside the persistent queue kernel (round 5).  Every pair: the oracle's matrix bit for
    bit, and the identical CSR through the general kernel chain (XR_OVERLAP_FUSED=0); big faces (coarse targets), relative
    weights, a mixed mesh against itself."""
    mxy, mf = meshgen.mixed_mesh(9000, 3)
    nxy, nf = meshgen.mixed_mesh(7000, 4, 25.0, 0.8)
    txy, tf = meshgen.triangle_mesh(6000, 5, 30.0, 0.7)
    qxy, qf = meshgen.quad_mesh(np.linspace(0.1, 0.9, 61), np.linspace(0.15, 0.85, 47))
    cxy, cf = meshgen.mixed_mesh(150, 6, 10.0, 0.9)  # coarse: every target is a big face on the fine source
    assert (mf[:, 3] >= 0).any() and (mf[:, 3] < 0).any()
    pairs = [(mxy, mf, nxy, nf), (nxy, nf, mxy, mf), (mxy, mf, txy, tf), (txy, tf, mxy, mf), (txy, tf, qxy, qf), (qxy, qf, nxy, nf),
             (mxy, mf, cxy, cf), (mxy, mf, mxy, mf)]
    for sxy, sf, qx, qfc in pairs:
        for relative in (False, True):
            xr_option("overlap_fused", None)
            _, fused = assert_overlap_parity(hip, oracle, sxy, sf, qx, qfc, relative=relative)
            xr_option("overlap_fused", "0")
            general = gpu_triplets(hip, sxy, sf, qx, qfc, relative)
            assert np.array_equal(fused[0], general[3]) and np.array_equal(fused[1], general[2]) and np.array_equal(fused[2], general[4])
    xr_option("overlap_fused", None)


def test_overlap_meshes_sharing_nodes(hip, oracle):
    """The reference clips a pair only if the two faces' EXACT bounding boxes overlap strictly (numba_celltree
    boxes_intersect, oracle/xr_oracle.c:530): faces that merely touch -- a mesh against itself: the neighbours across a
    corner -- never reach its clip, which would return a sliver of ~1e-36 for some of them (23 pairs of 60 394 on the mesh
    below; the engine's float boxes are supersets and used to let them through).  The search marks the pairs whose float boxes
    overlap by their rounding only, the clip kernels repeat the box test on the float64 vertices for those.  Also: the same
    mesh far from the origin (UTM-like coordinates), needle-thin and zero-area source triangles, big faces (side-stream chain)
    and the general kernel chain."""
    sxy, sf = meshgen.triangle_mesh(30000, 21)
    txy, tf = meshgen.triangle_mesh(25000, 22, 25.0, 0.8)
    assert_overlap_parity(hip, oracle, sxy, sf, sxy, sf)
    off = np.array([6.5e5, 5.9e6])
    assert_overlap_parity(hip, oracle, sxy * 3000.0 + off, sf, sxy * 3000.0 + off, sf)
    assert_overlap_parity(hip, oracle, sxy * 3000.0 + off, sf, txy * 3000.0 + off, tf)
    # a coarser mesh on a subset of the same nodes (every face of it a union of boundary-sharing fine faces is NOT
    # required: any triangulation of a node subset shares vertices and many edge directions with the fine mesh)
    from scipy.spatial import Delaunay

    sub = np.sort(np.random.default_rng(3).choice(sxy.shape[0], 4000, replace=False))
    cf = Delaunay(sxy[sub]).simplices.astype(np.int64)
    p = sxy[sub]
    u, v = p[cf[:, 1]] - p[cf[:, 0]], p[cf[:, 2]] - p[cf[:, 0]]
    cw = (u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) < 0
    cf[cw] = cf[cw][:, ::-1]
    assert_overlap_parity(hip, oracle, sxy, sf, p, cf)   # coarse targets: the big faces' chain
    assert_overlap_parity(hip, oracle, p, cf, sxy, sf)
    # degenerate sources: needles (height ~1e-16) and repeated nodes
    rng = np.random.default_rng(5)
    nxy = sxy.copy()
    bad = rng.choice(sf.shape[0], 400, replace=False)
    f2 = sf.copy()
    for k, f in enumerate(bad):
        a, b, c = f2[f]
        if k % 2 == 0:
            nxy = np.vstack([nxy, 0.5 * (nxy[a] + nxy[b]) + 1e-13 * (nxy[c] - nxy[a])])
            f2[f] = [a, b, nxy.shape[0] - 1]
        else:
            f2[f] = [a, b, b]
    assert_overlap_parity(hip, oracle, nxy, f2, txy, tf)
    assert_overlap_parity(hip, oracle, nxy, f2, sxy, sf)


def test_overlap_same_handle_as_tree_and_query(hip, oracle):
    """A grid regridded onto itself through the public API hands the SAME device mesh in as tree and as query (prepared light
    as a tree, then fully as a query, indexed and query-ordered on one handle): the oracle's matrix, and the mean of a field
    comes back unchanged up to the rounding of the few rows that hold more than their own face."""
    import xugrid_amd as xa

    sxy, sf = meshgen.triangle_mesh(20000, 5)
    ms = hip.engine.DeviceMesh(sxy, sf)
    data, idx, indptr = ms.overlap(ms).download()
    oq, os_, oa = oracle.CellTree2d(sxy, sf, -1).intersect_faces(sxy, sf, -1)
    assert np.array_equal(np.repeat(np.arange(indptr.size - 1), np.diff(indptr)), oq)
    assert np.array_equal(idx, os_) and np.array_equal(data, oa)
    g = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    v = np.random.default_rng(0).normal(size=sf.shape[0])
    np.testing.assert_allclose(xa.OverlapRegridder(g, g, method="mean").regrid(v), v, rtol=0, atol=1e-13)


def test_overlap_meshes_sharing_nodes_general_chain(hip, oracle, monkeypatch, xr_option):
    """... and through the general kernel chain (XR_OVERLAP_FUSED=0) and a quadrilateral mesh against itself."""
    xr_option("overlap_fused", "0")
    sxy, sf = meshgen.triangle_mesh(12000, 23)
    assert_overlap_parity(hip, oracle, sxy, sf, sxy, sf)
    xr_option("overlap_fused", None)
    qxy, qf = meshgen.quad_mesh(np.linspace(0.0, 1.0, 71), np.linspace(0.0, 1.0, 53))
    assert_overlap_parity(hip, oracle, qxy, qf, qxy, qf)
    assert_overlap_parity(hip, oracle, sxy, sf, qxy, qf)
    assert_overlap_parity(hip, oracle, qxy, qf, sxy, sf)


def test_overlap_graded_mesh_many_levels(hip, oracle):
    """face sizes spanning 4 orders of magnitude -> many grid levels."""
    rng = np.random.default_rng(9)
    r = 10.0 ** rng.uniform(-4, 0, 3000)
    th = rng.uniform(0, 2 * np.pi, 3000)
    pts = np.column_stack([r * np.cos(th), r * np.sin(th)])
    from scipy.spatial import Delaunay

    sf = Delaunay(pts).simplices.astype(np.int64)
    txy, tf = meshgen.triangle_mesh(1500, 4, 15.0, 1.6)
    txy = txy - 0.5
    assert_overlap_parity(hip, oracle, pts, sf, txy, tf)
    assert_overlap_parity(hip, oracle, txy, tf, pts, sf)


def test_overlap_sampled_tree_statistics(hip, oracle):
    """From 131072 source faces on, the tree side's grid is sized from a SAMPLE of the faces (every 8th block of 256;
    bounds exact from the nodes, number of levels from the domain).  A graded mesh whose sampled blocks misrepresent the
    mean -- fine faces first, coarse ones last, the few huge ones in blocks the sample skips -- must give the oracle's
    pairs and areas all the same (the grid is an accelerator only), and the default tolerance of a later locate on the
    same handle must still be the one of the exact statistics (largest bbox diagonal over ALL faces)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(19)
    n = 80_000
    r = 10.0 ** rng.uniform(-3, 0, n)
    th = rng.uniform(0, 2 * np.pi, n)
    pts = np.column_stack([r * np.cos(th), r * np.sin(th)])
    sf = Delaunay(pts).simplices.astype(np.int64)
    # faces ordered by size: the strided sample sees a biased mix, the largest faces sit in the last (partly unsampled) blocks
    p = pts[sf]
    ext = np.maximum(np.ptp(p[:, :, 0], axis=1), np.ptp(p[:, :, 1], axis=1))
    sf = sf[np.argsort(ext, kind="stable")]
    assert sf.shape[0] >= 131072
    txy, tf = meshgen.triangle_mesh(20_000, 4, 15.0, 1.6)
    txy = txy - 0.5
    csr, _ = assert_overlap_parity(hip, oracle, pts, sf, txy, tf)
    assert csr.nnz > tf.shape[0]
    # the same handle as a locate tree afterwards: tolerance from exact statistics, results as on a fresh handle
    E = hip.engine
    ms = E.DeviceMesh(pts, sf)
    ms.overlap(E.DeviceMesh(txy, tf))
    q = rng.uniform(-0.8, 0.8, (50_000, 2))
    q[:200] = pts[sf[:200, 0]]  # points ON vertices: decided by the tolerance
    assert np.array_equal(ms.locate_points(q), E.DeviceMesh(pts, sf).locate_points(q))
    assert np.array_equal(ms.locate_points(q), oracle.CellTree2d(pts, sf).locate_points(q))


def test_overlap_degenerate_inputs(hip, oracle):
    sxy, sf = meshgen.triangle_mesh(100, 1)
    # disjoint meshes -> empty matrix, apply gives NaN everywhere
    csr, q, idx, data, indptr = gpu_triplets(hip, sxy, sf, sxy + 10.0, sf)
    assert csr.nnz == 0 and np.array_equal(indptr, np.zeros(sf.shape[0] + 1, dtype=indptr.dtype))
    out = csr.apply(np.ones((2, sf.shape[0])))
    assert out.shape == (2, sf.shape[0]) and np.isnan(out).all()
    # single face vs single face, identical -> one pair with the face area
    one = np.array([[0, 1, 2]])
    tri = np.array([[0.0, 0.0], [2.0, 0.0], [0.0, 1.0]])
    csr, q, idx, data, _ = gpu_triplets(hip, tri, one, tri, one)
    assert csr.nnz == 1 and data[0] == 1.0
    # faces sharing only an edge / a vertex give no pair (area > 0 filter)
    tri2 = np.array([[2.0, 0.0], [2.0, 1.0], [0.0, 1.0]])
    csr, *_ = gpu_triplets(hip, tri, one, tri2, one)
    assert csr.nnz == 0
    # zero-area (collinear) face and repeated vertices
    xy = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [1.0, 1.0]])
    f = np.array([[0, 1, 2, -1], [0, 2, 3, -1], [0, 1, 1, 3]])
    assert_overlap_parity(hip, oracle, xy, f, tri, one[:, [0, 1, 2]])
    # empty query mesh
    E = hip.engine
    empty = E.DeviceMesh(sxy, np.zeros((0, 3), dtype=np.int64))
    csr = E.DeviceMesh(sxy, sf).overlap(empty)
    assert (csr.n, csr.m, csr.nnz) == (0, sf.shape[0], 0)
    csr = empty.overlap(E.DeviceMesh(sxy, sf))
    assert (csr.n, csr.m, csr.nnz) == (sf.shape[0], 0, 0)
    with pytest.raises(ValueError):
        E.DeviceMesh(sxy, np.array([[0, 1, 10**6]]))  # node index out of range
    with pytest.raises(ValueError):
        E.DeviceMesh(sxy, np.array([[0, 1, -1]]))  # fewer than 3 nodes
    with pytest.raises(OverflowError):
        E.DeviceMesh(sxy, np.zeros((1, 40), dtype=np.int64))  # more than 32 nodes per face


def test_mesh_geometry_goldens(hip, golden):
    """area / centroids kernels == connectivity.area / centroids of the reference, bit for bit."""
    g = golden("g5_geometry.npz")
    for t in ("tri", "quad", "mix"):
        mesh = hip.engine.DeviceMesh(g[t + "_xy"], g[t + "_faces"])
        assert np.array_equal(mesh.area(), g[t + "_area"])
        assert np.array_equal(mesh.centroids(), g[t + "_centroids"])


METHODS = [
    ("mean", 0, 0.0), ("harmonic_mean", 1, 0.0), ("geometric_mean", 2, 0.0), ("sum", 3, 0.0),
    ("minimum", 4, 0.0), ("maximum", 5, 0.0), ("mode", 6, 0.0), ("median", 7, 50.0), ("p5", 7, 5.0),
    ("p10", 7, 10.0), ("p25", 7, 25.0), ("p50", 7, 50.0), ("p75", 7, 75.0), ("p90", 7, 90.0), ("p95", 7, 95.0),
    ("p33.3", 7, 33.3), ("p0", 7, 0.0), ("p100", 7, 100.0), ("first_order_conservative", 8, 0.0),
    ("conductance", 8, 0.0), ("max_overlap", 9, 0.0),
]


@pytest.mark.parametrize("name,mid,p", METHODS)
def test_apply_golden_g2(hip, golden, name, mid, p):
    """every reducer on the reference's own (inputs, outputs): make_regrid(f)._regrid, golden G2."""
    g = golden("g2_apply.npz")
    T, S = int(g["T"]), int(g["S"])
    csr = hip.engine.DeviceCSR.from_arrays(g["data"], g["indices"], g["indptr"], T, S)
    for tag in ("64", "32"):
        got = csr.apply(g["src" + tag], mid, p)
        exp = g[f"out{tag}_{name}"]
        assert got.dtype == np.float64 and got.shape == exp.shape
        if name == "geometric_mean":
            np.testing.assert_allclose(got, exp, rtol=RTOL_GEOMETRIC, equal_nan=True)
        else:
            # rows of up to LONG_ROW entries: bit-identical to the reference loop; longer rows (here 33-40 entries)
            # are reduced cooperatively in a fixed tree order
            assert_apply_equal(got, exp, g["indptr"], name)


def test_apply_coo_golden(hip, golden):
    g = golden("g2_apply.npz")
    got = hip.engine.apply_coo(g["coo_row"], g["coo_col"], int(g["T"]), g["src64"])
    assert same_or_nan(got, g["coo_out64"]).all()


def test_csr_from_triplet_golden(hip, golden):
    g = golden("g3_csr.npz")
    csr = hip.engine.DeviceCSR.from_triplet(g["row"], g["col"], g["data"], int(g["n"]), int(g["m"]))
    data, idx, indptr = csr.download()
    assert np.array_equal(indptr, g["indptr"]) and np.array_equal(idx, g["indices"]) and np.array_equal(data, g["data"])
    with pytest.raises(ValueError):  # rows must be sorted (core/sparse.py:65)
        hip.engine.DeviceCSR.from_triplet(np.array([1, 0]), np.array([0, 0]), np.ones(2), 2, 1)


def test_apply_vs_oracle_on_overlap_weights(hip, oracle):
    sxy, sf = meshgen.triangle_mesh(2500, 0)
    txy, tf = meshgen.triangle_mesh(2000, 1, 30.0, 0.7)
    csr, _, idx, data, indptr = gpu_triplets(hip, sxy, sf, txy, tf)
    v = meshgen.smooth_field(oracle.centroids(sxy, sf), 0, nan_fraction=0.05)
    src = np.stack([v, np.abs(v) + 0.1, np.round(3 * v), -np.abs(v), np.zeros_like(v)])
    for name, mid, p in METHODS:
        m = ("percentile", p) if mid == 7 else name
        for dtype in (np.float64, np.float32):
            s_ = src.astype(dtype)
            got = csr.apply(s_, mid, p)
            exp = oracle.regrid_csr(m, s_.astype(np.float64), data, idx, indptr, csr.n)
            if name == "geometric_mean":
                np.testing.assert_allclose(got, exp, rtol=RTOL_GEOMETRIC, equal_nan=True)
            else:
                assert_apply_equal(got, exp, indptr, (name, dtype))
    # K not a multiple of the k-tile, and K = 1
    for K in (1, 7, 9, 17):
        s_ = np.random.default_rng(K).normal(size=(K, csr.m))
        got = csr.apply(s_, 0)
        assert_apply_equal(got, oracle.regrid_csr("mean", s_, data, idx, indptr, csr.n), indptr)
    with pytest.raises(ValueError):
        csr.apply(np.ones((1, csr.m + 1)))
    with pytest.raises(ValueError):
        csr.apply(np.ones((1, csr.m)), 7, 101.0)
    # integer input is promoted like the reference's float64 workspace
    got = csr.apply(np.arange(csr.m)[None, :], 0)
    exp = oracle.regrid_csr("mean", np.arange(csr.m, dtype=np.float64)[None, :], data, idx, indptr, csr.n)
    assert_apply_equal(got, exp, indptr)


def test_apply_long_rows(hip, oracle):
    """coarse target over a fine source: rows of thousands of entries are block-reduced."""
    sxy, sf = meshgen.triangle_mesh(40000, 3)
    txy, tf = meshgen.quad_mesh(np.linspace(-0.1, 1.1, 9), np.linspace(0.0, 1.0, 6))
    csr, _, idx, data, indptr = gpu_triplets(hip, sxy, sf, txy, tf)
    assert (np.diff(indptr) > LONG_ROW).sum() > 10
    v = meshgen.smooth_field(oracle.centroids(sxy, sf), 0, nan_fraction=0.05)
    src = np.stack([v, np.abs(v) + 0.1, np.round(3 * v)])
    for name, mid, p in METHODS:
        m = ("percentile", p) if mid == 7 else name
        got = csr.apply(src, mid, p)
        exp = oracle.regrid_csr(m, src, data, idx, indptr, csr.n)
        if mid in (6, 7, 4, 5, 9):  # order-independent reducers stay exact
            assert same_or_nan(got, exp).all(), name
        else:
            np.testing.assert_allclose(got, exp, rtol=RTOL_LONG, equal_nan=True, err_msg=name)
    # the same matrix uploaded from host arrays (from_weights path) takes the same route
    up = hip.engine.DeviceCSR.from_arrays(data, idx, indptr, csr.n, csr.m)
    assert np.array_equal(up.apply(src, 0), csr.apply(src, 0), equal_nan=True)
    q = np.repeat(np.arange(csr.n), np.diff(indptr))
    tr = hip.engine.DeviceCSR.from_triplet(q, idx, data, csr.n, csr.m)
    assert np.array_equal(tr.apply(src, 0), csr.apply(src, 0), equal_nan=True)


def test_apply_single_variable_one_launch(hip, oracle, monkeypatch):
    """K = 1 takes the one-launch kernel (wave-private staging windows, the long rows by blocks in front of the same
    grid): every reducer, float64 and float32 sources, on a matrix with empty rows, short rows, rows of 33 ... 2048
    entries (lane groups / waves) and rows beyond 2048 entries (a block each); windows that overflow their 320 entries;
    an uploaded copy of the same matrix (no stored row order) gives the same numbers; so does the block-wide kernel it
    replaced (XR_APPLY_K1=block cannot be flipped inside one process: compared through K = 2, which takes the direct
    kernel, row for row)."""
    rng = np.random.default_rng(23)
    T, S = 6000, 5000
    counts = rng.integers(0, 9, T)
    counts[rng.choice(T, 300, replace=False)] = 0            # empty rows -> NaN
    counts[rng.choice(T, 64, replace=False)] = rng.integers(20, 33, 64)     # long-ish rows: windows of > 320 entries per 64 rows
    counts[100:164] = 32                                       # a whole wave of 32-entry rows: 2048 entries, seven chunks
    counts[rng.choice(T, 40, replace=False)] = rng.integers(33, 600, 40)    # lane groups
    counts[rng.choice(T, 12, replace=False)] = rng.integers(600, 2049, 12)  # whole waves
    counts[[7, 4001]] = [2500, 3100]                           # beyond the wave kernel's reach: one block each
    indptr = np.concatenate([[0], np.cumsum(counts)])
    idx = np.concatenate([np.sort(rng.choice(S, c, replace=False)) for c in counts]) if indptr[-1] else np.zeros(0, int)
    data = rng.uniform(0.1, 2.0, indptr[-1])
    data[rng.random(indptr[-1]) < 0.02] = 0.0                  # zero weights
    csr = hip.engine.DeviceCSR.from_arrays(data, idx, indptr, T, S)
    v = rng.normal(size=S)
    v[rng.random(S) < 0.05] = np.nan
    sources = {"mixed": v, "positive": np.abs(v) + 0.1, "ties": np.round(2 * v), "all_nan": np.full(S, np.nan)}
    short = counts <= LONG_ROW
    for name, mid, p in METHODS:
        m = ("percentile", p) if mid == 7 else name
        for tag, vec in sources.items():
            for dtype in (np.float64, np.float32):
                one = vec.astype(dtype)[None, :]
                got = csr.apply(one, mid, p)
                exp = oracle.regrid_csr(m, one.astype(np.float64), data, idx, indptr, T)
                assert got.shape == (1, T)
                assert np.array_equal(np.isnan(got), np.isnan(exp)), (name, tag, dtype)
                if name == "geometric_mean":
                    np.testing.assert_allclose(got, exp, rtol=RTOL_GEOMETRIC, equal_nan=True)
                    continue
                # rows of up to 32 entries: the reference loop's additions in its order -> bit for bit
                assert same_or_nan(got[:, short], exp[:, short]).all(), (name, tag, dtype)
                if mid in (6, 7, 4, 5, 9):
                    assert same_or_nan(got, exp).all(), (name, tag, dtype)
                else:
                    # (sums of mixed sign cancel: the cooperative order of the long rows then shows beyond 1e-13 relative
                    # to the -- small -- result: an absolute floor of 1e-12 x the row's scale; 1e-6 for sums of w / v)
                    signed = tag in ("mixed", "ties")
                    loose = name == "harmonic_mean" and signed
                    np.testing.assert_allclose(got, exp, rtol=1e-6 if loose else (1e-9 if name == "harmonic_mean" else RTOL_LONG),
                                               atol=1e-9 if loose else (1e-11 if signed else 0.0), equal_nan=True,
                                               err_msg=f"{name} {tag}")
                # the two-variable path (another kernel family) agrees row for row on the short rows
                two = csr.apply(np.concatenate([one, one]), mid, p)
                assert same_or_nan(two[:1, short], got[:, short]).all() and same_or_nan(two[1:, short], got[:, short]).all()
    # a matrix BUILT on the device (stored row order, big rows last) through the same kernel: covered against the oracle
    sxy, sf = meshgen.triangle_mesh(30000, 3)
    txy, tf = meshgen.quad_mesh(np.linspace(-0.1, 1.1, 40), np.linspace(0.0, 1.0, 3))
    built, _, bidx, bdata, bindptr = gpu_triplets(hip, sxy, sf, txy, tf)
    bv = meshgen.smooth_field(oracle.centroids(sxy, sf), 1, nan_fraction=0.03)[None, :]
    for name, mid, p in (("mean", 0, 0.0), ("sum", 3, 0.0), ("maximum", 5, 0.0), ("max_overlap", 9, 0.0)):
        got = built.apply(bv, mid, p)
        exp = oracle.regrid_csr(name, bv, bdata, bidx, bindptr, built.n)
        if mid in (5, 9):
            assert same_or_nan(got, exp).all(), name
        else:
            np.testing.assert_allclose(got, exp, rtol=RTOL_LONG, atol=1e-11, equal_nan=True, err_msg=name)


def grid2d():
    xy = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [0.0, 1.0], [1.0, 1.0], [2.0, 1.0], [1.0, 2.0]])
    faces = np.array([[0, 1, 4, 3], [1, 2, 5, 4], [3, 4, 6, -1], [4, 5, 6, -1]])
    return xy, faces


def test_locate_and_barycentric_known_answers(hip):
    xy, faces = grid2d()
    mesh = hip.engine.DeviceMesh(xy, faces)
    assert np.array_equal(mesh.locate_points(mesh.centroids()), [0, 1, 2, 3])
    off = np.array([[-0.01, 1.0], [-0.01, 0.5]])
    assert np.array_equal(mesh.locate_points(off, 0.011), [0, 0])  # tests/test_ugrid2d.py:724-730
    # tests/test_ugrid2d.py:835-885, 1085-1108: the reference's point selections (locate_points results)
    oob = np.array([[-10.0, -10.0], [0.5, 0.5], [-20.0, -20.0], [1.5, 1.25], [-30.0, -30.0]])
    assert np.array_equal(mesh.locate_points(oob), [-1, 0, -1, 3, -1])
    gx, gy = np.meshgrid([0.4, 0.8, 1.2], [0.5, 1.1])
    assert np.array_equal(mesh.locate_points(np.column_stack([gx.ravel(), gy.ravel()])), [0, 0, 1, 2, 2, 3])
    pts = np.array([[0.0, 0.0], [0.5, 0.5], [1.5, 0.5], [0.5, 1.5], [2.0, 2.0]])
    face, w = mesh.compute_barycentric_weights(pts)  # tests/test_ugrid2d.py:751-791
    assert np.array_equal(face, [0, 0, 1, 2, -1])
    expected = np.array([[1.0, 0, 0, 0], [0.25] * 4, [0.25] * 4, [0.5, 0.0, 0.5, 0.0], [0.0] * 4])
    np.testing.assert_allclose(w, expected, atol=1e-15)
    pts[:, 0] -= 0.01
    face, w = mesh.compute_barycentric_weights(pts, tolerance=0.01)
    assert np.array_equal(face, [-1, 0, 1, 2, -1])
    expected[0] = 0.0
    np.testing.assert_allclose(w, expected, atol=0.05)


def test_locate_with_more_candidates_than_parking_slots(hip, oracle):
    """Stacked faces: 40 triangles and 25 quads that all cover the middle of a regular mesh.  A point there has far more
    candidate faces than the locate kernels park per point (8): those points take the plain walk; the lowest face id wins as
    in the oracle, with and without a tolerance, and the barycentric weights follow."""
    rng = np.random.default_rng(23)
    xy, faces = meshgen.triangle_mesh(1500, 2)
    lo, hi = xy.min(0), xy.max(0)
    mid, span = 0.5 * (lo + hi), (hi - lo).max()
    extra_xy, extra_faces = [], []
    for k in range(40):
        ang = rng.uniform(0, 2 * np.pi) + np.array([0.0, 2.1, 4.2])
        r = span * rng.uniform(0.05, 0.3)
        extra_faces.append(len(xy) + len(extra_xy) + np.arange(3))
        extra_xy.extend(mid + rng.normal(0, 0.01 * span, 2) + r * np.column_stack([np.cos(ang), np.sin(ang)]))
    stacked = np.vstack([xy, np.array(extra_xy)])
    all_faces = np.vstack([faces, np.array(extra_faces)])
    order = rng.permutation(all_faces.shape[0])  # the stacked faces get ids anywhere in the range
    all_faces = all_faces[order]
    pts = np.vstack([mid + rng.normal(0, 0.08 * span, (6000, 2)), rng.uniform(lo - 0.05 * span, hi + 0.05 * span, (6000, 2))])
    pts[:200] = stacked[rng.integers(0, stacked.shape[0], 200)]  # on vertices
    mesh, tree = hip.engine.DeviceMesh(stacked, all_faces), oracle.CellTree2d(stacked, all_faces)
    for tol in (None, 0.0, 1e-3 * span):
        got, exp = mesh.locate_points(pts, tol), tree.locate_points(pts, tol)
        assert np.array_equal(got, exp)
        fo, wo = tree.compute_barycentric_weights(pts, tol)
        fg, wg = mesh.compute_barycentric_weights(pts, tol)
        assert np.array_equal(fo, fg) and np.array_equal(wo, wg)
    # quads stacked on a quad mesh (the M = 4 kernels)
    qxy, qf = meshgen.quad_mesh(np.linspace(0.0, 1.0, 31), np.linspace(0.0, 1.0, 27))
    ex, ef = [], []
    for k in range(25):
        c, hw = np.array([0.5, 0.5]) + rng.normal(0, 0.02, 2), rng.uniform(0.05, 0.3, 2)
        ef.append(len(qxy) + len(ex) + np.arange(4))
        ex.extend(c + hw * np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]]))
    sq, sqf = np.vstack([qxy, np.array(ex)]), np.vstack([qf, np.array(ef)])
    sqf = sqf[rng.permutation(sqf.shape[0])]
    qpts = np.vstack([0.5 + rng.normal(0, 0.1, (5000, 2)), rng.uniform(-0.05, 1.05, (5000, 2))])
    mesh, tree = hip.engine.DeviceMesh(sq, sqf), oracle.CellTree2d(sq, sqf)
    for tol in (None, 1e-4):
        assert np.array_equal(mesh.locate_points(qpts, tol), tree.locate_points(qpts, tol))
    assert (np.bincount(tree.locate_points(qpts) + 1)[1:] > 0).sum() > 300  # (many different faces win)


def test_locate_and_barycentric_vs_oracle(hip, oracle):
    from xugrid_amd import connectivity as C, voronoi

    xy, faces = meshgen.triangle_mesh(5000, 0)
    rng = np.random.default_rng(5)
    pts = rng.random((20000, 2)) * 1.1 - 0.05
    # points exactly on vertices and edge midpoints (ties -> lowest face index)
    pts[:500] = xy[rng.integers(0, len(xy), 500)]
    e = faces[rng.integers(0, len(faces), 500)]
    pts[500:1000] = 0.5 * (xy[e[:, 0]] + xy[e[:, 1]])
    mesh, tree = hip.engine.DeviceMesh(xy, faces), oracle.CellTree2d(xy, faces)
    for tol in (None, 0.0, 1e-3):
        assert np.array_equal(mesh.locate_points(pts, tol), tree.locate_points(pts, tol))
        fo, wo = tree.compute_barycentric_weights(pts, tol)
        fg, wg = mesh.compute_barycentric_weights(pts, tol)
        assert np.array_equal(fo, fg) and np.array_equal(wo, wg)
    # polygons (voronoi cells incl. exterior) -> Wachspress branch
    cen = oracle.centroids(xy, faces)
    enc, fec = C.edge_connectivity(faces)
    v, cells, _, _ = voronoi.voronoi_topology(C.invert_dense_to_sparse(faces, n_rows=len(xy)), xy, cen,
                                              C.invert_dense(fec), enc, True, True, True)
    mesh, tree = hip.engine.DeviceMesh(v, cells), oracle.CellTree2d(v, cells)
    fo, wo = tree.compute_barycentric_weights(pts)
    fg, wg = mesh.compute_barycentric_weights(pts)
    assert np.array_equal(fo, fg) and np.array_equal(wo, wg)
    inside = fo >= 0
    np.testing.assert_allclose(wo[inside].sum(axis=1), 1.0, rtol=1e-12)


def test_elevation_nl_config1(hip, oracle, golden):
    """BASELINE config 1: elevation_nl (5248 triangles) -> 200 x 200 raster, mean."""
    g = golden("g8_elevation_nl.npz")
    xy = np.column_stack([g["node_x"], g["node_y"]])
    faces = g["face_nodes"].astype(np.int64)
    xmin, ymin, xmax, ymax = xy[:, 0].min(), xy[:, 1].min(), xy[:, 0].max(), xy[:, 1].max()
    txy, tf = meshgen.quad_mesh(np.linspace(xmin, xmax, 201), np.linspace(ymin, ymax, 201))
    csr, (data, idx, indptr) = assert_overlap_parity(hip, oracle, xy, faces, txy, tf)
    elev = g["elevation"][None, :]
    assert elev.dtype == np.float32
    got = csr.apply(elev, 0)
    exp = oracle.regrid_csr("mean", elev.astype(np.float64), data, idx, indptr, csr.n)
    assert_apply_equal(got, exp, indptr)
    total = oracle.area(xy, faces).sum()
    assert abs(data.sum() / total - 1) < 1e-10
    valid = ~np.isnan(got)
    assert got[valid].min() >= -60.67 and got[valid].max() <= 252.74


@pytest.mark.parametrize("n_points", [500_000])
def test_full_size_properties(hip, n_points):
    """BASELINE config 2 size (1M -> 1M triangles): size-independent properties of the result."""
    sxy, sf = meshgen.triangle_mesh(n_points, 0, delaunay=False)
    txy, tf = meshgen.triangle_mesh(n_points, 1, 30.0, 0.7, delaunay=False)
    E = hip.engine
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    csr = ms.overlap(mt)
    data, idx, indptr = csr.download()
    assert csr.n == tf.shape[0] and csr.m == sf.shape[0] and data.size == csr.nnz == indptr[-1]
    assert (data > 0).all() and idx.min() >= 0 and idx.max() < csr.m
    # rows strictly ascending in the source index
    inner = np.ones(idx.size, dtype=bool)
    inner[indptr[:-1][np.diff(indptr) > 0]] = False
    assert (np.diff(idx)[inner[1:]] > 0).all()
    # the target mesh lies inside the source hull: every target face is fully covered
    t_area = mt.area()
    row_sum = np.add.reduceat(data, indptr[:-1][np.diff(indptr) > 0])
    full = np.diff(indptr) > 0
    np.testing.assert_allclose(row_sum, t_area[full], rtol=1e-9)
    assert full.all()
    # column sums never exceed the source face area; total area is conserved
    col_sum = np.bincount(idx, weights=data, minlength=csr.m)
    assert (col_sum <= ms.area() * (1 + 1e-9)).all()
    assert abs(data.sum() / t_area.sum() - 1) < 1e-10
    # a constant field regrids to the same constant; linearity of mean in the data
    one = csr.apply(np.full((1, csr.m), 3.25))
    assert np.array_equal(one, np.full((1, csr.n), 3.25)) or np.allclose(one, 3.25, rtol=1e-14)
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(1, csr.m)), rng.normal(size=(1, csr.m))
    lhs = csr.apply(2.0 * a + b)
    rhs = 2.0 * csr.apply(a) + csr.apply(b)
    np.testing.assert_allclose(lhs, rhs, rtol=1e-10, atol=1e-12)
    # determinism: rebuilding gives the identical matrix
    ms.invalidate(); mt.invalidate()
    d2, i2, p2 = ms.overlap(mt).download()
    assert np.array_equal(d2, data) and np.array_equal(i2, idx) and np.array_equal(p2, indptr)
    # relative weights: column sums == covered fraction <= 1; conservative regridding of ones
    rel = ms.overlap(mt, relative=True)
    rd, ri, rp = rel.download()
    assert np.array_equal(ri, idx) and np.array_equal(rd, data / ms.area()[idx])


def test_apply_many_variables_row_tiling(hip, oracle):
    """K >= 8 applies regroup the stored rows into 2-D tiles once (keys from xr_overlap, or from
    xr_csr_set_row_keys for uploaded weights): results, downloads and later K = 1 applies are unchanged."""
    from xugrid_amd import engine as E

    sxy, sf = meshgen.triangle_mesh(20000, 3, delaunay=False)
    txy, tf = meshgen.triangle_mesh(24000, 4, 30.0, 0.8, delaunay=False)
    tree = oracle.CellTree2d(sxy, sf, -1)
    oq, os_, oa = tree.intersect_faces(txy, tf, -1)
    indptr = oracle.to_csr_indptr(oq, tf.shape[0])
    rng = np.random.default_rng(5)
    v = rng.normal(size=(24, sf.shape[0]))
    v[3, ::11] = np.nan
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    csr = ms.overlap(mt)
    before = csr.download()
    out1 = csr.apply(v[:1], 0)
    for method, mid in (("mean", 0), ("maximum", 5), ("sum", 3)):
        got = csr.apply(v, mid)  # first call re-tiles the rows
        exp = oracle.regrid_csr(method, v, oa, os_, indptr, csr.n)
        assert_apply_equal(got, exp, indptr, method)
    after = csr.download()
    for a, b in zip(before, after):
        assert np.array_equal(a, b)
    assert same_or_nan(csr.apply(v[:1], 0), out1).all()
    assert_apply_equal(csr.apply(v, 7, 50.0), oracle.regrid_csr("median", v, oa, os_, indptr, csr.n), indptr, "median")
    # uploaded weights + explicit keys
    up = E.DeviceCSR.from_arrays(oa, os_, indptr, csr.n, csr.m)
    keys, key_range = E.morton_row_keys(oracle.centroids(txy, tf), faces_per_tile=64)
    assert key_range > 1
    up.set_row_keys(keys, key_range)
    assert_apply_equal(up.apply(v, 0), oracle.regrid_csr("mean", v, oa, os_, indptr, csr.n), indptr, "uploaded")
    d, i, p = up.download()
    assert np.array_equal(d, oa) and np.array_equal(i, os_) and np.array_equal(p, indptr)
    # keys for rows that are already stored in a spatial order: per CALLER row, the permutations compose (finer tiles here)
    keys2, key_range2 = E.morton_row_keys(oracle.centroids(txy, tf), faces_per_tile=4)
    up.set_row_keys(keys2, key_range2)
    assert_apply_equal(up.apply(v, 0), oracle.regrid_csr("mean", v, oa, os_, indptr, csr.n), indptr, "re-keyed")
    d, i, p = up.download()
    assert np.array_equal(d, oa) and np.array_equal(i, os_) and np.array_equal(p, indptr)
    with pytest.raises(ValueError):
        E.DeviceCSR.from_arrays(oa, os_, indptr, csr.n, csr.m).set_row_keys(keys + key_range, key_range)


def test_apply_plan_per_block_and_merged(hip, oracle, monkeypatch, xr_option):
    """The two forms of the many-variable apply plan (xr_apply.hip: ensure_plan) -- a distinct-column list per block of 256 rows,
    or one per group of neighbouring blocks (chosen by itself when the blocks use the source lines poorly: qhull numberings) -- and the
    automatic choice give the oracle's numbers, NaNs and a ragged last group included."""
    from xugrid_amd import engine as E

    sxy, sf = meshgen.triangle_mesh(30000, 5, delaunay=True)
    txy, tf = meshgen.triangle_mesh(33000, 6, 30.0, 0.8, delaunay=True)
    tree = oracle.CellTree2d(sxy, sf, -1)
    oq, os_, oa = tree.intersect_faces(txy, tf, -1)
    indptr = oracle.to_csr_indptr(oq, tf.shape[0])
    rng = np.random.default_rng(11)
    v = rng.normal(size=(20, sf.shape[0]))
    v[5, ::13] = np.nan
    v[11] = np.nan
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    for mode in ("0", "1", None):
        if mode is None:
            xr_option("plan_merge", None)
        else:
            xr_option("plan_merge", mode)
        csr = ms.overlap(mt)  # (a fresh matrix: the plan is built once per matrix, on its first many-variable apply)
        for method, mid in (("mean", 0), ("maximum", 5), ("sum", 3), ("minimum", 4)):
            assert_apply_equal(csr.apply(v, mid), oracle.regrid_csr(method, v, oa, os_, indptr, csr.n), indptr, f"{method} merge={mode}")
        v32 = v[:9].astype(np.float32)
        assert_apply_equal(csr.apply(v32, 0), oracle.regrid_csr("mean", v32, oa, os_, indptr, csr.n), indptr, f"f32 merge={mode}")
    xr_option("plan_merge", None)


def test_overlap_projected_coordinates(hip, oracle):
    """UTM-like coordinates (offsets of 5e5 / 6e6 m, 10-50 m cells): the f32 record bboxes are stored relative
    to the grid origin and rounded outwards, so nothing is lost against the f64 oracle."""
    sxy, sf = meshgen.triangle_mesh(4000, 11)
    txy, tf = meshgen.triangle_mesh(5000, 12, 20.0, 0.9)
    off = np.array([512_345.678, 6_123_456.789])
    sxy, txy = off + 3000.0 * sxy, off + 3000.0 * txy
    assert_overlap_parity(hip, oracle, sxy, sf, txy, tf)
    assert_overlap_parity(hip, oracle, sxy, sf, txy, tf, relative=True)
    E = hip.engine
    pts = off + 3000.0 * np.random.default_rng(0).random((20000, 2)) * 1.1 - 100.0
    tree = oracle.CellTree2d(sxy, sf, -1)
    assert np.array_equal(E.DeviceMesh(sxy, sf).locate_points(pts), tree.locate_points(pts))
    fo, wo = tree.compute_barycentric_weights(pts)
    fg, wg = E.DeviceMesh(sxy, sf).compute_barycentric_weights(pts)
    assert np.array_equal(fg, fo) and np.array_equal(wg, wo)


def test_device_pipelines_with_empty_sides(hip):
    """locate / barycentric CSR builders with an empty query or an empty tree."""
    E = hip.engine
    sxy, sf = meshgen.triangle_mesh(300, 2)
    src = E.DeviceMesh(sxy, sf)
    empty = E.DeviceMesh(sxy, np.zeros((0, 3), dtype=np.int64))
    c = E.locate_csr(src, query=empty)
    assert (c.n, c.m, c.nnz) == (0, sf.shape[0], 0)
    c = E.locate_csr(empty, query=src)
    assert (c.n, c.m, c.nnz) == (sf.shape[0], 0, 0)
    assert np.isnan(c.apply(np.zeros((2, 0)), E.METHOD_IDS["select"])).all()
    c = E.locate_csr(src, points=np.zeros((0, 2)))
    assert (c.n, c.nnz) == (0, 0)
    # all points outside
    c = E.locate_csr(src, points=np.full((5, 2), 99.0))
    assert (c.n, c.nnz) == (5, 0)
    out = c.apply(np.ones((1, sf.shape[0])), E.METHOD_IDS["select"])
    assert out.shape == (1, 5) and np.isnan(out).all()


def test_apply_sorted_rows_mode_and_percentiles(hip, oracle):
    """rows of 33..2048 entries take the wave-sorted kernel for mode / percentiles: values must be IDENTICAL to the
    reference loops (order statistics and left-to-right weight sums do not depend on the search strategy).  Data with
    ties, NaN, +-inf, signed zeros and zero weights; float32 sources as well."""
    from xugrid_amd import engine as E

    rng = np.random.default_rng(12)
    T, S = 900, 5000
    lens = np.concatenate([rng.integers(33, 300, 700), rng.integers(300, 2048, 150), rng.integers(0, 33, 50)])
    rng.shuffle(lens)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(indptr[-1])
    indices = np.concatenate([np.sort(rng.choice(S, n, replace=n > S)) for n in lens]).astype(np.int64)
    data = rng.random(nnz)
    data[rng.random(nnz) < 0.05] = 0.0
    zero_rows = rng.choice(T, 20, replace=False)
    for r in zero_rows[:10]:
        data[indptr[r]:indptr[r + 1]] = 0.0  # all weights zero -> NaN
    src = np.empty((4, S))
    src[0] = rng.integers(0, 7, S)                      # categorical: many ties
    src[1] = rng.normal(size=S)
    src[2] = np.where(rng.random(S) < 0.3, np.nan, rng.integers(-2, 3, S) * 0.5)  # NaN + signed zeros
    src[2][rng.random(S) < 0.05] = -0.0
    src[3] = rng.normal(size=S)
    src[3][rng.random(S) < 0.02] = np.inf
    src[3][rng.random(S) < 0.02] = -np.inf
    src[3][rng.random(S) < 0.1] = np.nan
    nan_rows = zero_rows[10:]
    for r in nan_rows:  # all-NaN rows
        src[2][indices[indptr[r]:indptr[r + 1]]] = np.nan
    csr = E.DeviceCSR.from_arrays(data, indices, indptr, T, S)
    for source in (src, src.astype(np.float32)):
        for method, mid, p in (("mode", 6, 0.0), ("median", 7, 50.0), (("percentile", 33.3), 7, 33.3),
                               (("percentile", 0.0), 7, 0.0), (("percentile", 100.0), 7, 100.0), (("percentile", 95.0), 7, 95.0)):
            got = csr.apply(source, mid, p)
            exp = oracle.regrid_csr(method, source, data, indices, indptr, T)
            bad = ~same_or_nan(got, exp)
            assert not bad.any(), (method, int(bad.sum()), np.argwhere(bad)[:5], got[bad][:5], exp[bad][:5])
            if method == "mode":
                assert np.array_equal(np.signbit(got[~np.isnan(got)]), np.signbit(exp[~np.isnan(exp)]))


def test_full_size_exact_vs_oracle(hip, oracle):
    """BASELINE config 2 itself (1M -> 1M Delaunay triangles): the whole weight matrix against the CPU oracle --
    pair sets and areas bit for bit -- and every streaming reducer on it (rows of up to 32 entries bit-identical)."""
    sxy, sf = meshgen.triangle_mesh(500_000, 0)
    txy, tf = meshgen.triangle_mesh(500_000, 1, 30.0, 0.7)
    csr, (data, idx, indptr) = assert_overlap_parity(hip, oracle, sxy, sf, txy, tf)
    assert csr.nnz > 4_000_000
    v = meshgen.smooth_field(oracle.centroids(sxy, sf), 0, nan_fraction=0.01)[None, :]
    for name, mid in (("mean", 0), ("maximum", 5), ("max_overlap", 9), ("sum", 3)):
        assert_apply_equal(csr.apply(v, mid), oracle.regrid_csr(name, v, data, idx, indptr, csr.n), indptr, name)


def test_apply_contracted_is_opt_in_and_within_its_bound(hip, oracle, xr_option):
    """Option "apply_contract" (round 6): fused multiply-adds and one reciprocal per row in the many-variable kernel.  Off by
    default -- the default is the reference's operation order, bit for bit.  On, every value differs from the oracle by at most
    (n + 2) ulp of sum |w v| / sum w (n = entries of the row); NaN patterns are identical; both plans (per block, merged)."""
    E = hip.engine
    sxy, sf = meshgen.triangle_mesh(20000, 41)
    txy, tf = meshgen.triangle_mesh(16000, 42, 30.0, 0.75)
    oq, os_, oa = oracle.CellTree2d(sxy, sf).intersect_faces(txy, tf)
    indptr = oracle.to_csr_indptr(oq, tf.shape[0])
    rng = np.random.default_rng(9)
    v = rng.normal(size=(24, sf.shape[0]))
    v[3, ::17] = np.nan  # (a tile with NaNs takes the exact path in both modes)
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    eps = np.finfo(np.float64).eps
    counts = np.diff(indptr)
    for merge in (0, 1):
        xr_option("plan_merge", merge)
        csr = ms.overlap(mt)
        for method, mid in (("mean", 0), ("first_order_conservative", 8)):
            ref = oracle.regrid_csr(method, v, oa, os_, indptr, csr.n)
            assert_apply_equal(csr.apply(v, mid), ref, indptr, f"{method} exact merge={merge}")
            xr_option("apply_contract", 1)
            got = csr.apply(v, mid)
            xr_option("apply_contract", 0)
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            absw = oracle.regrid_csr(method, np.abs(np.nan_to_num(v)), oa, os_, indptr, csr.n)  # sum |w v| (/ sum w for the mean)
            bound = (counts[None, :] + 2) * eps * np.nan_to_num(absw)
            ok = ~np.isnan(ref)
            assert (np.abs(got[ok] - ref[ok]) <= bound[ok]).all(), (method, merge, float(np.max(np.abs(got[ok] - ref[ok]) / np.maximum(bound[ok], 1e-300))))
            assert (got[ok] != ref[ok]).any()  # (the switch does switch)
    xr_option("plan_merge", None)


def test_apply_host_arrays_in_chunks(hip, oracle, monkeypatch, xr_option):
    """xr_apply_csr stages the stacked variables through the device in chunks (any K fits): same result for any
    chunk size (XR_APPLY_CHUNK_BYTES is a test hook)."""
    from xugrid_amd import engine as E

    sxy, sf = meshgen.triangle_mesh(1500, 0)
    txy, tf = meshgen.triangle_mesh(1800, 1, 30.0, 0.7)
    csr = E.DeviceMesh(sxy, sf).overlap(E.DeviceMesh(txy, tf))
    v = np.random.default_rng(3).normal(size=(37, csr.m))
    whole = csr.apply(v, 0)
    per_k = csr.m * 8 + csr.n * 8
    for budget in (per_k, 5 * per_k + 1, 36 * per_k):
        xr_option("apply_chunk_bytes", str(budget))
        assert np.array_equal(csr.apply(v, 0), whole, equal_nan=True)
        v32 = v.astype(np.float32)
        assert np.array_equal(csr.apply(v32, 7, 50.0), csr.apply(v32.astype(np.float64), 7, 50.0), equal_nan=True)


def test_apply_columns_renumbered_by_spatial_key(hip, oracle):
    """xr_csr_set_col_keys: the columns (source cells) are renumbered by a spatial key so that the gathers of a row
    block are neighbours in memory; entry order inside the rows is untouched -> results bit-identical, downloads in
    the caller's ids, and a source block handed over in the stored order (expect_permuted) gives the same numbers."""
    from xugrid_amd import engine as E

    sxy, sf = meshgen.triangle_mesh(20000, 3)  # qhull numbering
    txy, tf = meshgen.triangle_mesh(24000, 4, 30.0, 0.8)
    csr = E.DeviceMesh(sxy, sf).overlap(E.DeviceMesh(txy, tf))
    data, idx, indptr = csr.download()
    rng = np.random.default_rng(6)
    v = rng.normal(size=(24, csr.m))
    v[5, ::7] = np.nan
    ref = {mid: csr.apply(v, mid) for mid in (0, 3, 5, 9)}
    ref1 = csr.apply(v[:1], 0)
    med = csr.apply(v[:4], 7, 50.0)
    keys, key_range = E.morton_row_keys(oracle.centroids(sxy, sf), faces_per_tile=64)
    csr.set_col_keys(keys, key_range)
    order = csr.col_order()
    assert np.array_equal(np.sort(order), np.arange(csr.m)) and not np.array_equal(order, np.arange(csr.m))
    d2, i2, p2 = csr.download()
    assert np.array_equal(d2, data) and np.array_equal(i2, idx) and np.array_equal(p2, indptr)
    for mid, exp in ref.items():
        assert same_or_nan(csr.apply(v, mid), exp).all(), mid
    assert same_or_nan(csr.apply(v[:1], 0), ref1).all()
    assert same_or_nan(csr.apply(v[:4], 7, 50.0), med).all()
    csr.expect_permuted(True)
    vp = np.ascontiguousarray(v[:, order])
    for mid, exp in ref.items():
        assert same_or_nan(csr.apply(vp, mid), exp).all(), mid
    csr.expect_permuted(False)
    assert same_or_nan(csr.apply(v.astype(np.float32), 0), csr.apply(v.astype(np.float32).astype(np.float64), 0)).all()
    with pytest.raises(ValueError):
        csr.set_col_keys(keys, key_range)  # already renumbered


@pytest.mark.parametrize("kind", ["built", "uploaded"])
def test_apply_in_engine_order(hip, oracle, kind):
    """Both sides of the apply in the engine's own order (DeviceCSR.engine_order: xr_csr_set_row_keys on the caller's rows
    -- also for a matrix xr_overlap built, where it replaces the runs-of-16 grouping --, xr_csr_set_col_keys,
    xr_csr_expect_permuted, xr_csr_output_stored_order): ``out[:, r]`` is the caller's row ``row_order[r]``, for every
    kernel family (K = 1 one-launch, direct, planned, long rows, workspace reducers); downloads stay in the caller's
    ids; switching the stored-order output off again gives the caller's rows back."""
    from xugrid_amd import engine as E

    sxy, sf = meshgen.triangle_mesh(20000, 3)
    txy, tf = meshgen.triangle_mesh(24000, 4, 30.0, 0.8)
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    csr = ms.overlap(mt)
    data, idx, indptr = csr.download()
    if kind == "uploaded":
        csr = E.DeviceCSR.from_arrays(data, idx, indptr, csr.n, csr.m)
    rng = np.random.default_rng(16)
    v = rng.normal(size=(40, csr.m))
    v[3, ::5] = np.nan
    cases = [(0, 0.0, 1), (0, 0.0, 3), (0, 0.0, 9), (0, 0.0, 40), (3, 0.0, 40), (5, 0.0, 9), (9, 0.0, 2), (7, 50.0, 4), (6, 0.0, 2)]
    ref = {c: csr.apply(v[: c[2]], c[0], c[1]) for c in cases}
    col_order, row_order = csr.engine_order(ms.centroids(), mt.centroids(), K=40, row_tile=4, col_tile=8)
    assert np.array_equal(np.sort(row_order), np.arange(csr.n)) and not np.array_equal(row_order, np.arange(csr.n))
    assert np.array_equal(np.sort(col_order), np.arange(csr.m))
    vp = np.ascontiguousarray(v[:, col_order])
    for c in cases:
        got = csr.apply(vp[: c[2]], c[0], c[1])
        back = np.empty_like(got)
        back[:, row_order] = got
        assert same_or_nan(back, ref[c]).all(), (kind, c)
    d2, i2, p2 = csr.download()
    assert np.array_equal(d2, data) and np.array_equal(i2, idx) and np.array_equal(p2, indptr)
    csr.output_stored_order(False)
    csr.expect_permuted(False)
    for c in cases[:4]:
        assert same_or_nan(csr.apply(v[: c[2]], c[0], c[1]), ref[c]).all(), (kind, c, "caller order again")


def test_stored_row_order_is_frozen_once_handed_out(hip):
    """The stored row order is settled by the call that hands it out, whatever the number of variables of the later
    applies: read the order FIRST (the old default was K = 1, which left the row tiling pending), then apply 1 and 16
    variables -- both come out in the order that was read.  While the output is delivered in stored order new row keys
    are refused; after switching it off they are accepted and the order read afterwards is the new one."""
    from xugrid_amd import engine as E

    sxy, sf = meshgen.triangle_mesh(20000, 3)
    txy, tf = meshgen.triangle_mesh(24000, 4, 30.0, 0.8)
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    for kind in ("built", "uploaded"):
        csr = ms.overlap(mt)
        if kind == "uploaded":
            data, idx, indptr = csr.download()
            csr = E.DeviceCSR.from_arrays(data, idx, indptr, csr.n, csr.m)
            csr.set_row_keys(*E.morton_row_keys(mt.centroids(), faces_per_tile=64))
        v = np.random.default_rng(3).normal(size=(16, csr.m))
        ref = csr.apply(v[:1]), None
        csr_ref = ms.overlap(mt)
        ref16 = csr_ref.apply(v)
        csr.output_stored_order(True)
        order = csr.row_order()  # (no K: the tiling is settled here)
        for K in (1, 16, 3, 16):
            got = csr.apply(v[:K])
            back = np.empty_like(got)
            back[:, order] = got
            assert same_or_nan(back, ref16[:K]).all(), (kind, K)
        assert np.array_equal(csr.row_order(), order)
        with pytest.raises(ValueError, match="frozen"):
            csr.set_row_keys(*E.morton_row_keys(mt.centroids(), faces_per_tile=4))
        csr.output_stored_order(False)
        assert same_or_nan(csr.apply(v), ref16).all()
        csr.set_row_keys(*E.morton_row_keys(mt.centroids(), faces_per_tile=4))
        csr.output_stored_order(True)
        order2 = csr.row_order()
        assert not np.array_equal(order2, order)
        got = csr.apply(v)
        back = np.empty_like(got)
        back[:, order2] = got
        assert same_or_nan(back, ref16).all(), kind


def test_overlap_apply_in_one_call_matches_two_calls(hip, monkeypatch, xr_option):
    """xr_overlap_apply_dev (weights + their first use in one entry point; for K = 1 the apply is enqueued before the host
    has read the matrix' sizes back) == xr_overlap followed by xr_apply_csr_dev, bit for bit: every reducer family, K = 1
    and K = 3, synchronous and asynchronous mode (xr_set_async), the regrow path (a tiny pair-queue margin makes the first
    attempt fail AFTER its apply was enqueued: the result must come from the final attempt), polling and event mailbox."""
    import ctypes

    from xugrid_amd import _lib, engine as E

    lib = _lib.load()
    sxy, sf = meshgen.triangle_mesh(30000, 5)
    txy, tf = meshgen.triangle_mesh(26000, 6, 30.0, 0.7)
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    S, T = sf.shape[0], tf.shape[0]
    v = np.random.default_rng(8).normal(size=(3, S))
    v[1, ::7] = np.nan
    d_src, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.xr_dev_alloc(8 * 3 * S, ctypes.byref(d_src)))
    _lib.check(lib.xr_dev_alloc(8 * 3 * T, ctypes.byref(d_out)))
    _lib.check(lib.xr_dev_upload(d_src, v.ctypes.data_as(ctypes.c_void_p), 8 * 3 * S))
    ref_csr = ms.overlap(mt)
    ref_w = ref_csr.download()

    def fetch(K):
        out = np.empty((K, T))
        _lib.check(lib.xr_dev_download(out.ctypes.data_as(ctypes.c_void_p), d_out, 8 * K * T))
        return out

    def check_all(tag):
        for method_id, pct, K in ((0, 0.0, 1), (3, 0.0, 1), (5, 0.0, 1), (9, 0.0, 1), (0, 0.0, 3), (6, 0.0, 1), (7, 50.0, 1)):
            ref = ref_csr.apply(v[:K], method_id, pct)
            for a in (False, True):
                E.set_async(a)
                try:
                    ms.invalidate()
                    mt.invalidate()
                    csr = ms.overlap_apply_dev(mt, d_src.value, E.XR_F64, K, d_out.value, method_id, pct)
                    E.dev_sync()
                    got = fetch(K)
                finally:
                    E.set_async(False)
                assert same_or_nan(got, ref).all(), (tag, method_id, K, a)
                assert csr.nnz == ref_csr.nnz
                w = csr.download()
                assert all(np.array_equal(x, y) for x, y in zip(w, ref_w)), (tag, method_id, K, a)
                # the matrix is a normal one afterwards
                assert same_or_nan(csr.apply(v[:K], method_id, pct), ref).all()

    check_all("default")
    xr_option("queue_margin", "64")  # the big faces do not fit the first time: redo, apply enqueued again
    check_all("regrow")
    xr_option("queue_margin", None)
    # many steps back to back without a host synchronisation in between (the benchmark's loop)
    E.set_async(True)
    try:
        for _ in range(20):
            ms.invalidate()
            mt.invalidate()
            keep = ms.overlap_apply_dev(mt, d_src.value, E.XR_F64, 1, d_out.value, 0)
        E.dev_sync()
    finally:
        E.set_async(False)
    assert same_or_nan(fetch(1), ref_csr.apply(v[:1], 0)).all()
    del keep
    lib.xr_dev_free(d_src)
    lib.xr_dev_free(d_out)
