"""
GPU (-m gpu): NetworkGridder.  xr_edge_length_csr -- grid walk + Cyrus-Beck clip + CSR assembly on
the device -- must equal the oracle's intersect_edges post-processed as the reference does (unstructured.py:203-215,
gridder.py:66-73) BIT FOR BIT: same (face, edge) pairs, same lengths, rows ordered by edge id; the regridder built
on it reproduces the known answers of the reference's tests/test_regrid/test_network_gridder.py.
"""
import numpy as np
import pytest

import xugrid_amd as xa
from conftest import same_or_nan
from network_cases import (burn_lines_case, csr_from_pairs, line_selection_cases, line_selection_of_pairs, random_network,
                           raster_quads, reference_case)
from xugrid_amd import engine, meshgen

pytestmark = pytest.mark.gpu


def device_vs_oracle(oracle, nodes, faces, edges):
    tree = oracle.CellTree2d(nodes, faces)
    e, f, xy = tree.intersect_edges(edges)
    w, cols, indptr = csr_from_pairs(e, f, xy, faces.shape[0])
    csr = engine.edge_length_csr(engine.DeviceMesh(nodes, faces), edges)
    assert (csr.n, csr.m, csr.nnz) == (faces.shape[0], edges.shape[0], e.size)
    data, indices, ip = csr.download()
    assert np.array_equal(ip, indptr)
    assert np.array_equal(indices, cols)
    assert np.array_equal(data, w)
    return csr, (w, cols, indptr)


def test_device_csr_equals_oracle_random(hip, oracle):
    rng = np.random.default_rng(12)
    quads = raster_quads(np.linspace(0.0, 1.0, 41), np.concatenate(([0.0], np.cumsum(rng.uniform(0.5, 2, 33)))) / 40)
    meshes = [meshgen.triangle_mesh(2500, 0), meshgen.triangle_mesh(1800, 1, 30.0, 0.7), quads]
    for nodes, faces in meshes:
        lo, hi = nodes.min(axis=0), nodes.max(axis=0)
        span = (hi - lo).max()
        for mean_len in (0.01, 0.05, 0.4):  # shorter than a cell ... crossing a large part of the mesh (big-edge path)
            edges = random_network(rng, 1500, lo.min() - 0.1 * span, hi.max() + 0.1 * span, mean_len * span)
            csr, (w, cols, indptr) = device_vs_oracle(oracle, nodes, faces, edges)
            values = rng.normal(size=(3, edges.shape[0]))
            values[rng.random(values.shape) < 0.05] = np.nan
            long_rows = np.diff(indptr) > 32
            for method, mid in (("mean", 0), ("sum", 3), ("maximum", 5), ("mode", 6), ("max_overlap", 9)):
                got = csr.apply(values, mid)
                exp = oracle.regrid_csr(method, values, w, cols, indptr, faces.shape[0])
                assert same_or_nan(got, exp)[:, ~long_rows].all(), method
                np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-13, equal_nan=True)


def test_axis_aligned_and_degenerate_edges(hip, oracle):
    """Edges on cell boundaries (both neighbours hold them), through cell corners (touches dropped), of zero length,
    outside the mesh, with NaN coordinates; empty inputs."""
    nodes, faces = raster_quads(np.arange(9.0), np.arange(7.0))
    edges = np.array([
        [[0.5, 0.5], [0.5, 0.5]],
        [[-2.0, -2.0], [-1.0, -1.0]],
        [[1.0, 0.0], [1.0, 6.0]],
        [[0.0, 3.0], [8.0, 3.0]],
        [[0.0, 6.0], [6.0, 0.0]],
        [[np.nan, 0.0], [1.0, 1.0]],
        [[2.5, 2.5], [7.5, 2.5]],
        [[8.0, 0.0], [8.0, 6.0]],   # on the outer boundary
        [[-1.0, 2.0], [9.0, 2.0]],  # sticks out on both sides, along a grid line
    ])
    csr, (w, cols, indptr) = device_vs_oracle(oracle, nodes, faces, edges)
    per_edge = np.bincount(cols, weights=w, minlength=edges.shape[0])
    assert per_edge[0] == 0 and per_edge[1] == 0 and per_edge[5] == 0
    np.testing.assert_allclose(per_edge[[2, 3, 8]], [12.0, 16.0, 16.0])  # both neighbours of the line count it
    np.testing.assert_allclose(per_edge[4], 6.0 * np.sqrt(2))
    np.testing.assert_allclose(per_edge[[6, 7]], [5.0, 6.0])
    empty = engine.edge_length_csr(engine.DeviceMesh(nodes, faces), np.zeros((0, 2, 2)))
    assert (empty.n, empty.m, empty.nnz) == (faces.shape[0], 0, 0)
    assert np.isnan(empty.apply(np.zeros((1, 0)))).all()
    with pytest.raises(ValueError):
        engine.edge_length_csr(engine.DeviceMesh(nodes, faces), np.zeros((3, 2)))


def test_many_edges_through_few_faces(hip, oracle):
    """Rows far longer than a thread can sort: 15000 edges through four coarse faces (LDS bitonic and ranked
    rows), still ordered by edge id and identical to the oracle."""
    nodes, faces = raster_quads(np.array([0.0, 1.0, 2.0]), np.array([0.0, 1.0, 2.0]))
    rng = np.random.default_rng(0)
    edges = random_network(rng, 15000, 0.05, 1.95, 0.6)
    edges[:300] = random_network(rng, 300, 0.05, 0.45, 0.1)  # some stay inside face 0
    csr, (w, cols, indptr) = device_vs_oracle(oracle, nodes, faces, edges)
    assert np.diff(indptr).max() > 4096 > np.diff(indptr).min() - 10**9
    values = rng.normal(size=(2, 15000))
    got = csr.apply(values, 0)
    exp = oracle.regrid_csr("mean", values, w, cols, indptr, 4)
    np.testing.assert_allclose(got, exp, rtol=1e-12)


def test_every_edge_kernel_path_equals_oracle(hip, oracle, xr_option):
    """The walk emits (edge, record) candidates into a flat queue -- a wave's 64 edges share an LDS stage of 1024 entries --
    and one thread per candidate clips.  A small stage sends the edges of the waves it does not hold to the wave-per-edge
    kernel, a tiny big-box threshold nearly all edges, a huge one none (a long edge then fills its wave's stage); a queue far
    too short is regrown from what the cursors counted; with and without the tile sort of the edges: the CSR is the oracle's each time.  Edge lengths from far below a cell
    to a third of the mesh, so that the candidate lists range from 1 to hundreds."""
    rng = np.random.default_rng(31)
    nodes, faces = meshgen.triangle_mesh(4000, 4)
    lo, hi = nodes.min(), nodes.max()
    span = hi - lo
    edges = np.concatenate([random_network(rng, 3000, lo, hi, 0.01 * span), random_network(rng, 2000, lo, hi, 0.06 * span),
                            random_network(rng, 400, lo, hi, 0.3 * span)])
    edges = edges[rng.permutation(edges.shape[0])]
    settings = [{}, {"edge_stage": 64}, {"edge_stage": 300, "edge_big": 8}, {"edge_big": 100000}, {"edge_big": 8},
                {"edge_queue": 1000}, {"edge_queue": 20000, "edge_stage": 128}, {"edge_stage": 1}, {"edge_sort": 0},
                {"edge_sort": 0, "edge_stage": 100}]
    for options in settings:
        for k in ("edge_stage", "edge_big", "edge_queue", "edge_sort"):
            xr_option(k, None)
        for k, v in options.items():
            xr_option(k, v)
        device_vs_oracle(oracle, nodes, faces, edges)


def test_more_pieces_than_the_fill_guessed(hip, oracle):
    """The CSR fill is enqueued into arrays sized by a guess (five pieces per edge) before the host knows nnz; a matrix that does
    not fit is filled again into arrays of its real size: 12 000 edges across half of an 80 x 80 raster, ~50 pieces each."""
    nodes, faces = raster_quads(np.linspace(0.0, 1.0, 81), np.linspace(0.0, 1.0, 81))
    rng = np.random.default_rng(5)
    edges = random_network(rng, 12_000, 0.0, 1.0, 0.5)
    csr, (w, cols, indptr) = device_vs_oracle(oracle, nodes, faces, edges)
    assert csr.nnz > 5 * edges.shape[0] + (1 << 16), "the case no longer exceeds the guess"


def _sample(grid_values, shape, x_loc, y_loc, y_descending):
    ny, nx = shape
    i = np.floor(x_loc).astype(int)
    j = np.floor(y_loc).astype(int)
    if y_descending:
        j = ny - 1 - j
    return grid_values.reshape(grid_values.shape[:-2] + (ny * nx,))[..., j * nx + i] if grid_values.ndim >= 2 else None


def test_network_gridder_reference_known_answers(hip):
    """test_network_gridder.py:75-184: structured and unstructured targets, static and transient data."""
    raster, node_xy, edge_nodes, data, (x_loc, y_loc, expected) = reference_case()
    network = xa.Ugrid1d(*node_xy.T, -1, edge_nodes)
    target_raster = xa.Raster(**raster)
    target_mesh = xa.regrid.StructuredGrid2d(target_raster).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    for target, shape in ((target_raster, (4, 4)), (target_mesh, (16,))):
        gridder = xa.NetworkGridder(network, target, method="mean")
        w = gridder._ensure_host_weights()
        assert (w.n, w.m, w.nnz) == (16, 4, 8)
        gridded = gridder.regrid(data)
        assert gridded.shape == shape and np.isnan(gridded).sum() == 11
        np.testing.assert_allclose(_sample(gridded.reshape(4, 4), (4, 4), x_loc, y_loc, True), expected)
        transient = np.stack([data, 2.0 * data])
        gridded_t = gridder.regrid(transient)
        assert gridded_t.shape == (2,) + shape and np.isnan(gridded_t).sum() == 22
        sampled = _sample(gridded_t.reshape(2, 4, 4), (4, 4), x_loc, y_loc, True)
        np.testing.assert_allclose(sampled[0], expected)
        np.testing.assert_allclose(sampled[1], 2 * expected)
        # cached weights: to_dataset -> from_weights / from_dataset reproduce the result exactly
        again = xa.NetworkGridder.from_weights(gridder.weights, target, method="mean")
        assert np.array_equal(again.regrid(data), gridded, equal_nan=True)
        again = xa.NetworkGridder.from_dataset(gridder.to_dataset())
        assert np.array_equal(again.regrid(data), gridded, equal_nan=True)
        assert xa.NetworkGridder(network, target, method="maximum").regrid(data).reshape(-1)[[9, 10]].tolist() == [2.0, 4.0]  # (y descending)
    with pytest.raises(TypeError):
        xa.NetworkGridder(target_mesh, target_mesh)
    with pytest.raises(ValueError):
        xa.NetworkGridder(network, target_mesh, method="conductance")
    with pytest.raises(ValueError):
        gridder.regrid(np.zeros(5))


def test_network_gridder_large(hip, oracle):
    """A 200k-edge network over a ~250k-triangle mesh: conservation (every edge inside the convex domain is tiled by
    its pieces) and equality with the oracle on the whole matrix."""
    nodes, faces = meshgen.triangle_mesh(125_000, 0)
    lo, hi = nodes.min(axis=0), nodes.max(axis=0)
    span = (hi - lo).max()
    rng = np.random.default_rng(4)
    edges = random_network(rng, 200_000, lo.max() + 0.2 * span, hi.min() - 0.2 * span, 0.004 * span)
    csr, (w, cols, indptr) = device_vs_oracle(oracle, nodes, faces, edges)
    per_edge = np.bincount(cols, weights=w, minlength=edges.shape[0])
    full = np.hypot(*(edges[:, 1] - edges[:, 0]).T)
    inside = ((edges > lo + 0.05 * span) & (edges < hi - 0.05 * span)).all(axis=(1, 2))
    np.testing.assert_allclose(per_edge[inside], full[inside], rtol=1e-9)


def test_network_gridder_pipelined_upload(hip, oracle):
    """From 20 MB of edge coordinates on the upload goes in pieces and the count pass of what has arrived runs on the side stream
    beside the DMA of the rest (xr_edges.hip: edge_length_csr; launches of 16 MB = 524288 edges): 700k edges = one such launch and a
    remainder, the same matrix as the oracle's, bit for bit."""
    nodes, faces = meshgen.triangle_mesh(60_000, 3)
    lo, hi = nodes.min(axis=0), nodes.max(axis=0)
    span = (hi - lo).max()
    rng = np.random.default_rng(9)
    edges = random_network(rng, 700_000, lo.min() - 0.02 * span, hi.max() + 0.02 * span, 0.006 * span)
    assert edges.nbytes >= (20 << 20)
    device_vs_oracle(oracle, nodes, faces, edges)
    # ... and with the end points already in HBM (xr_edge_length_csr_dev, round 6): no upload, one count launch, the same matrix
    from xugrid_amd import engine

    mesh = engine.DeviceMesh(nodes, faces, -1)
    host = engine.edge_length_csr(mesh, edges).download()
    dev = engine.edge_length_csr(mesh, engine.DeviceArray.from_host(edges)).download()
    assert all(np.array_equal(a, b) for a, b in zip(host, dev))


def test_celltree_intersect_edges_adapter(hip, oracle):
    """CellTree2d.intersect_edges (numba_celltree's call shape, unstructured.py:203-215): edge ids, face ids and
    the end points of every piece equal the oracle's, bit for bit, ordered by (edge, face)."""
    rng = np.random.default_rng(8)
    for nodes, faces in (meshgen.triangle_mesh(3000, 3), raster_quads(np.linspace(0, 1, 31), np.linspace(0, 1, 27))):
        edges = random_network(rng, 2500, -0.05, 1.05, 0.08)
        edges[:50, 1, 0] = edges[:50, 0, 0]  # some vertical ones
        tree = xa.CellTree2d(nodes, faces, -1)
        e, f, xy = tree.intersect_edges(edges)
        oe, of, oxy = oracle.CellTree2d(nodes, faces).intersect_edges(edges)
        assert np.array_equal(e, oe) and np.array_equal(f, of)
        assert np.array_equal(xy, oxy)
        # the reference's own post-processing of the triple gives the gridder's weights
        length = np.linalg.norm(np.diff(xy, axis=1)[:, 0, :], axis=-1)
        csr = engine.edge_length_csr(tree.device_mesh, edges)
        data, cols, indptr = csr.download()
        order = np.lexsort((e, f))
        assert np.array_equal(cols, e[order]) and np.allclose(data, length[order], rtol=1e-15)
    empty = xa.CellTree2d(nodes, faces, -1).intersect_edges(np.zeros((0, 2, 2)))
    assert empty[0].size == 0 and empty[2].shape == (0, 2, 2)


def test_reference_line_selection_known_answers(hip, oracle):
    """The numba_celltree results behind the reference's intersect_line / intersect_linestring / sel(x=slice, y=c) tests
    (tests/test_ugrid2d.py:1120-1190, tests/test_ugrid_dataset.py:255-281) from the device path: faces crossed,
    piece midpoints and distance along the line; pieces bit-equal to the oracle's."""
    nodes, faces, cases = line_selection_cases()
    tree = xa.CellTree2d(nodes, faces, -1)
    for segments, exp_faces, exp_x, exp_y, exp_s in cases:
        for seg, ef, es in ((segments, exp_faces, np.asarray(exp_s)),
                            (segments[::-1, ::-1], exp_faces[::-1], None)):
            e, f, xy = tree.intersect_edges(seg)
            oe, of, oxy = oracle.CellTree2d(nodes, faces).intersect_edges(seg)
            assert np.array_equal(e, oe) and np.array_equal(f, of) and np.array_equal(xy, oxy)
            got_f, got_x, got_y, got_s = line_selection_of_pairs(seg, e, f, xy)
            assert np.array_equal(got_f, ef)
            if es is not None:
                np.testing.assert_allclose(got_x, exp_x, rtol=0, atol=1e-15)
                np.testing.assert_allclose(got_y, exp_y, rtol=0, atol=1e-15)
                np.testing.assert_allclose(got_s, es, rtol=1e-15)


def test_reference_burn_lines_known_answer(hip):
    """tests/test_burn.py:135-141 through the device path (CellTree2d.intersect_edges)."""
    nodes, faces, segments, values, expected = burn_lines_case()
    e, f, _ = xa.CellTree2d(nodes, faces, -1).intersect_edges(segments)
    out = np.full(faces.shape[0], -1.0)
    out[f] = values[e]
    assert np.array_equal(out, expected)


def test_intersection_length_relative_reproduces_reference_formula(hip):
    """unstructured.py:213-214: ``length /= other.length[source_index]`` with source_index = FACE ids (quirk kept)."""
    import xugrid_amd as xa
    from xugrid_amd.regrid.network import Network1d
    from xugrid_amd.regrid.unstructured import UnstructuredGrid2d

    xy, faces = meshgen.quad_mesh(np.arange(0.0, 5.0), np.arange(0.0, 3.0))  # 8 faces
    grid = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)
    nodes = np.array([[0.2, 0.5], [3.7, 0.5], [3.7, 1.6], [0.3, 1.9], [2.2, 0.1], [2.4, 1.9], [0.1, 0.1], [3.9, 1.9],
                      [1.0, 1.0]])
    edges = np.array([[0, 1], [1, 2], [2, 3], [4, 5], [6, 7], [3, 8], [8, 4], [0, 8]])  # 8 edges >= max face id + 1
    net = Network1d(xa.Ugrid1d(nodes[:, 0], nodes[:, 1], -1, edges))
    g = UnstructuredGrid2d(grid)
    s_abs, t_abs, l_abs = g.intersection_length(net, relative=False)
    s_rel, t_rel, l_rel = g.intersection_length(net, relative=True)
    assert np.array_equal(s_abs, s_rel) and np.array_equal(t_abs, t_rel)
    assert np.array_equal(l_rel, l_abs / net.length[t_abs])
    # fewer edges than faces: the reference's indexing runs out of range, and so does this
    small = Network1d(xa.Ugrid1d(nodes[:, 0], nodes[:, 1], -1, edges[:3]))
    with pytest.raises(IndexError):
        g.intersection_length(small, relative=True)
