"""
GPU (-m gpu): BASELINE.json configs 4 and 5 at their full single-GPU sizes, against the CPU oracle.

config 5  1M -> 1M Delaunay triangles, 256 stacked variables re-using cached weights (from_weights route:
          weights downloaded, uploaded again with xr_csr_upload + xr_csr_set_row_keys): the K > 192 "groups of 128"
          branch of the many-variable apply, the apply plan at 1M rows, with NaNs and with XR_APPLY_NO_PLAN.
          Reference: make_regrid(func)._regrid (xugrid/regrid/regridder.py:41-67), from_weights (:334-348).
config 4  10M -> 10M triangles on ONE GPU (the 8-GPU source-sharded run is the driver's): size-independent
          properties of the whole matrix plus a bit-exact comparison with the oracle on a 100k-row sample, and the
          bench's multi-GPU code path (`bench.py --force-dist`) at that size through a one-rank RCCL group.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import same_or_nan
from test_gpu_parity import assert_apply_equal
from xugrid_amd import meshgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stacked_field(centroids, K, seed):
    """(K, S) float64: the C2 field with a phase shift per variable."""
    rng = np.random.default_rng(seed)
    x, y = centroids[:, 0], centroids[:, 1]
    noise = 0.1 * rng.normal(size=(8, x.size))
    v = np.empty((K, x.size))
    for k in range(K):
        v[k] = np.sin(6 * np.pi * x + 0.37 * k) * np.cos(4 * np.pi * y - 0.11 * k) + noise[k % 8]
    return v


def test_config5_cached_weights_k256(hip, oracle, monkeypatch, xr_option):
    from xugrid_amd import engine as E

    K = 256
    sxy, sf = meshgen.triangle_mesh(500_000, 0)
    txy, tf = meshgen.triangle_mesh(500_000, 1, 30.0, 0.7)
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    built = ms.overlap(mt)
    data, idx, indptr = built.download()
    T, S = built.n, built.m
    assert T > 990_000 and S > 990_000 and built.nnz > 4_000_000
    # cached weights: the from_weights route (host arrays -> xr_csr_upload) + the locality hint for uploaded rows
    cached = E.DeviceCSR.from_arrays(data, idx, indptr, T, S)
    keys, key_range = E.morton_row_keys(oracle.centroids(txy, tf), faces_per_tile=64)
    cached.set_row_keys(keys, key_range)
    v = _stacked_field(oracle.centroids(sxy, sf), K, 5)
    exp = oracle.regrid_csr("mean", v, data, idx, indptr, T)
    got = cached.apply(v, 0)  # K = 256 > 192: two groups of 128 (xr_apply.hip: apply_dispatch)
    assert_apply_equal(got, exp, indptr, "mean K=256 cached")
    # the matrix xr_overlap left on the device (keys set by the build) gives the same numbers
    got_built = built.apply(v, 0)
    assert same_or_nan(got_built, got).all()
    # downloads are unchanged by the re-tiling
    d2, i2, p2 = cached.download()
    assert np.array_equal(d2, data) and np.array_equal(i2, idx) and np.array_equal(p2, indptr)
    # without the plan (direct gathers)
    xr_option("apply_plan", 0)
    got_np = cached.apply(v, 0)
    xr_option("apply_plan", None)
    assert_apply_equal(got_np, exp, indptr, "mean K=256 no plan")
    del got_np, got_built
    # NaNs: 1 % in every fourth variable, one variable all-NaN (NaN-free tiles take the short path, the others not)
    rng = np.random.default_rng(9)
    for k in range(0, K, 4):
        v[k, rng.random(S) < 0.01] = np.nan
    v[130] = np.nan
    exp = oracle.regrid_csr("mean", v, data, idx, indptr, T)
    got = cached.apply(v, 0)
    assert_apply_equal(got, exp, indptr, "mean K=256 NaN")
    assert np.isnan(got[130]).all()
    # a second reducer on the same plan, float32 sources
    v32 = v[:200].astype(np.float32)
    exp = oracle.regrid_csr("maximum", v32, data, idx, indptr, T)
    assert_apply_equal(cached.apply(v32, 5), exp, indptr, "maximum K=200 f32")


def test_config2_mesh_onto_itself_full_size(hip, oracle):
    """The 1M-triangle benchmark source regridded onto ITSELF, and onto a copy of itself shifted by one side of one of its faces
    (one node lands exactly on another): every face touches ~12 neighbours without overlapping them -- the pairs the reference
    drops before it clips (strict box test, SAT).  Pair set, order and areas bit for bit against the oracle
    at full size; the diagonal of the self-overlap is the face areas' clip with themselves."""
    from test_gpu_parity import assert_overlap_parity

    sxy, sf = meshgen.triangle_mesh(500_000, 0)
    _, (data, idx, indptr) = assert_overlap_parity(hip, oracle, sxy, sf, sxy, sf)
    assert indptr.size == sf.shape[0] + 1
    rows = np.repeat(np.arange(sf.shape[0]), np.diff(indptr))
    assert (np.bincount(rows[idx == rows], minlength=sf.shape[0]) == 1).all()  # every face overlaps itself
    f = sf[sf.shape[0] // 2]
    assert_overlap_parity(hip, oracle, sxy, sf, sxy + (sxy[f[1]] - sxy[f[0]]), sf)  # a copy shifted by one side of one face


def test_config4_10m_single_gpu(hip, oracle):
    from xugrid_amd import engine as E

    n_points = 5_000_000
    sxy, sf = meshgen.triangle_mesh(n_points, 0, delaunay=False)
    txy, tf = meshgen.triangle_mesh(n_points, 1, 30.0, 0.7, delaunay=False)
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    csr = ms.overlap(mt)
    data, idx, indptr = csr.download()
    T, S = csr.n, csr.m
    assert T == tf.shape[0] and S == sf.shape[0] and T > 9_900_000
    assert data.size == csr.nnz == indptr[-1] and csr.nnz > 40_000_000
    assert (data > 0).all() and idx.min() >= 0 and idx.max() < S
    counts = np.diff(indptr)
    assert (counts > 0).all()  # the target lies inside the source hull
    inner = np.ones(idx.size, dtype=bool)
    inner[indptr[:-1]] = False
    assert (np.diff(idx)[inner[1:]] > 0).all()  # rows strictly ascending in the source index
    t_area = mt.area()
    row_sum = np.add.reduceat(data, indptr[:-1])
    # (a handful of the 10M jittered-lattice faces are near-degenerate slivers: absolute floor of 1e-12 mean areas)
    np.testing.assert_allclose(row_sum, t_area, rtol=1e-9, atol=1e-12 * t_area.mean())
    col_sum = np.bincount(idx, weights=data, minlength=S)
    assert (col_sum <= ms.area() * (1 + 1e-9)).all()
    assert abs(data.sum() / t_area.sum() - 1) < 1e-10
    one = csr.apply(np.full((1, S), 3.25))
    assert np.allclose(one, 3.25, rtol=1e-14)
    # oracle on a 100k-row sample of the target faces (the tree is built over all 10M source faces)
    rng = np.random.default_rng(4)
    sample = np.sort(rng.choice(T, 100_000, replace=False))
    tree = oracle.CellTree2d(sxy, sf)
    oq, os_, oa = tree.intersect_faces(txy, tf[sample])
    o_indptr = oracle.to_csr_indptr(oq, sample.size)
    assert np.array_equal(np.diff(o_indptr), counts[sample]), "row lengths differ from the oracle"
    cnt = counts[sample]
    flat = np.repeat(indptr[sample] - (np.cumsum(cnt) - cnt), cnt) + np.arange(cnt.sum())  # entries of the sampled rows
    assert np.array_equal(idx[flat], os_), "source indices differ from the oracle"
    assert np.array_equal(data[flat], oa), "areas not bit-exact"
    v = meshgen.smooth_field(oracle.centroids(sxy, sf), 0, nan_fraction=0.01)[None, :]
    got = csr.apply(v, 0)
    exp = oracle.regrid_csr("mean", v, oa, os_, o_indptr, sample.size)
    assert_apply_equal(got[:, sample], exp, o_indptr, "mean 10M sample")


def test_bench_multi_gpu_path_10m_one_rank():
    """The SCALE path of bench.py at config 4's size through a one-rank RCCL group (the box has one GPU)."""
    # (no torch.distributed variables in the environment: bench.py launches itself under torch.distributed.run, the way the
    # driver's plain `python bench.py --gpus N` would)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for exchange in ("dense", "sparse"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--points", "5000000", "--steps", "3",
               "--warmup", "1", "--exchange", exchange]
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
        assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
        line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])  # (RCCL prints banners)
        assert line["n_gpus"] == 1 and line["config"]["target_faces"] > 9_900_000
        assert line["value"] > 1e8 and line["config"]["nnz"] > 40_000_000
        _check_scale_fields(line, exchange, 1)
        inc = line["config"]["including_setup"]  # the set-up N = 1 counts, from HBM-resident meshes
        assert inc["ms_per_step_including_setup"] > line["ms_per_step"] and inc["value_including_setup"] > 1e7
    # the same through an explicit environment (how torch.distributed.run starts the ranks)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the other workloads of the SCALE line: config 4 as a strong-scaling pair; config 5 (K = 256) on cached weights --
    # target rows partitioned (default: no collective) and source-sharded (partial states exchanged in 8 tiles of 32)
    runs = ((["--strong", "--strong-points", "5000000"], "strong", "sparse"),
            (["--k", "256", "--points", "500000", "--no-delaunay"], "weak", "none"),
            (["--k", "256", "--k-mode", "source", "--points", "500000", "--no-delaunay"], "weak", "sparse"))
    for extra, scaling, exchange in runs:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1"] + extra
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
        assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
        line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["scaling"] == scaling and line["n_gpus"] == 1
        _check_scale_fields(line, exchange, 1)
        if "--k" in extra:
            assert line["config"]["variables"] == 256 and line["unit"] == "target cell-variables/s"
            if exchange == "none":
                assert line["config"]["per_rank"]["collectives_per_step"] == 0 and line["value"] > 5e10
            else:
                # 256 variables in 8 tiles of 32 -- and a group of ONE rank makes no collective call for any of them
                assert line["config"]["per_rank"]["collectives_per_step"] == 0
                assert line["value"] > 2e9
        else:
            assert line["config"]["target_faces"] > 9_900_000 and "config 4" in line["config"]["workload"]


def test_bench_one_invocation_three_multi_gpu_numbers():
    """`bench.py --gpus N` (N > 1 by default; one rank here with --multi-extras) carries the config-4 strong and the
    config-5 K-variable workloads as `other_configs` of the ONE JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--multi-extras", "--steps", "3",
           "--warmup", "1", "--points", "200000", "--no-delaunay", "--strong-points", "400000", "--extras-k", "32"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    line = json.loads(lines[-1])
    _check_scale_fields(line, "sparse", 1)
    strong = line["other_configs"]["config4_strong_10M"]
    assert strong["scaling"] == "strong" and strong["config"]["target_faces"] > 700_000 and strong["value"] > 1e8
    k32 = line["other_configs"]["config5_apply_K32"]
    assert k32["config"]["variables"] == 32 and k32["config"]["exchange"] == "none" and k32["value"] > 1e10


def test_bench_projection_of_eight_shards():
    """`bench.py --project-shards 8`: the eight shards of the multi-GPU workload one after another on the one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for extra in ([], ["--strong", "--strong-points", "800000"]):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--project-shards", "8", "--steps", "3", "--warmup", "1",
               "--points", "100000", "--no-delaunay"] + extra
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
        out = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["shards"] == 8 and len(out["per_shard"]) == 8 and "PROJECTED" in out["what"]
        assert sum(r["source_faces"] for r in out["per_shard"]) == out["source_faces"]  # disjoint and complete
        assert all(r["nnz"] > 0 and r["compute_ms"] > 0 for r in out["per_shard"])
        # (three timed steps per shard on a tiny mesh: the figures are measurements -- one host hiccup moves them by tens of per
        # cent, seen once in a full-suite run -- so only their sanity is asserted, not their size)
        assert 1.0 <= out["imbalance_max_over_mean"] < 10.0
        assert out["projected_step_ms"] > 0 and 0 < out["projected_efficiency_1_to_W"] < 10.0


def _check_scale_fields(line, exchange, world):
    """What somebody reading SCALE_rNN.json needs to interpret a line: the exchange time, the bytes a rank sends, the
    spread of the ranks' step times, the size of the RCCL group."""
    cfg = line["config"]
    assert cfg["exchange"] == exchange and cfg["rccl_ranks"] == world and cfg["collective_backend"] == "nccl"
    if exchange == "none" or world == 1:
        assert cfg["exchange_ms"] == 0.0  # (a group of one rank makes no collective call: every state already is where it is combined)
    else:
        assert 0.0 < cfg["exchange_ms"] < line["ms_per_step"]
    pr = cfg["per_rank"]
    assert len(pr["step_ms_per_rank"]["all"]) == world
    assert 0 < pr["step_ms_per_rank"]["min"] <= pr["step_ms_per_rank"]["max"] <= line["ms_per_step"] * 1.001
    assert pr["exchange_bytes_sent_per_rank"]["max"] >= 0  # (one rank: everything stays on the GPU)
    assert pr["source_faces_per_rank"]["min"] > 0 and pr["target_faces_per_rank"]["min"] > 0
    assert cfg["other_exchange"]["exchange_ms"] > 0 or exchange == "none" or world == 1
    assert line["roofline"]["frac"] > 0


def test_config3_barycentric_4m_targets(hip, oracle):
    """BASELINE config 3 at its full size: 1M-triangle source, 4M target faces (their centroids are the 4M query
    points, unstructured.py:147).  Size-independent properties of the whole weight matrix, and the oracle's
    step-by-step restatement on a 150k-point sample of the same points: identical triplets."""
    import xugrid_amd as xa
    from stepwise import oracle_barycentric_triplets

    sxy, sf = meshgen.triangle_mesh(500_000, 0, delaunay=False)
    txy, tf = meshgen.triangle_mesh(2_000_000, 2, 30.0, 0.7, delaunay=False)
    src = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    dcsr = xa.regrid.UnstructuredGrid2d(src).barycentric_device(xa.regrid.UnstructuredGrid2d(tgt))
    data, indices, indptr = dcsr.download()
    n = tf.shape[0]
    assert dcsr.n == n > 3_900_000 and dcsr.m == sf.shape[0] and data.size == dcsr.nnz > 20_000_000
    counts = np.diff(indptr)
    assert (counts >= 1).all() and counts.max() <= 32  # every target centroid lies inside the source mesh
    assert (data > 0).all() and indices.min() >= 0 and indices.max() < dcsr.m
    # barycentric weights sum to one -- except in the concave exterior Voronoi cells, where the generalised weights have
    # negative components that the reference drops (unstructured.py:191-198: weights > 0 only; `mean` renormalises)
    sums = np.add.reduceat(data, indptr[:-1])
    assert (np.abs(sums - 1.0) < 1e-12).mean() > 0.999 and sums.min() > 0
    # linear fields are reproduced at interior points (barycentric interpolation is exact for them): v = 2x - 3y + 1
    cen_s = oracle.centroids(sxy, sf)
    cen_t = oracle.centroids(txy, tf)
    lin = dcsr.apply((2.0 * cen_s[:, 0] - 3.0 * cen_s[:, 1] + 1.0)[None, :], 0)[0]
    interior = counts >= 3
    err = np.abs(lin - (2.0 * cen_t[:, 0] - 3.0 * cen_t[:, 1] + 1.0))[interior]
    assert np.percentile(err, 99.9) < 1e-9
    # oracle on a sample of the query points
    rng = np.random.default_rng(3)
    sample = np.sort(rng.choice(n, 150_000, replace=False))
    os_, ot, ow = oracle_barycentric_triplets(oracle, src, cen_t[sample])
    cnt = counts[sample]
    flat = np.repeat(indptr[sample] - (np.cumsum(cnt) - cnt), cnt) + np.arange(cnt.sum())
    assert np.array_equal(np.bincount(ot, minlength=sample.size), cnt), "row lengths differ from the oracle"
    assert np.array_equal(indices[flat], os_) and np.array_equal(data[flat], ow)


def test_config3_barycentric_delaunay_source_as_benchmarked(hip, oracle):
    """BASELINE config 3 on the workload bench.py TIMES: a 1M-triangle DELAUNAY source (node degree up to ~17, hull slivers,
    concave exterior Voronoi cells) and the 4M centroids of a Delaunay target.  The other full-size tests use lattice-split
    meshes (degree <= 8).  Properties of the whole matrix; the oracle's step-by-step restatement (unstructured.py:146-201) on
    a 150k-point sample of the benchmark's points; and -- because the benchmark's target lies INSIDE the source hull and so
    never meets the exterior cells -- on 60k extra points in and around the source's boundary layer."""
    import xugrid_amd as xa
    from stepwise import oracle_barycentric_triplets
    from xugrid_amd import engine

    sxy, sf = meshgen.triangle_mesh(500_000, 0, delaunay=True)
    txy, tf = meshgen.triangle_mesh(2_000_000, 2, 30.0, 0.7, delaunay=True)
    src = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    us = xa.regrid.UnstructuredGrid2d(src)
    dcsr = us.barycentric_device(xa.regrid.UnstructuredGrid2d(tgt))
    data, indices, indptr = dcsr.download()
    n = tf.shape[0]
    assert dcsr.n == n > 3_900_000 and dcsr.m == sf.shape[0] and data.size == dcsr.nnz > 20_000_000
    counts = np.diff(indptr)
    assert (counts >= 1).all()  # every target centroid lies inside the source mesh
    assert (data > 0).all() and indices.min() >= 0 and indices.max() < dcsr.m
    sums = np.add.reduceat(data, indptr[:-1])
    assert (np.abs(sums - 1.0) < 1e-12).mean() > 0.999 and sums.min() > 0
    cen_t = oracle.centroids(txy, tf)
    rng = np.random.default_rng(5)
    sample = np.sort(rng.choice(n, 150_000, replace=False))
    os_, ot, ow = oracle_barycentric_triplets(oracle, src, cen_t[sample])
    cnt = counts[sample]
    flat = np.repeat(indptr[sample] - (np.cumsum(cnt) - cnt), cnt) + np.arange(cnt.sum())
    assert np.array_equal(np.bincount(ot, minlength=sample.size), cnt), "row lengths differ from the oracle"
    assert np.array_equal(indices[flat], os_) and np.array_equal(data[flat], ow)
    del data, indices, indptr, dcsr
    # the boundary layer: points within ~3 cells of the unit square's sides, half of them outside the hull
    h = 1.0 / np.sqrt(500_000)
    side = rng.integers(0, 4, 60_000)
    along = rng.uniform(-2 * h, 1 + 2 * h, 60_000)
    across = rng.uniform(-3 * h, 3 * h, 60_000)
    px = np.where(side == 0, along, np.where(side == 1, along, np.where(side == 2, across, 1 - across)))
    py = np.where(side == 0, across, np.where(side == 1, 1 - across, np.where(side == 2, along, along)))
    pts = np.ascontiguousarray(np.column_stack([px, py]))
    voronoi_mesh, face_index_tail, n2n = us._voronoi_device()
    c2 = engine.barycentric_csr(voronoi_mesh, src.device_mesh, face_index_tail, n2n, points=pts, n_identity=src.n_face)
    d2, i2, p2 = c2.download()
    os2, ot2, ow2 = oracle_barycentric_triplets(oracle, src, pts)
    rows2 = np.repeat(np.arange(c2.n), np.diff(p2))
    empty = np.diff(p2) == 0
    assert 0.2 < empty.mean() < 0.8  # a good share outside, a good share inside exterior cells
    assert np.array_equal(i2, os2) and np.array_equal(rows2, ot2) and np.array_equal(d2, ow2)


def test_full_size_mixed_and_raster_pairs_bit_exact(hip, oracle):
    """The one-round-trip pipeline for non-triangle pairs at full size (round 5): a ~1M-face mixed triangle / quadrilateral mesh
    onto a rotated one, and the 1M-triangle benchmark source onto a 1000 x 1000 raster (BASELINE config 1's shape at scale:
    quadrilateral targets) -- pair sets and areas equal to the oracle's bit for bit, row sums = target areas."""
    from xugrid_amd import engine as E

    sxy, sf = meshgen.mixed_mesh(660_000, 0)
    txy, tf = meshgen.mixed_mesh(660_000, 1, 30.0, 0.7)
    assert 950_000 < sf.shape[0] < 1_050_000
    cases = [(sxy, sf, txy, tf)]
    bxy, bf = meshgen.triangle_mesh(500_000, 0, delaunay=False)
    edges = np.linspace(0.02, 0.98, 1001)
    rxy, rf = meshgen.quad_mesh(edges, edges)
    cases.append((bxy, bf, rxy, rf))
    for s_xy, s_f, q_xy, q_f in cases:
        ms, mq = E.DeviceMesh(s_xy, s_f, -1), E.DeviceMesh(q_xy, q_f, -1)
        csr = ms.overlap(mq)
        data, idx, indptr = csr.download()
        oq, os_, oa = oracle.CellTree2d(s_xy, s_f, -1).intersect_faces(q_xy, q_f, -1)
        assert np.array_equal(np.repeat(np.arange(csr.n), np.diff(indptr)), oq)
        assert np.array_equal(idx, os_) and np.array_equal(data, oa)
        np.testing.assert_allclose(np.bincount(oq, weights=data, minlength=csr.n), mq.area(), rtol=1e-9)
