"""
CPU: the N > 1 path (source-sharded OverlapRegridder, one exchange step) with world_size 2 over
gloo.  The compute backend is the oracle (tests may use it); the sharding / collective / finalise
logic under test is the product's xugrid_amd.distributed.
"""
import os
import socket
import subprocess
import sys

import numpy as np

from xugrid_amd import meshgen
from xugrid_amd.distributed import partition_faces

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_faces_balanced_and_compact():
    xy, f = meshgen.triangle_mesh(4000, 0)
    cen = xy[f].mean(axis=1)
    for world in (1, 2, 3, 8):
        for mode in ("morton", "hash"):
            owner = partition_faces(cen, world, mode)
            counts = np.bincount(owner, minlength=world)
            assert counts.sum() == f.shape[0] and counts.max() - counts.min() <= 1
    owner = partition_faces(cen, 8, "morton")
    # spatially compact shards: the bbox area of a shard is a small fraction of the domain
    areas = [np.prod(np.ptp(cen[owner == r], axis=0)) for r in range(8)]
    assert max(areas) < 0.3


def test_work_balanced_partition():
    """The benchmark pair (target = 0.7 x rotated copy inside the source square): Morton blocks of equal source
    counts leave the central ranks with several times the targets of the corner ranks; blocks of equal estimated
    work (1 per source + 4 per attributed target) even that out while staying contiguous along the curve."""
    from xugrid_amd.distributed import _targets_near_shard, work_weights

    sxy, sf = meshgen.triangle_mesh(40_000, 0, delaunay=False)
    txy, tf = meshgen.triangle_mesh(40_000, 1, 30.0, 0.7, delaunay=False)
    cen, tcen = sxy[sf].mean(axis=1), txy[tf].mean(axis=1)
    w = work_weights(cen, tcen)
    assert w.shape == (sf.shape[0],) and w.min() >= 1.0
    # every target is attributed once (the target mesh lies inside the source mesh)
    np.testing.assert_allclose(w.sum(), sf.shape[0] + 4.0 * tf.shape[0], rtol=1e-3)

    def imbalance(owner):
        cost = []
        for r in range(8):
            local = np.nonzero(owner == r)[0]
            cost.append(local.size + 4 * _targets_near_shard(sxy, sf[local], txy, tf).size)
        return max(cost) / np.mean(cost)

    plain = partition_faces(cen, 8, "morton")
    balanced = partition_faces(cen, 8, "morton", weights=w)
    assert np.bincount(balanced, minlength=8).min() > 0
    assert imbalance(plain) > 1.3 and imbalance(balanced) < 1.12
    per_rank = np.bincount(balanced, weights=w, minlength=8)
    assert per_rank.max() / per_rank.mean() < 1.01
    # uniform weights reproduce the equal-count split
    assert np.array_equal(partition_faces(cen, 8, "morton", weights=np.ones(cen.shape[0])), plain)


def test_shard_lists_rules_are_explicit():
    """shard_lists has two rules (the engine's Morton cells, the torch per-face cut) that cut the curve at different faces
    (ADVICE round 5): which one runs is explicit -- ``rule="engine"`` without an engine backend is an error, ``rule="torch"``
    ignores a backend's shard_plan, and ``"auto"`` with such a backend delegates."""
    import pytest
    import torch

    from xugrid_amd.distributed import shard_lists

    sxy, sf = meshgen.triangle_mesh(400, 0, delaunay=False)
    txy, tf = meshgen.triangle_mesh(400, 1, 30.0, 0.7, delaunay=False)
    full = tuple(torch.as_tensor(a) for a in (sxy, sf, txy, tf))

    class EngineLike:
        calls = 0

        def shard_plan(self, full, world, rank, partition):
            EngineLike.calls += 1
            return torch.arange(3), torch.arange(5)

    own = [shard_lists(full, 3, r, "balanced") for r in range(3)]
    faces = torch.cat([o[0] for o in own]).sort().values
    assert torch.equal(faces, torch.arange(sf.shape[0]))  # disjoint and complete
    with pytest.raises(ValueError):
        shard_lists(full, 3, 0, "balanced", rule="engine")
    with pytest.raises(ValueError):
        shard_lists(full, 3, 0, "balanced", rule="cells")
    a = shard_lists(full, 3, 1, "balanced", backend=EngineLike(), rule="torch")
    assert EngineLike.calls == 0 and torch.equal(a[0], own[1][0]) and torch.equal(a[1], own[1][1])
    b = shard_lists(full, 3, 1, "balanced", backend=EngineLike())
    assert EngineLike.calls == 1 and b[0].numel() == 3


def test_sharded_regridder_world2_gloo(tmp_path, oracle):
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    out = np.load(tmp_path / "dist_out.npz")
    assert int(out["world"]) == 2
    # single-process oracle result on the same inputs
    sxy, sf = meshgen.triangle_mesh(1500, 0)
    txy, tf = meshgen.triangle_mesh(1203, 1, 30.0, 0.7)
    data = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(3)])
    q, s, a = oracle.CellTree2d(sxy, sf).intersect_faces(txy, tf)
    exp = oracle.regrid_csr("mean", data, a, s, oracle.to_csr_indptr(q, tf.shape[0]), tf.shape[0])
    for mode in ("morton", "hash", "morton_dense", "balanced"):
        got = out[mode]
        assert got.shape == exp.shape
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        np.testing.assert_allclose(got, exp, rtol=1e-12, equal_nan=True)  # summation order differs across shards
        np.testing.assert_allclose(out[mode + "_1d"], exp[0], rtol=1e-12, equal_nan=True)
        if mode != "balanced":  # (equal work, not equal counts)
            assert abs(int(out[mode + "_n_local"]) - sf.shape[0] / 2) <= 1
    # the sparse all-to-all exchange and the dense reduce-scatter agree to the last bit on 2 ranks
    assert np.array_equal(out["morton"], out["morton_dense"], equal_nan=True)
    # spatially compact shards only look at the targets near them; hash shards see (almost) all
    assert int(out["morton_n_local_targets"]) < 0.8 * tf.shape[0]
    assert int(out["hash_n_local_targets"]) > 0.95 * tf.shape[0]
    # every reducer that decomposes over source shards (reduce.py:16-123, 206-222), sparse and dense exchange, the
    # variables exchanged in tiles of 2: against the single-process oracle on the unsharded matrix
    data7 = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(7)])
    data7[1] = np.abs(data7[1]) + 0.1
    data7[2, ::3] = 0.0
    data7[5] = np.nan
    indptr7 = oracle.to_csr_indptr(q, tf.shape[0])
    rel = a / oracle.area(sxy, sf)[s]
    for method in ("sum", "first_order_conservative", "harmonic_mean", "geometric_mean", "minimum", "maximum", "mean"):
        w = rel if method == "first_order_conservative" else a
        single = oracle.regrid_csr(method, data7, w, s, indptr7, tf.shape[0])
        for exchange in ("sparse", "dense"):
            got = out[f"m_{method}_{exchange}"]
            assert np.array_equal(np.isnan(got), np.isnan(single)), (method, exchange)
            rtol = 1e-9 if method == "harmonic_mean" else 1e-12  # (mixed-sign harmonic sums cancel)
            np.testing.assert_allclose(got, single, rtol=rtol, equal_nan=True, err_msg=f"{method} {exchange}")
        assert np.array_equal(out[f"m_{method}_sparse"], out[f"m_{method}_dense"], equal_nan=True), method
    assert np.isnan(out["m_mean_sparse"][5]).all()
    # integer source data are cast to float64 (as engine._source_2d), never reinterpreted
    np.testing.assert_allclose(out["int_source"], oracle.regrid_csr("mean", np.nan_to_num(10 * data).astype(np.int32).astype(np.float64),
                                                                  a, s, indptr7, tf.shape[0]), rtol=1e-12, equal_nan=True)
    # weights persisted shard by shard and reloaded without meshes: same exchange, same result
    assert np.array_equal(out["morton"], out["morton_from_file"], equal_nan=True)
    assert np.array_equal(out["morton_dense"], out["morton_from_file_dense"], equal_nan=True)
    # target-partitioned replicas (row-wise reducers): exactly the single-process result, no collective
    indptr = oracle.to_csr_indptr(q, tf.shape[0])
    for method in ("mode", "median", "max_overlap", "minimum", "sum", "mean"):
        single = oracle.regrid_csr(method, data, a, s, indptr, tf.shape[0])
        assert np.array_equal(out["tp_" + method], single, equal_nan=True), method
    assert int(out["tp_n_local_sources"]) < 0.85 * sf.shape[0]


def test_sharded_regridder_world3_and_8_loopback(oracle):
    """The same product code with W = 3 and W = 8 ranks as threads of this process and looped-back collectives
    (tests/loopback_dist.py; the GPU suite runs the HIP backend through the very same harness,
    tests/test_gpu_sharded_loopback.py).  Oracle-backed compute backend, expected values from the unsharded matrix."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import OracleBackend
    from loopback_dist import run_ranks

    from xugrid_amd.distributed import ShardedOverlapRegridder

    sxy, sf = meshgen.triangle_mesh(900, 0)
    txy, tf = meshgen.triangle_mesh(701, 1, 30.0, 0.7)
    data = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(5)])
    data[1] = np.abs(data[1]) + 0.1
    data[3] = np.nan
    q, s_, a = oracle.CellTree2d(sxy, sf).intersect_faces(txy, tf)
    T = tf.shape[0]
    indptr = oracle.to_csr_indptr(q, T)
    for W, partition in ((3, "balanced"), (8, "hash")):
        def body(dist, rank, partition=partition):
            rg = ShardedOverlapRegridder(sxy, sf, txy, tf, OracleBackend(), partition=partition, k_tile=2, dist=dist)
            out = {}
            for method in ("mean", "sum", "geometric_mean", "minimum"):
                rg.set_method(method)
                for exchange in ("sparse", "dense"):
                    rg.exchange = exchange
                    out[method, exchange] = rg.regrid(data)
            out["senders"] = int(np.diff(rg._recv_indptr.numpy()).max())
            out["n_local"] = rg.local_faces.size
            return out

        per_rank, world = run_ranks(W, body)
        assert sum(o["n_local"] for o in per_rank) == sf.shape[0]
        assert max(o["senders"] for o in per_rank) >= (W if partition == "hash" else 2)
        assert min(world.sent_bytes) > 0
        for method in ("mean", "sum", "geometric_mean", "minimum"):
            expected = oracle.regrid_csr(method, data, a, s_, indptr, T)
            for exchange in ("sparse", "dense"):
                for o in per_rank:
                    np.testing.assert_allclose(o[method, exchange], expected, rtol=1e-12, atol=1e-14, equal_nan=True)
    # ownership of the rows (round 6): "partition" -- the lowest rank whose shard touches the row -- against id chunks.  Same
    # results; every touched row has exactly one owner, who touches it; and only the boundary layer leaves a rank
    def owners(dist, rank):
        res = {}
        for ownership in ("partition", "chunk"):
            rg = ShardedOverlapRegridder(sxy, sf, txy, tf, OracleBackend(), partition="morton", k_tile=3, dist=dist, ownership=ownership)
            res[ownership] = (rg.regrid(data), rg.exchange_bytes(), rg.local_targets.copy(),
                              rg.owned_targets.numpy().copy() if ownership == "partition" else None, rg.regrid(data, gather=False).shape)
        return res

    per_rank, _ = run_ranks(4, owners)
    expected = oracle.regrid_csr("mean", data, a, s_, indptr, T)
    for o in per_rank:
        for ownership in ("partition", "chunk"):
            np.testing.assert_allclose(o[ownership][0], expected, rtol=1e-12, atol=1e-14, equal_nan=True)
    owned = [o["partition"][3] for o in per_rank]
    touched = np.unique(np.concatenate([o["partition"][2] for o in per_rank]))
    assert np.array_equal(np.sort(np.concatenate(owned)), touched)  # disjoint and complete over the touched rows
    for o, mine in zip(per_rank, owned):
        assert np.isin(mine, o["partition"][2]).all() and (np.diff(mine) > 0).all()
        assert o["partition"][4] == (data.shape[0], mine.size)
    off_partition = sum(o["partition"][1]["sent_off_gpu"] for o in per_rank)
    off_chunk = sum(o["chunk"][1]["sent_off_gpu"] for o in per_rank)
    assert off_partition < 0.35 * off_chunk, (off_partition, off_chunk)  # (a 1400-face target: the boundary layer is thick)

    # relative and absolute weights are not interchangeable; whole-row reducers are refused
    def refuse(dist, rank):
        rg = ShardedOverlapRegridder(sxy, sf, txy, tf, OracleBackend(), dist=dist)
        for bad in ("first_order_conservative", "median"):
            try:
                rg.set_method(bad)
            except ValueError:
                continue
            raise AssertionError(bad)
        return True

    assert run_ranks(2, refuse)[0] == [True, True]


def test_bench_self_launches_two_ranks_on_gloo(tmp_path):
    """`python bench.py --gpus 2` with no torch.distributed environment: bench.py re-execs itself under
    torch.distributed.run (free port, one rank per "GPU"), rank 0 prints ONE JSON line, the exit code is that of the job.
    On CPU the collectives are gloo and the compute backend is the oracle-backed stand-in of tests/dist_worker.py (a test
    hook of bench.py); what runs is bench.py's own launcher, workload set-up, timing protocol and reporting."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--points", "1500",
           "--no-delaunay", "--dist-backend", "gloo", "--compute-backend", "tests.dist_worker:OracleBackend",
           "--multi-extras", "--strong-points", "1200", "--extras-k", "3"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    cfg = line["config"]
    assert cfg["rccl_ranks"] == 2 and cfg["collective_backend"] == "gloo"
    assert len(cfg["per_rank"]["step_ms_per_rank"]["all"]) == 2
    inc = cfg["including_setup"]  # like for like with N = 1: the set-up a step from raw arrays would count
    assert inc["ms_per_step_including_setup"] > 0 and inc["value_including_setup"] > 0
    assert cfg["other_exchange"]["exchange"] == "dense"
    others = line["other_configs"]
    strong = others["config4_strong_10M"]
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["value"] > 0
    k3 = others["config5_apply_K3"]
    assert k3["config"]["variables"] == 3 and k3["unit"] == "target cell-variables/s" and k3["config"]["exchange"] == "none"
    # a failing rank makes the launcher's exit code non-zero
    bad = subprocess.run(cmd[:-6] + ["--compute-backend", "tests.dist_worker:NoSuchBackend"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert bad.returncode != 0
