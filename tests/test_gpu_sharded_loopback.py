"""
GPU (-m gpu): the source-sharded multi-GPU path with W > 1 CONTRIBUTIONS PER TARGET on the one GPU of the box.
W threads of one process run the real ``ShardedOverlapRegridder`` + ``HipBackend`` with looped-back collectives
(tests/loopback_dist.py, tests/loopback_worker_gpu.py), so ``xr_reduce_partial_rows_dev`` combines multi-sender
lists, the dense form goes through ``xr_partial_fill_identity_dev`` + plane-wise sum / max +
``xr_finalize_partial_dev``, and every shard works in shard-local column ids.  Expected values: the CPU oracle on the
UNSHARDED matrix (xugrid/regrid/reduce.py:16-123, 206-222).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from xugrid_amd import meshgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "loopback_worker_gpu.py")


def _run(args, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, WORKER] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-6000:]


def test_every_shard_reducer_with_several_contributions_per_target(hip, oracle, tmp_path):
    from loopback_worker_gpu import ABSOLUTE, RELATIVE, small_data

    _run([str(tmp_path), "small"], 1500)
    out = np.load(tmp_path / "loopback_small.npz")
    sxy, sf = meshgen.triangle_mesh(3000, 0)
    txy, tf = meshgen.triangle_mesh(2503, 1, 30.0, 0.7)
    data7 = small_data(sxy, sf)
    q, s_, a = oracle.CellTree2d(sxy, sf).intersect_faces(txy, tf)
    T = tf.shape[0]
    indptr = oracle.to_csr_indptr(q, T)
    rel = a / oracle.area(sxy, sf)[s_]
    for W in (2, 8):
        for partition in ("balanced", "hash"):
            tag = f"W{W}_{partition}"
            n_local = out[f"{tag}_mean_n_local"]
            assert n_local.size == W and n_local.sum() == sf.shape[0] and n_local.min() > 0
            # the point of the test: owned targets receive partial states from several senders
            assert out[f"{tag}_mean_max_senders"].max() >= (W if partition == "hash" else 2), (tag, out[f"{tag}_mean_max_senders"])
            for method in ABSOLUTE + RELATIVE:
                expected = oracle.regrid_csr(method, data7, rel if method in RELATIVE else a, s_, indptr, T)
                for exchange in ("sparse", "dense"):
                    got = out[f"{tag}_{method}_{exchange}"]
                    assert got.shape == (7, T)
                    assert np.array_equal(np.isnan(got), np.isnan(expected)), (tag, method, exchange)
                    if method in ("minimum", "maximum"):
                        assert np.array_equal(got, expected, equal_nan=True), (tag, method, exchange)
                    else:
                        # shards are summed in rank order, not in column order: rounding differs from the sequential
                        # loop; harmonic means of mixed-sign data cancel
                        rtol = 1e-9 if method == "harmonic_mean" else 1e-12
                        np.testing.assert_allclose(got, expected, rtol=rtol, atol=1e-14 if method != "harmonic_mean" else 0,
                                                   equal_nan=True, err_msg=f"{tag} {method} {exchange}")
                assert np.array_equal(out[f"{tag}_{method}_sparse"], out[f"{tag}_{method}_dense"], equal_nan=True), (tag, method)
            exp32 = oracle.regrid_csr("mean", data7[:1].astype(np.float32), a, s_, indptr, T)[0]
            np.testing.assert_allclose(out[f"{tag}_mean_f32_1d"], exp32, rtol=1e-12, equal_nan=True)


    # 19 variables in tiles of 8, three ranks: the K-tiled partial-state kernel (whole tiles and a short one)
    from loopback_worker_gpu import wide_data

    data19 = wide_data(sxy, sf)
    for method in ("mean", "geometric_mean", "minimum", "harmonic_mean"):
        expected = oracle.regrid_csr(method, data19, a, s_, indptr, T)
        for exchange in ("sparse", "dense"):
            got = out[f"K19_{method}_{exchange}"]
            assert np.array_equal(np.isnan(got), np.isnan(expected)), (method, exchange)
            if method == "minimum":
                assert np.array_equal(got, expected, equal_nan=True)
            else:
                np.testing.assert_allclose(got, expected, rtol=1e-9 if method == "harmonic_mean" else 1e-12,
                                           atol=1e-14 if method != "harmonic_mean" else 0, equal_nan=True,
                                           err_msg=f"K19 {method} {exchange}")
    exp32 = oracle.regrid_csr("mean", data19.astype(np.float32), a, s_, indptr, T)
    np.testing.assert_allclose(out["K19_f32"], exp32, rtol=1e-12, atol=1e-14, equal_nan=True)


def test_sharded_w8_at_10m_faces_properties(hip, tmp_path):
    """BASELINE config 4's shape (10M -> 10M triangles, 8 source shards) on one GPU: properties only."""
    _run([str(tmp_path), "full", "5000000"], 3000)
    out = np.load(tmp_path / "loopback_full.npz")
    S, T = int(out["n_source"]), int(out["n_target"])
    assert S > 9_900_000 and T > 9_900_000
    assert out["n_local"].sum() == S and out["n_local"].min() > 0.05 * S
    # shards partition the weights: every (target, source) pair lives on exactly one rank
    assert out["nnz"].sum() == int(out["nnz_single"])
    assert out["max_senders"].max() >= 2
    # balanced partition: targets per rank within 2x of each other (the target mesh covers the middle of the source)
    assert out["n_local_targets"].max() < 2.0 * out["n_local_targets"].mean()
    mean, single = out["mean_sparse"], out["single"]
    assert np.array_equal(out["mean_sparse"], out["mean_dense"], equal_nan=True)
    assert np.array_equal(np.isnan(mean), np.isnan(single))
    np.testing.assert_allclose(mean, single, rtol=1e-12, atol=1e-14, equal_nan=True)
    ok = ~np.isnan(mean[0])
    assert ok.mean() > 0.999
    assert np.abs(mean[0][ok] - 1.0).max() < 1e-14  # constants are preserved
    # a linear field: the area-weighted mean over the target face of the source-centroid values stays within the
    # field's range over the (slightly larger) neighbourhood -- here only checked against the single-GPU answer and
    # against the target centroid value to O(h)
    lin = 2.0 * out["target_cx"] - 3.0 * out["target_cy"] + 1.0
    assert np.abs(mean[1][ok] - lin[ok]).max() < 0.02
    assert np.array_equal(out["maximum"], out["single_max"], equal_nan=True)


def test_shard_plan_on_the_device(hip):
    """xr_shard_plan_dev (round 5: the set-up of a rank as HIP kernels instead of torch tensor operations) -- the checks live in
    tests/shard_plan_worker_gpu.py (a process of its own: HipBackend puts the engine on torch's stream)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shard_plan_worker_gpu.py")], env=env, capture_output=True,
                          text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-6000:]
    assert "shard plan ok" in proc.stdout
