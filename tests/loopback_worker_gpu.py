"""Worker of tests/test_gpu_sharded_loopback.py: W "ranks" = W threads of this process, each running the REAL
``ShardedOverlapRegridder`` over the product's ``HipBackend`` on the one GPU of the box, with the collectives looped
back (tests/loopback_dist.py).  Unlike the one-rank RCCL run (tests/dist_worker_gpu.py) every owned target here
receives SEVERAL partial states: the HIP combine kernels see multi-sender lists, identity planes and shard-local
column ids.  Runs in a process of its own because HipBackend puts the engine on torch's stream.

    python loopback_worker_gpu.py <out_dir> small     every SHARD_METHODS reducer x {sparse, dense} x W in {2, 8}
    python loopback_worker_gpu.py <out_dir> full N    one 2N-face -> 2N-face pair at W = 8 (properties only)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from loopback_dist import run_ranks  # noqa: E402
from xugrid_amd import meshgen  # noqa: E402
from xugrid_amd.distributed import HipBackend, ShardedOverlapRegridder  # noqa: E402

ABSOLUTE = ("mean", "sum", "harmonic_mean", "geometric_mean", "minimum", "maximum")
RELATIVE = ("first_order_conservative", "conductance")


def small_data(sxy, sf):
    data7 = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(7)])
    data7[1] = np.abs(data7[1]) + 0.1      # no negatives: geometric mean defined
    data7[2, ::3] = 0.0                    # zeros (harmonic / geometric means skip them)
    data7[3] = -np.abs(data7[3]) - 0.5     # all negative
    data7[5] = np.nan                      # an all-NaN variable
    return data7


def wide_data(sxy, sf):
    data = np.concatenate([small_data(sxy, sf), np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), 40 + k, 0.02) for k in range(12)])])
    data[9] = np.abs(data[9]) + 0.3
    data[17, ::5] = 0.0
    return data


def small(out_dir):
    sxy, sf = meshgen.triangle_mesh(3000, 0)
    txy, tf = meshgen.triangle_mesh(2503, 1, 30.0, 0.7)  # T not divisible by 2 or 8
    data7 = small_data(sxy, sf)
    results = {}
    for W in (2, 8):
        for partition in ("balanced", "hash"):
            for methods in (ABSOLUTE, RELATIVE):
                def rank_body(dist, rank, methods=methods, partition=partition):
                    backend = HipBackend(0)
                    rg = ShardedOverlapRegridder(sxy, sf, txy, tf, backend, partition=partition, method=methods[0],
                                                 k_tile=2, dist=dist)
                    out = {"n_local": rg.local_faces.size, "n_local_targets": rg.local_targets.size,
                           "max_senders": int(torch.diff(rg._recv_indptr).max())}
                    for method in methods:
                        rg.set_method(method)
                        for exchange in ("sparse", "dense"):
                            rg.exchange = exchange
                            out[f"{method}_{exchange}"] = rg.regrid(data7)
                    # float32 source, K = 1, 1-D in / 1-D out
                    rg.set_method(methods[0])
                    rg.exchange = "sparse"
                    out["f32_1d"] = rg.regrid(data7[0].astype(np.float32))
                    return out

                per_rank, world = run_ranks(W, rank_body)
                tag = f"W{W}_{partition}"
                for method in methods:
                    for exchange in ("sparse", "dense"):
                        key = f"{method}_{exchange}"
                        for r in range(1, W):  # the gathered result is the same on every rank
                            assert np.array_equal(per_rank[0][key], per_rank[r][key], equal_nan=True), (tag, key, r)
                        results[f"{tag}_{key}"] = per_rank[0][key]
                results[f"{tag}_{methods[0]}_f32_1d"] = per_rank[0]["f32_1d"]
                results[f"{tag}_{methods[0]}_n_local"] = np.array([o["n_local"] for o in per_rank])
                results[f"{tag}_{methods[0]}_max_senders"] = np.array([o["max_senders"] for o in per_rank])
    # K = 19 variables in tiles of 8 (two full tiles + a short one): the K-tiled partial-state kernel, both layouts
    data19 = wide_data(sxy, sf)

    def tiled_body(dist, rank):
        rg = ShardedOverlapRegridder(sxy, sf, txy, tf, HipBackend(0), partition="balanced", method="mean", k_tile=8, dist=dist)
        out = {}
        for method in ("mean", "geometric_mean", "minimum", "harmonic_mean"):
            rg.set_method(method)
            for exchange in ("sparse", "dense"):
                rg.exchange = exchange
                out[f"{method}_{exchange}"] = rg.regrid(data19)
        rg.set_method("mean")
        rg.exchange = "sparse"
        out["f32"] = rg.regrid(data19.astype(np.float32))
        return out

    per_rank, _ = run_ranks(3, tiled_body)
    for key, value in per_rank[0].items():
        results["K19_" + key] = value
    np.savez(os.path.join(out_dir, "loopback_small.npz"), **results)


def full(out_dir, n_points):
    W = 8
    sxy, sf = meshgen.triangle_mesh(n_points, 0, delaunay=False)
    txy, tf = meshgen.triangle_mesh(n_points, 1, 30.0, 0.7, delaunay=False)
    cen = sxy[sf].mean(axis=1)
    data = np.stack([np.ones(sf.shape[0]), 2.0 * cen[:, 0] - 3.0 * cen[:, 1] + 1.0, meshgen.smooth_field(cen, 0, 0.01)])

    def rank_body(dist, rank):
        backend = HipBackend(0)
        rg = ShardedOverlapRegridder(sxy, sf, txy, tf, backend, partition="balanced", method="mean", k_tile=2, dist=dist)
        out = {"n_local": rg.local_faces.size, "n_local_targets": rg.local_targets.size,
               "nnz": rg.weights.nnz, "max_senders": int(torch.diff(rg._recv_indptr).max())}
        for exchange in ("sparse", "dense"):
            rg.exchange = exchange
            out["mean_" + exchange] = rg.regrid(data) if rank == 0 else rg.regrid(data)[:, :1]
        rg.exchange = "sparse"
        rg.set_method("maximum")
        full_max = rg.regrid(data)  # (a collective: every rank takes part)
        out["maximum"] = full_max if rank == 0 else None
        return out

    per_rank, world = run_ranks(W, rank_body)
    # the single-GPU answer on the same device
    from xugrid_amd import engine as E

    torch.cuda.synchronize()
    csr = E.DeviceMesh(sxy, sf).overlap(E.DeviceMesh(txy, tf))
    single = csr.apply(data, E.METHOD_IDS["mean"], 0.0)
    single_max = csr.apply(data, E.METHOD_IDS["maximum"], 0.0)
    np.savez(os.path.join(out_dir, "loopback_full.npz"),
             mean_sparse=per_rank[0]["mean_sparse"], mean_dense=per_rank[0]["mean_dense"], maximum=per_rank[0]["maximum"],
             single=single, single_max=single_max, nnz_single=csr.nnz,
             n_local=np.array([o["n_local"] for o in per_rank]), nnz=np.array([o["nnz"] for o in per_rank]),
             n_local_targets=np.array([o["n_local_targets"] for o in per_rank]),
             max_senders=np.array([o["max_senders"] for o in per_rank]), n_source=sf.shape[0], n_target=tf.shape[0],
             target_cx=txy[tf].mean(axis=1)[:, 0], target_cy=txy[tf].mean(axis=1)[:, 1])


if __name__ == "__main__":
    import faulthandler

    # a stuck collective (a rank that never arrives) would otherwise hang silently until the caller's timeout:
    # dump every thread's stack to stderr after XR_LOOPBACK_DUMP_S seconds (and again at every multiple)
    faulthandler.dump_traceback_later(int(os.environ.get("XR_LOOPBACK_DUMP_S", "600")), repeat=True, file=sys.stderr)
    if sys.argv[2] == "small":
        small(sys.argv[1])
    else:
        full(sys.argv[1], int(sys.argv[3]))
