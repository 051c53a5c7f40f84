"""Worker of the GPU distributed test: the product's HipBackend (C ABI through torch device pointers) inside a
torch.distributed process group over RCCL ("nccl").  The GPU box has one device, so the group has one rank; the
collectives still run through RCCL.  Results go to an .npz that the test compares with the oracle."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (initialised before the engine binds the device)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from xugrid_amd import meshgen  # noqa: E402
from xugrid_amd.distributed import (  # noqa: E402
    HipBackend,
    ShardedOverlapRegridder,
    TargetPartitionedRegridder,
    init_process_group_from_env,
)


def main():
    out_dir = sys.argv[1]
    init_process_group_from_env("nccl")
    rank = dist.get_rank()
    backend = HipBackend(int(os.environ.get("LOCAL_RANK", rank)))
    sxy, sf = meshgen.triangle_mesh(3000, 0)
    txy, tf = meshgen.triangle_mesh(2501, 1, 30.0, 0.7)
    data = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(3)])
    results = {}
    for exchange in ("sparse", "dense"):
        # (always_exchange: a one-rank group would otherwise skip the collective -- here the RCCL calls are the point)
        rg = ShardedOverlapRegridder(sxy, sf, txy, tf, backend, exchange=exchange, always_exchange=True)
        results["mean_" + exchange] = rg.regrid(data)
        rg.rebuild()
        results["mean_rebuilt_" + exchange] = rg.regrid(data.astype(np.float32))
        # weight build + partial states as one engine call (xr_overlap_partial_dev): one variable rides on the build, three do not
        for k in (1, 3):
            results[f"mean_fused{k}_" + exchange] = rg.rebuild_regrid_local(rg.local_source(data[:k])).cpu().numpy()[:, : rg.n_target]
    rg.to_file(os.path.join(out_dir, "sharded"))  # (dense exchange stored; read back as sparse)
    dist.barrier()
    results["mean_from_file"] = ShardedOverlapRegridder.from_file(
        os.path.join(out_dir, "sharded"), backend, exchange="sparse").regrid(data)
    # every shard-decomposable reducer through the HIP partial-state kernels, K exchanged in tiles of 2
    data7 = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(7)])
    data7[1] = np.abs(data7[1]) + 0.1
    data7[2, ::3] = 0.0
    data7[5] = np.nan
    for method in ("sum", "first_order_conservative", "harmonic_mean", "geometric_mean", "minimum", "maximum"):
        for exchange in ("sparse", "dense"):
            rg = ShardedOverlapRegridder(sxy, sf, txy, tf, backend, exchange=exchange, method=method, k_tile=2)
            results[f"m_{method}_{exchange}"] = rg.regrid(data7)
            # one variable at a time: the wave-window partial-state kernel, one specialisation per reducer
            results[f"m1_{method}_{exchange}"] = np.stack([rg.regrid(data7[k]) for k in range(data7.shape[0])])
    results["int_source"] = ShardedOverlapRegridder(sxy, sf, txy, tf, backend).regrid(np.nan_to_num(10 * data).astype(np.int32))
    for method in ("mode", "median", "max_overlap", "minimum"):
        results["tp_" + method] = TargetPartitionedRegridder(sxy, sf, txy, tf, backend, method=method).regrid(data)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dist_gpu_out.npz"), **results)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
