import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from xugrid_amd import meshgen
from xugrid_amd.distributed import HipBackend, _t, shard_lists, ShardedOverlapRegridder
class FakeDist:
    def get_rank(self, g=None): return 0
    def get_world_size(self, g=None): return 1
    def get_backend(self, g=None): return "nccl"
    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        out.copy_(inp); return None
be = HipBackend(0)
sxy, sf = meshgen.triangle_mesh(500_000, 0, delaunay=False)
txy, tf = meshgen.triangle_mesh(500_000, 1, 30.0, 0.7, delaunay=False)
rg = ShardedOverlapRegridder(sxy, sf, txy, tf, be, dist=FakeDist())
def t(f, n=10):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return 1e3*(time.perf_counter()-t0)/n
full = rg._full
print("shard_lists %.3f ms" % t(lambda: shard_lists(full, 1, 0, "balanced", be)))
lf, lt = shard_lists(full, 1, 0, "balanced", be)
print("gathers %.3f ms" % t(lambda: (full[1][lf], full[3][lt])))
sfa_l, tfa_l = full[1][lf], full[3][lt]
print("build_weights_t %.3f ms" % t(lambda: be.build_weights_t(full[0], sfa_l, full[2], tfa_l)))
print("rebuild_weights %.3f ms" % t(lambda: be.rebuild_weights()))
print("sparse exchange setup %.3f ms" % t(lambda: rg._setup_sparse_exchange()))
print("setup() %.3f ms" % t(lambda: rg.setup()))
