"""
Test infrastructure: a loop-back stand-in for the four ``torch.distributed`` calls ``xugrid_amd.distributed`` uses,
so that W "ranks" -- W threads of ONE process -- run the real ``ShardedOverlapRegridder`` + ``HipBackend`` code on
ONE GPU.  The GPU box has a single device; RCCL itself only ever sees world_size 1 there
(tests/test_gpu_distributed.py), so this is how the HIP combine kernels get to see several contributions per
target: ``xr_reduce_partial_rows_dev`` with multi-sender lists, ``xr_partial_fill_identity_dev`` + the plane-wise
sum / max + ``xr_finalize_partial_dev``, shard-local column ids.  The semantics are those of the collectives
(``all_to_all_single`` with split sizes, ``reduce_scatter_tensor`` SUM / MAX in rank order, ``all_gather``,
``barrier``); the transport is device-to-device copies on the one device.
"""
import threading
from types import SimpleNamespace

import torch


class _Done:
    def wait(self):
        return True


class LoopbackWorld:
    def __init__(self, world_size):
        self.world_size = world_size
        self.barrier = threading.Barrier(world_size)
        self.slots = [None] * world_size
        self.sent_bytes = [0] * world_size

    def view(self, rank):
        return LoopbackDist(self, rank)


class LoopbackDist:
    ReduceOp = SimpleNamespace(SUM="sum", MAX="max")

    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world.world_size

    def get_backend(self, group=None):
        return "nccl"  # take the reduce_scatter_tensor / device-tensor branches of the product code

    def _exchange(self, payload):
        """every rank deposits, all wait, every rank reads all deposits -> list; a second wait before reuse."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()  # the threads share one device: what a rank produced is complete before it is read
        self.world.slots[self.rank] = payload
        self.world.barrier.wait()
        got = list(self.world.slots)
        return got

    def _release(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.world.barrier.wait()

    def barrier(self, group=None):
        self.world.barrier.wait()

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None, group=None,
                          async_op=False):
        W = self.world.world_size
        if input_split_sizes is None:
            assert input.shape[0] % W == 0
            input_split_sizes = [input.shape[0] // W] * W
        got = self._exchange((input, list(input_split_sizes)))
        pieces = []
        for sender in range(W):
            tensor, splits = got[sender]
            start = sum(splits[: self.rank])
            pieces.append(tensor[start:start + splits[self.rank]])
        if output_split_sizes is not None:
            assert [p.shape[0] for p in pieces] == list(output_split_sizes), "split sizes disagree between ranks"
        cat = torch.cat(pieces, dim=0) if pieces else input[:0]
        assert cat.shape == output.shape, (cat.shape, output.shape)
        output.copy_(cat)
        self.world.sent_bytes[self.rank] += input.numel() * input.element_size()
        self._release()
        return _Done() if async_op else None

    def reduce_scatter_tensor(self, output, input, op="sum", group=None, async_op=False):
        got = self._exchange(input)
        acc = got[0][self.rank].clone()
        for sender in range(1, self.world.world_size):
            piece = got[sender][self.rank]
            acc = torch.maximum(acc, piece) if op == "max" else acc + piece
        output.copy_(acc)
        self.world.sent_bytes[self.rank] += input.numel() * input.element_size()
        self._release()
        return _Done() if async_op else None

    def all_gather(self, tensor_list, tensor, group=None):
        got = self._exchange(tensor)
        for dst, src in zip(tensor_list, got):
            dst.copy_(src)
        self._release()


def run_ranks(world_size, fn):
    """fn(dist_view, rank) on ``world_size`` threads; returns the list of results, re-raises the first failure."""
    world = LoopbackWorld(world_size)
    results, errors = [None] * world_size, []

    def body(rank):
        try:
            results[rank] = fn(world.view(rank), rank)
        except BaseException as exc:  # noqa: BLE001 -- reported to the caller below
            errors.append((rank, exc))
            world.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world_size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)] or errors
        raise RuntimeError(f"rank {real[0][0]} failed: {real[0][1]!r}") from real[0][1]
    return results, world
