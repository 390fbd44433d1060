"""Sample-sharded multi-GPU execution: one process per GPU.

Samples are independent in the regressor assembly and the QR needs exactly one exchange, so each
rank keeps a contiguous range of samples and only O(n^2) numbers cross GPUs:

* ``sum_columns``      all-reduce(sum) of diag(W^T W) (84..560 doubles) -> identical ``idx_e`` on every rank;
* ``stack_triangles``  all-gather of the per-rank nc x nc R factor (<= 1 MB) -> every rank reduces the stack
                       redundantly with ``figh_tsqr_merge`` and gets the same triangle.  (An all-reduce of
                       R^T R would square the condition number and lose the |R_kk| > 1e-8 rank decision,
                       SURVEY.md section 7, so the Householder factors themselves are exchanged.)

``RcclExchange`` moves device buffers with RCCL through the C-ABI (``figh_comm_*``); the unique id is
shipped through the ``torch.distributed`` store that the launcher (``torch.distributed.run``) set up.
``TorchExchange`` does the same exchange through ``torch.distributed`` collectives on host copies
(gloo on CPU -- used by the world_size-2 tests -- or any initialised backend).
"""
import os

import numpy as np


def shard_range(n_total, rank, world_size):
    """Contiguous sample range [lo, hi) of ``rank``: floor(p N / P) .. floor((p+1) N / P)."""
    lo = (n_total * rank) // world_size
    hi = (n_total * (rank + 1)) // world_size
    return lo, hi


class TorchExchange:
    """Exchange through an initialised ``torch.distributed`` process group, staging through the host."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    # host-level primitives (CPU-testable)
    def allreduce_sum_host(self, arr):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def allgather_host(self, arr):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        outs = [torch.empty_like(t) for _ in range(self.world_size)]
        self._dist.all_gather(outs, t, group=self.group)
        return np.stack([o.numpy() for o in outs])

    def barrier(self):
        self._dist.barrier(group=self.group)

    # pipeline interface (device buffers)
    def sum_columns(self, d_colsq, ncols):
        return self.allreduce_sum_host(d_colsq.to_host())

    def stack_triangles(self, d_R, nc):
        from . import _lib

        mine = np.empty(nc * nc)
        _lib.check(_lib.load().figh_memcpy_d2h(mine.ctypes.data, d_R.ptr, mine.nbytes))
        stack = self.allgather_host(mine)
        return _lib.DeviceArray.from_host(stack.reshape(-1)), self.world_size


class RcclExchange:
    """Exchange on device buffers with RCCL over xGMI (``figh_comm_*``)."""

    def __init__(self, world_size, rank, broadcast_bytes):
        """``broadcast_bytes(payload_or_None) -> bytes``: ships rank 0's 128-byte id to every rank."""
        import ctypes as C

        from . import _lib

        self._lib = _lib
        self.world_size, self.rank = world_size, rank
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.figh_comm_unique_id(buf))
        payload = broadcast_bytes(bytes(buf.raw) if rank == 0 else None)
        ident = C.create_string_buffer(payload, 128)
        _lib.check(lib.figh_comm_init(world_size, rank, ident))

    def sum_columns(self, d_colsq, ncols):
        self._lib.check(self._lib.load().figh_comm_allreduce_sum(d_colsq.ptr, ncols))
        return d_colsq.to_host()

    def stack_triangles(self, d_R, nc):
        need = self.world_size * nc * nc
        stack = getattr(self, "_stack", None)
        if stack is None or stack.size < need:  # kept across steps: hipMalloc / hipFree synchronise the device
            stack = self._stack = self._lib.DeviceArray((need,), np.float64)
        self._lib.check(self._lib.load().figh_comm_allgather(d_R.ptr, stack.ptr, nc * nc))
        return stack, self.world_size

    def close(self):
        self._lib.load().figh_comm_destroy()


def exchange_from_env(prefer="rccl"):
    """Build the exchange for a ``torch.distributed.run`` launch (RANK / WORLD_SIZE / MASTER_* in the env).

    Returns (exchange, info).  World size 1 -> the single-process no-op exchange.
    """
    from .pipeline import Exchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return Exchange(), {"collective": "none"}
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")  # control plane only: id exchange + barriers
    rank = dist.get_rank()
    if prefer == "rccl":
        def bcast(payload):
            obj = [payload]
            dist.broadcast_object_list(obj, src=0)
            return obj[0]

        # RCCL needs one distinct GPU per rank; if the communicator cannot be built on ANY rank (e.g. two ranks
        # sharing a device in a test), every rank takes the host-staged exchange instead -- decided collectively
        # so that no rank is left waiting in a collective, and reported in the bench line.
        ex, err = None, ""
        try:
            ex = RcclExchange(world, rank, bcast)
        except Exception as e:  # noqa: BLE001
            err = str(e)
        flags = [None] * world
        dist.all_gather_object(flags, ex is not None)
        if all(flags):
            return ex, {"collective": "rccl"}
        if ex is not None:
            ex.close()
        return TorchExchange(), {"collective": "torch.distributed/gloo host-staged (rccl unavailable: %s)" % err[:80]}
    return TorchExchange(), {"collective": "torch.distributed/" + dist.get_backend()}
