"""Sample-sharded multi-GPU execution: one process per GPU.

Samples are independent in the regressor assembly and the QR needs exactly one exchange, so each
rank keeps a contiguous range of samples and only O(n^2) numbers cross GPUs:

* ``sum_columns``      all-reduce(sum) of diag(W^T W) (84..560 doubles) -> identical ``idx_e`` on every rank;
* ``stack_triangles``  all-gather of the per-rank nc x nc R factor (<= 1 MB) -> every rank reduces the stack
                       redundantly with ``figh_tsqr_merge`` and gets the same triangle.  (An all-reduce of
                       R^T R would square the condition number and lose the |R_kk| > 1e-8 rank decision,
                       SURVEY.md section 7, so the Householder factors themselves are exchanged.)

``RcclExchange`` moves device buffers with RCCL through the C-ABI (``figh_comm_*``); the unique id travels over a
control plane, of which there are two: ``torch.distributed`` (gloo; what ``torch.distributed.run`` launches expect), or
``SocketGroup`` -- a few hundred bytes of TCP through rank 0, standard library only, so that PyTorch is optional
(``exchange_from_env(rendezvous="socket")``, ``bench.py --rendezvous socket``: RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT from the environment, the group listens on MASTER_PORT + 101).  ``TorchExchange`` / ``SocketExchange`` do the
exchange itself on host copies through the same control plane (CPU tests, and the fall-back when RCCL is unavailable).
"""
import os

import numpy as np


def shard_range(n_total, rank, world_size):
    """Contiguous sample range [lo, hi) of ``rank``: floor(p N / P) .. floor((p+1) N / P)."""
    lo = (n_total * rank) // world_size
    hi = (n_total * (rank + 1)) // world_size
    return lo, hi


class SocketGroup:
    """A control plane without PyTorch: rank 0 listens, the others connect; every collective is a gather to rank 0 and a
    broadcast back (pickled Python objects, length-prefixed).  For the handful of small messages of a run -- RCCL set-up,
    barriers, and the host-staged exchange of column norms and triangles -- not for bulk data."""

    PORT_OFFSET = 101  # MASTER_PORT itself belongs to the launcher's own store

    def __init__(self, rank, world_size, addr, port, timeout=120.0):
        import socket
        import time

        self.rank, self.world_size = int(rank), int(world_size)
        self._peers = []
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world_size)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < self.world_size - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                r = self._recv(c)
                conns[int(r)] = c
            srv.close()
            self._peers = [conns[r] for r in range(1, self.world_size)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout)
            self._send(c, self.rank)
            self._peers = [c]

    @classmethod
    def from_env(cls):
        return cls(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   int(os.environ["MASTER_PORT"]) + cls.PORT_OFFSET)

    @staticmethod
    def _send(c, obj):
        import pickle
        import struct

        data = pickle.dumps(obj, protocol=4)
        c.sendall(struct.pack("<Q", len(data)) + data)

    @staticmethod
    def _recv(c):
        import pickle
        import struct

        def exactly(n):
            buf = bytearray()
            while len(buf) < n:
                chunk = c.recv(n - len(buf))
                if not chunk:
                    raise ConnectionError("peer closed the control connection")
                buf += chunk
            return bytes(buf)
        (n,) = struct.unpack("<Q", exactly(8))
        return pickle.loads(exactly(n))

    def all_gather_object(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.rank == 0:
            objs = [obj] + [self._recv(c) for c in self._peers]
            for c in self._peers:
                self._send(c, objs)
            return objs
        self._send(self._peers[0], obj)
        return self._recv(self._peers[0])

    def broadcast_object(self, obj, src=0):
        return self.all_gather_object(obj)[src]

    def barrier(self):
        self.all_gather_object(None)

    def close(self):
        for c in self._peers:
            try:
                c.close()
            except OSError:
                pass
        self._peers = []


class TorchGroup:
    """The same control-plane interface on an initialised ``torch.distributed`` process group."""

    def __init__(self):
        import torch.distributed as dist

        self._dist = dist
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()

    def all_gather_object(self, obj):
        out = [None] * self.world_size
        self._dist.all_gather_object(out, obj)
        return out

    def broadcast_object(self, obj, src=0):
        payload = [obj]
        self._dist.broadcast_object_list(payload, src=src)
        return payload[0]

    def barrier(self):
        self._dist.barrier()


class SocketExchange:
    """Host-staged exchange over a :class:`SocketGroup` (no PyTorch): sums and stacks are formed from the gathered copies
    in rank order on every rank -- bit-identical results everywhere."""

    collective = True

    def __init__(self, group):
        self.group = group
        self.world_size, self.rank = group.world_size, group.rank

    def allreduce_sum_host(self, arr):
        parts = self.group.all_gather_object(np.ascontiguousarray(arr, dtype=np.float64))
        total = parts[0].copy()
        for p in parts[1:]:
            total += p
        return total

    def allgather_host(self, arr):
        return np.stack(self.group.all_gather_object(np.ascontiguousarray(arr, dtype=np.float64)))

    def barrier(self):
        self.group.barrier()

    def sum_columns(self, d_colsq, ncols):
        return self.allreduce_sum_host(d_colsq.to_host())

    def sum_columns_device(self, d_colsq, ncols):
        from . import _lib

        mine = np.empty(ncols)
        lib = _lib.load()
        _lib.check(lib.figh_memcpy_d2h(mine.ctypes.data, d_colsq.ptr, mine.nbytes))
        total = np.ascontiguousarray(self.allreduce_sum_host(mine))
        _lib.check(lib.figh_memcpy_h2d(d_colsq.ptr, total.ctypes.data, total.nbytes))

    def stack_triangles(self, d_R, nc):
        from . import _lib

        mine = np.empty(nc * nc)
        _lib.check(_lib.load().figh_memcpy_d2h(mine.ctypes.data, d_R.ptr, mine.nbytes))
        stack = self.allgather_host(mine)
        return _lib.DeviceArray.from_host(stack.reshape(-1)), self.world_size

    def close(self):
        self.group.close()


class TorchExchange:
    """Exchange through an initialised ``torch.distributed`` process group, staging through the host."""

    collective = True

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

    def sum_columns_device(self, d_colsq, ncols):
        """Host-staged: the sum is written back so that the device-side selection sees the global norms."""
        from . import _lib

        mine = np.empty(ncols)
        lib = _lib.load()
        _lib.check(lib.figh_memcpy_d2h(mine.ctypes.data, d_colsq.ptr, mine.nbytes))
        total = np.ascontiguousarray(self.allreduce_sum_host(mine))
        _lib.check(lib.figh_memcpy_h2d(d_colsq.ptr, total.ctypes.data, total.nbytes))

    def stack_triangles(self, d_R, nc):
        from . import _lib

        mine = np.empty(nc * nc)
        _lib.check(_lib.load().figh_memcpy_d2h(mine.ctypes.data, d_R.ptr, mine.nbytes))
        stack = self.allgather_host(mine)
        return _lib.DeviceArray.from_host(stack.reshape(-1)), self.world_size


class RcclExchange:
    """Exchange on device buffers with RCCL over xGMI (``figh_comm_*``)."""

    collective = True

    def __init__(self, world_size, rank, ident):
        """``ident``: rank 0's 128-byte ncclUniqueId (see :func:`rccl_unique_id`), already shipped to this rank.
        Callers must have agreed beforehand that EVERY rank can build the communicator (:func:`rccl_preflight`):
        ncclCommInitRank blocks until all ranks have joined."""
        import ctypes as C

        from . import _lib

        self._lib = _lib
        self.world_size, self.rank = world_size, rank
        buf = C.create_string_buffer(bytes(ident), 128)
        _lib.check(_lib.load().figh_comm_init(world_size, rank, buf))

    def sum_columns(self, d_colsq, ncols):
        self._lib.check(self._lib.load().figh_comm_allreduce_sum(d_colsq.ptr, ncols))
        return d_colsq.to_host()

    def sum_columns_device(self, d_colsq, ncols):
        self._lib.check(self._lib.load().figh_comm_allreduce_sum(d_colsq.ptr, ncols))

    def stack_triangles(self, d_R, nc):
        need = self.world_size * nc * nc
        stack = getattr(self, "_stack", None)
        if stack is None or stack.size < need:  # kept across steps: hipMalloc / hipFree synchronise the device
            stack = self._stack = self._lib.DeviceArray((need,), np.float64)
        self._lib.check(self._lib.load().figh_comm_allgather(d_R.ptr, stack.ptr, nc * nc))
        return stack, self.world_size

    def close(self):
        self._lib.load().figh_comm_destroy()


def rccl_preflight():
    """Local check, no communication: (ok, reason).  librccl loads with every symbol and a HIP device is present."""
    from . import _lib

    try:
        _lib.check(_lib.load().figh_comm_available())
        return True, ""
    except Exception as e:  # noqa: BLE001
        return False, str(e)


def rccl_unique_id():
    """Rank 0: a fresh 128-byte ncclUniqueId."""
    import ctypes as C

    from . import _lib

    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().figh_comm_unique_id(buf))
    return bytes(buf.raw)


def exchange_from_env(prefer="rccl", device_key=None, rendezvous="torch"):
    """Build the exchange for a ``torch.distributed.run``-style launch (RANK / WORLD_SIZE / MASTER_* in the env).
    ``rendezvous``: "torch" (gloo process group) or "socket" (:class:`SocketGroup`: no PyTorch anywhere).

    Returns (exchange, info).  World size 1 -> the single-process no-op exchange.

    The RCCL set-up runs in phases so that no failure can leave a rank blocked in a collective:

    1. every rank runs the local preflight and states which device it drives (``device_key``, default
       ``(hostname, LOCAL_RANK)``); the outcomes are all-gathered over the gloo control plane;
    2. only if EVERY rank passed and no two ranks share a device (RCCL needs one GPU per rank) does rank 0 create the
       unique id; the id broadcast is executed by every rank in either case (an empty payload = "no RCCL");
    3. all ranks call ``ncclCommInitRank``.  A rank on which that call cannot even be entered (device binding, id
       buffer, allocation: an error raised LOCALLY before the rendezvous) must not go on to a collective while its peers
       are blocked inside ``ncclCommInitRank`` waiting for it: it exits non-zero at once and the launcher tears the job
       down.  Only a failure reported by the rendezvous itself (every rank returns from it) is agreed on collectively,
       and then every rank takes the host-staged exchange.

    Otherwise all ranks take the host-staged exchange -- decided collectively, reported in the bench line.
    """
    from .pipeline import Exchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return Exchange(), {"collective": "none"}
    import socket

    if rendezvous == "socket":
        group = SocketGroup.from_env()
        host_exchange, plane = (lambda: SocketExchange(group)), "socket control plane"
    else:
        import torch.distributed as dist

        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")  # control plane only: id exchange + barriers
        group = TorchGroup()
        host_exchange, plane = TorchExchange, "torch.distributed/gloo"
    rank = group.rank
    if prefer != "rccl":
        ex = host_exchange()
        ex.control = group
        return ex, {"collective": plane + " host-staged"}
    # phase 1
    ok, why = rccl_preflight()
    if device_key is None:
        # the device this process actually drives: launchers bind LOCAL_RANK modulo the number of visible devices
        local = int(os.environ.get("LOCAL_RANK", rank))
        try:
            from . import _lib
            ndev = _lib.device_count()  # hipGetDeviceCount: creates no context
        except Exception:  # noqa: BLE001
            ndev = 0
        device_key = (socket.gethostname(), local % ndev if ndev > 0 else local)
        try:  # the physical device behind that index: launchers that isolate one GPU per rank give every rank index 0
            import ctypes as C
            from . import _lib
            bus = C.create_string_buffer(64)
            if _lib.load().figh_device_pci_bus_id(device_key[1], bus, 64) == 0 and bus.value:
                device_key = (device_key[0], bus.value.decode())
        except Exception:  # noqa: BLE001
            pass
    states = group.all_gather_object((bool(ok), why, tuple(device_key)))
    all_ok = all(s[0] for s in states)
    distinct = len({s[2] for s in states}) == world
    # phase 2 (every rank takes part in the broadcast, whatever phase 1 said)
    payload = [None, ""]  # (unique id | None, why not) -- the reason travels with it so that every rank reports the same
    if rank == 0:
        if all_ok and distinct:
            try:
                payload[0] = rccl_unique_id()
            except Exception as e:  # noqa: BLE001
                payload[1] = str(e)
        else:
            payload[1] = "two ranks share a device" if all_ok else next(s[1] for s in states if not s[0])
    payload = group.broadcast_object(payload, src=0)
    if payload[0] is None:
        ex = host_exchange()
        ex.control = group
        return ex, {"collective": "%s host-staged (rccl unavailable: %s)" % (plane, payload[1][:80])}
    # phase 3: every rank attempts the initialisation and reports; one failure sends all of them to the host-staged
    # exchange (a rank that fails returns from ncclCommInitRank with an error, it does not leave the others inside it)
    ex, err = None, ""
    try:
        ex = RcclExchange(world, rank, payload[0])
    except Exception as e:  # noqa: BLE001
        err = str(e)
        from . import _lib
        if getattr(e, "code", None) != _lib.ERR_COMM:
            # not an answer of the rendezvous (figh_comm_init reports those as FIGH_ERR_COMM): this rank never joined it,
            # its peers are still inside ncclCommInitRank and no collective can be reached -- fail fast, no fallback
            import sys
            sys.stderr.write("rank %d: RCCL set-up failed before the rendezvous (%s); exiting\n" % (rank, err))
            sys.stderr.flush()
            os._exit(3)
    outcomes = group.all_gather_object((ex is not None, err))
    if all(o[0] for o in outcomes):
        ex.control = group
        return ex, {"collective": "rccl", "control_plane": plane}
    if ex is not None:
        ex.close()
    reason = next(o[1] for o in outcomes if not o[0])
    ex = host_exchange()
    ex.control = group
    return ex, {"collective": "%s host-staged (rccl unavailable: %s)" % (plane, reason[:80])}
