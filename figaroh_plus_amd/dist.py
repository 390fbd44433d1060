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


def _wire_encode(obj, out):
    """Tagged, self-describing encoding of the few value types the control plane carries (None, bool, int, float, str,
    bytes, list / tuple, dict with string keys, float64 / int64 arrays).  Data only: decoding never constructs anything but these types -- the frames
    of a TCP peer are not unpickled."""
    import struct

    if obj is None:
        out += b"N"
    elif isinstance(obj, (bool, np.bool_)):
        out += b"T" if obj else b"F"
    elif isinstance(obj, (int, np.integer)):
        out += b"i" + struct.pack("<q", int(obj))
    elif isinstance(obj, (float, np.floating)):
        out += b"d" + struct.pack("<d", float(obj))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out += b"s" + struct.pack("<Q", len(raw)) + raw
    elif isinstance(obj, (bytes, bytearray)):
        out += b"b" + struct.pack("<Q", len(obj)) + bytes(obj)
    elif isinstance(obj, (list, tuple)):
        out += (b"l" if isinstance(obj, list) else b"t") + struct.pack("<Q", len(obj))
        for item in obj:
            _wire_encode(item, out)
    elif isinstance(obj, dict):
        out += b"m" + struct.pack("<Q", len(obj))
        for key, item in obj.items():
            if not isinstance(key, str):
                raise TypeError("control plane: dictionary keys are strings")
            _wire_encode(key, out)
            _wire_encode(item, out)
    elif isinstance(obj, np.ndarray):
        if obj.dtype == np.float64:
            code = b"D"
        elif obj.dtype == np.int64:
            code = b"I"
        else:
            raise TypeError("control plane: arrays travel as float64 or int64, not %s" % obj.dtype)
        arr = np.ascontiguousarray(obj)
        out += b"a" + code + struct.pack("<B", arr.ndim) + struct.pack("<%dQ" % arr.ndim, *arr.shape) + arr.tobytes()
    else:
        raise TypeError("control plane cannot carry a %s" % type(obj).__name__)


def _wire_decode(buf, pos=0, depth=0):
    import struct

    if depth > 8:
        raise ValueError("control plane frame nested too deeply")
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag in (b"T", b"F"):
        return tag == b"T", pos
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"d":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"s", b"b"):
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if pos + n > len(buf):
            raise ValueError("control plane frame truncated")
        raw = bytes(buf[pos:pos + n])
        return (raw.decode("utf-8") if tag == b"s" else raw), pos + n
    if tag in (b"l", b"t"):
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf):
            raise ValueError("control plane frame: impossible element count")
        items = []
        for _ in range(n):
            item, pos = _wire_decode(buf, pos, depth + 1)
            items.append(item)
        return (items if tag == b"l" else tuple(items)), pos
    if tag == b"m":
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf):
            raise ValueError("control plane frame: impossible element count")
        items = {}
        for _ in range(n):
            key, pos = _wire_decode(buf, pos, depth + 1)
            if not isinstance(key, str):
                raise ValueError("control plane frame: dictionary key")
            items[key], pos = _wire_decode(buf, pos, depth + 1)
        return items, pos
    if tag == b"a":
        code = buf[pos:pos + 1]
        if code not in (b"D", b"I"):
            raise ValueError("control plane frame: unknown array type")
        ndim = buf[pos + 1]
        pos += 2
        if ndim > 4:
            raise ValueError("control plane frame: array rank")
        shape = struct.unpack_from("<%dQ" % ndim, buf, pos)
        pos += 8 * ndim
        count = 1
        for d in shape:
            count *= d
        nbytes = 8 * count
        if pos + nbytes > len(buf):
            raise ValueError("control plane frame truncated")
        arr = np.frombuffer(buf, dtype=np.float64 if code == b"D" else np.int64, count=count, offset=pos).reshape(shape).copy()
        return arr, pos + nbytes
    raise ValueError("control plane frame: unknown tag %r" % tag)


class SocketGroup:
    """A control plane without PyTorch: rank 0 listens, the others connect; every collective is a gather to rank 0 and a
    broadcast back.  Frames are length-prefixed and carry plain data in a tagged encoding (``_wire_encode``: no pickle --
    nothing a peer sends is ever executed).  With ``FIGH_COMM_SECRET`` in the environment (same value on every rank) every
    frame is authenticated with HMAC-SHA256 and frames from anybody else are rejected.  For the handful of small messages
    of a run -- RCCL set-up, barriers, and the host-staged exchange of column norms and triangles -- not for bulk data.

    ``timeout`` bounds the rendezvous only (accept / connect / hello); once the group is formed the sockets block without a
    limit: a rank may legitimately wait at a barrier for as long as its slowest peer computes."""

    PORT_OFFSET = 101  # MASTER_PORT itself belongs to the launcher's own store
    MAX_FRAME = 1 << 30

    def __init__(self, rank, world_size, addr, port, timeout=120.0, secret=None):
        import socket
        import time

        self.rank, self.world_size = int(rank), int(world_size)
        if not 0 <= self.rank < self.world_size:
            raise ValueError("rank %d outside [0, %d)" % (self.rank, self.world_size))
        if secret is None:
            secret = os.environ.get("FIGH_COMM_SECRET", "")
        self._key = secret.encode("utf-8") if isinstance(secret, str) else bytes(secret)
        self._peers = []
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world_size)
            deadline = time.time() + timeout
            conns = {}
            try:
                while len(conns) < self.world_size - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise TimeoutError("control plane: %d of %d ranks joined within %.0f s"
                                           % (len(conns) + 1, self.world_size, timeout))
                    srv.settimeout(left)
                    c, _ = srv.accept()
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(min(10.0, timeout))
                    # a connection that does not introduce itself properly (stray client, wrong secret, rank out of range,
                    # a rank that is already here) is dropped; the rendezvous goes on waiting for the real one
                    try:
                        hello = self._recv(c)
                        if not (isinstance(hello, tuple) and len(hello) == 3 and hello[0] == "figh-hello"):
                            raise ValueError("not a hello frame")
                        r, w = hello[1], hello[2]
                        if not (isinstance(r, int) and isinstance(w, int)) or w != self.world_size or not 1 <= r < w:
                            raise ValueError("rank %r of %r" % (r, w))
                        if r in conns:
                            raise ValueError("rank %d joined twice" % r)
                    except Exception:  # noqa: BLE001
                        c.close()
                        continue
                    conns[r] = c
            finally:
                srv.close()
            self._peers = [conns[r] for r in range(1, self.world_size)]
            for c in self._peers:
                self._send(c, ("figh-welcome", self.world_size))
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
            self._send(c, ("figh-hello", self.rank, self.world_size))
            ack = self._recv(c)
            if ack != ("figh-welcome", self.world_size):
                raise ConnectionError("control plane: unexpected answer from rank 0")
            self._peers = [c]
        for c in self._peers:
            c.settimeout(None)

    @classmethod
    def from_env(cls):
        return cls(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   int(os.environ["MASTER_PORT"]) + cls.PORT_OFFSET)

    def _mac(self, data):
        import hashlib
        import hmac

        return hmac.new(self._key, data, hashlib.sha256).digest() if self._key else b""

    def _send(self, c, obj):
        import struct

        data = bytearray()
        _wire_encode(obj, data)
        data = bytes(data)
        c.sendall(struct.pack("<Q", len(data)) + self._mac(data) + data)

    def _recv(self, c):
        import hmac
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
        if n > self.MAX_FRAME:
            raise ValueError("control plane frame of %d bytes refused" % n)
        mac = exactly(32) if self._key else b""
        data = exactly(n)
        if self._key and not hmac.compare_digest(mac, self._mac(data)):
            raise ConnectionError("control plane frame failed authentication")
        obj, pos = _wire_decode(data)
        if pos != len(data):
            raise ValueError("control plane frame: trailing bytes")
        return obj

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


def allreduce_normal_terms(exchange, colsq, G, g, tau_sq, rows):
    """Collective (1) of SURVEY.md section 8e: ``[colsq | G | W^T tau | tau^T tau | rows]`` packed into ONE fp64 buffer and
    summed over the ranks with one all-reduce -- everything the normal-equation consumers of the path need from all shards:
    the elimination (``colsq``, regressor.py:258-279), the SIP quadratic program's data terms (identification_tools.py:528-531:
    ``W^T W``, ``W^T tau``) and the residual variance of ``relative_stdev`` (:204-234: ``tau^T tau - 2 phi^T W^T tau + phi^T W^T
    W phi`` over ``rows - n`` degrees of freedom).  n^2 + 2 n + 2 doubles for n columns (57 KB .. 2.5 MB): one latency-bound
    message.  (NOT for the rank decision: ``|R_kk| > tol_qr`` needs the Householder factors, see ``stack_triangles``.)

    Arguments are this rank's host arrays / numbers (``colsq`` may be None); returns the summed (colsq, G, g, tau_sq, rows).
    Host-staged exchanges sum the pack on the host; ``RcclExchange`` sums it in HBM."""
    G = np.ascontiguousarray(G, dtype=np.float64)
    n = G.shape[0]
    ncs = 0 if colsq is None else len(colsq)
    pack = np.concatenate([np.zeros(0) if colsq is None else np.asarray(colsq, dtype=np.float64), G.reshape(-1),
                           np.asarray(g, dtype=np.float64).reshape(-1), [float(tau_sq), float(rows)]])
    if exchange is None or exchange.world_size == 1:
        total = pack
    elif hasattr(exchange, "allreduce_sum_host"):
        total = np.asarray(exchange.allreduce_sum_host(pack))
    else:
        from . import _lib

        d = _lib.DeviceArray.from_host(pack)
        total = np.asarray(exchange.sum_columns(d, pack.size))
        d.free()
    colsq_t = None if colsq is None else total[:ncs].copy()
    G_t = total[ncs:ncs + n * n].reshape(n, n).copy()
    g_t = total[ncs + n * n:ncs + n * n + n].copy()
    return colsq_t, G_t, g_t, float(total[-2]), float(total[-1])


def allgather_max(exchange, value):
    """max over the ranks of a host scalar (``sf2 = 1 / (max(tau) len(tau))`` of the SIP program needs the global maximum):
    a gather of one number per rank through the exchange."""
    if exchange is None or exchange.world_size == 1:
        return float(value)
    if hasattr(exchange, "allgather_host"):
        return float(np.max(exchange.allgather_host(np.array([float(value)]))))
    from . import _lib

    d = _lib.DeviceArray.from_host(np.array([float(value)]))
    stack, count = exchange.stack_triangles(d, 1)
    out = np.empty(count)
    _lib.check(_lib.load().figh_memcpy_d2h(out.ctypes.data, stack.ptr, out.nbytes))
    return float(out.max())


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
