"""CPU suite, part 4: the N>1 path with world_size 2 over gloo -- sharding, column-norm all-reduce and the
stack-of-triangles exchange give the single-process result (local factors come from the oracle here)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT


def _free_port():
    """A port p with p + 101 free as well: MASTER_PORT belongs to the launcher's store, the socket control plane listens on
    MASTER_PORT + 101 (dist.SocketGroup.PORT_OFFSET) -- checking only p left a rare collision on p + 101."""
    for _ in range(50):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        if port + 101 > 65535:
            continue
        try:
            with socket.socket() as s2:
                s2.bind(("127.0.0.1", port + 101))
            return port
        except OSError:
            continue
    raise RuntimeError("no free port pair")


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import json
    import torch.distributed as dist
    import oracle_c
    from figaroh_plus_amd.dist import TorchExchange, shard_range
    from figaroh_plus_amd.model import Model

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = TorchExchange()
    g = np.load(os.path.join(GOLD, "cfg2_ur10.npz"))
    flat = Model.from_flat(os.path.join(ROOT, "figaroh_plus_amd", "models", "ur10.json")).to_flat()
    om = oracle_c.OracleModel(flat)
    N = len(g["q_big"])
    lo, hi = shard_range(N, rank, world)
    W = om.build_regressor_basic(g["q_big"][lo:hi], g["v_big"][lo:hi], g["a_big"][lo:hi], 0, 0)
    tau_full = g["tau"].reshape(6, N)
    tau = np.ascontiguousarray(tau_full[:, lo:hi]).reshape(-1)
    colsq = ex.allreduce_sum_host(oracle_c.colsq(W))
    keep = [i for i in range(W.shape[1]) if not colsq[i] < 1e-6]
    R, qtb = oracle_c.householder_r(W, keep, tau)
    # local (n+1)x(n+1) augmented triangle, as figh_tsqr returns it
    n = len(keep)
    Wt = np.c_[W[:, keep], tau]
    Raug = np.linalg.qr(Wt, mode="r")
    stack = ex.allgather_host(Raug)
    assert stack.shape == (world, n + 1, n + 1)
    Rm = np.linalg.qr(stack.reshape(world * (n + 1), n + 1), mode="r")
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), colsq=colsq, R=Rm, lo=lo, hi=hi)
    ex.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_exchange_matches_single_process(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 200, 200, 400)
    g = np.load(os.path.join(GOLD, "cfg2_ur10.npz"))
    assert np.array_equal(r0["colsq"], r1["colsq"])
    assert np.abs(r0["colsq"] - g["colsq_big"]).max() <= 1e-12 * g["colsq_big"].max()
    assert np.array_equal(r0["R"], r1["R"])  # every rank reduces the same stack -> identical triangle
    d = np.abs(np.diag(r0["R"]))[:-1]
    assert [i for i in range(len(d)) if d[i] > 1e-8] == list(g["idx_base"])
    # phi from the merged triangle == single-process pinv solution
    n = len(d)
    idx_base = list(g["idx_base"])
    idx_regroup = [i for i in range(n) if i not in set(idx_base)]
    Rp = np.linalg.qr(r0["R"][:, idx_base + idx_regroup + [n]], mode="r")
    r = len(idx_base)
    phi = np.linalg.solve(Rp[:r, :r], Rp[:r, n])
    assert np.abs(phi - g["phi_pinv"]).max() <= 1e-8 * np.abs(g["phi_pinv"]).max()


def _hip_fixture_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from figaroh_plus_amd.dist import TorchExchange

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = TorchExchange()
    z = np.load(os.path.join(GOLD, "hip_triangles_ur10.npz"))
    stack = ex.allgather_host(z["R_rank%d" % rank])  # what stack_triangles hands to figh_tsqr_merge on every rank
    nc = stack.shape[1]
    Rm = np.linalg.qr(stack.reshape(world * nc, nc), mode="r")
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), R=Rm)
    ex.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_exchange_of_recorded_hip_triangles(tmp_path):
    """The same exchange fed with factors the HIP path produced (tests/golden/hip_triangles_ur10.npz, recorded on an MI355X
    by tools/record_hip_triangles.py: the level-0 + merge triangle of each of two sample shards, and the one-process
    triangle): the stack of the two device triangles reduces to the one-process device triangle (|R| to 1e-12), the rank
    decision on it is the golden base set."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_hip_fixture_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz")["R"], np.load(tmp_path / "rank1.npz")["R"]
    assert np.array_equal(r0, r1)
    z = np.load(os.path.join(GOLD, "hip_triangles_ur10.npz"))
    one = z["R_one_process"]
    g = np.load(os.path.join(GOLD, "cfg2_ur10.npz"))
    d = np.abs(np.diag(r0))[:-1]
    assert [i for i in range(len(d)) if d[i] > 1e-8] == list(g["idx_base"])
    first_dep = min(i for i in range(len(d)) if d[i] <= 1e-8)
    assert np.abs(np.abs(r0[:first_dep]) - np.abs(one[:first_dep])).max() <= 1e-12 * np.abs(one).max()
    assert np.abs(r0.T @ r0 - one.T @ one).max() <= 1e-12 * np.abs(one.T @ one).max()


def _setup_worker(rank, world, port, out_dir, scenario):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    from figaroh_plus_amd import dist as fd

    calls = {"unique_id": 0}
    if scenario == "rank1_fails":      # librccl missing on ONE rank only
        fd.rccl_preflight = lambda: (rank != 1, "simulated: librccl missing on rank 1" if rank == 1 else "")
    elif scenario == "id_fails":       # every preflight passes, rank 0 cannot create the id
        fd.rccl_preflight = lambda: (True, "")
    elif scenario == "shared_device":  # both ranks name the same GPU
        fd.rccl_preflight = lambda: (True, "")
    elif scenario in ("init_fails", "init_local_failure"):
        # preflight and id are fine; "init_fails": ncclCommInitRank itself reports an error on one rank (FIGH_ERR_COMM:
        # every rank returns from the rendezvous); "init_local_failure": one rank cannot even enter it
        fd.rccl_preflight = lambda: (True, "")
        from figaroh_plus_amd import _lib as flib

        class FakeRccl:
            closed = False

            def __init__(self, world_, rank_, ident):
                if rank_ == 1 and scenario == "init_fails":
                    raise flib.FighError(flib.ERR_COMM, "simulated: ncclCommInitRank failed on rank 1")
                if rank_ == 1:
                    raise RuntimeError("simulated: hipSetDevice failed on rank 1")

            def close(self):
                calls["closed"] = calls.get("closed", 0) + 1

        fd.RcclExchange = FakeRccl

    def unique_id():
        calls["unique_id"] += 1
        if scenario in ("init_fails", "init_local_failure"):
            return b"0" * 128
        raise RuntimeError("simulated: ncclGetUniqueId failed")

    fd.rccl_unique_id = unique_id
    key = ("box", 0) if scenario == "shared_device" else None
    ex, info = fd.exchange_from_env("rccl", device_key=key)
    # the fallback exchange must work right away on every rank: no rank is stuck in another collective
    total = ex.allreduce_sum_host(np.array([float(rank + 1)]))
    json.dump({"collective": info["collective"], "kind": type(ex).__name__, "sum": float(total[0]),
               "unique_id_calls": calls["unique_id"], "closed": calls.get("closed", 0)}, open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    ex.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario", ["rank1_fails", "id_fails", "shared_device", "no_device", "init_fails"])
def test_rccl_setup_is_decided_collectively(tmp_path, scenario):
    """exchange_from_env (the RCCL set-up could deadlock on an asymmetric failure): preflight outcomes are
    all-gathered, every rank takes part in the id broadcast, and all ranks take the same exchange.  Scenarios: the
    preflight fails on one rank only; rank 0 fails to create the id after a clean preflight; two ranks name one device;
    and the real preflight of this GPU-less container."""
    import json
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_setup_worker, args=(2, port, str(tmp_path), scenario), nprocs=2, join=True)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert res[0]["collective"] == res[1]["collective"] and "host-staged" in res[0]["collective"]
    assert all(r["kind"] == "TorchExchange" and r["sum"] == 3.0 for r in res)
    if scenario in ("rank1_fails", "shared_device", "no_device"):
        assert res[0]["unique_id_calls"] == 0  # the id is only created after a clean, collective phase 1
    if scenario == "rank1_fails":
        assert "rank 1" in res[0]["collective"]
    if scenario == "shared_device":
        assert "share a device" in res[0]["collective"]
    if scenario == "id_fails":
        assert res[0]["unique_id_calls"] == 1 and res[1]["unique_id_calls"] == 0
    if scenario == "init_fails":  # the rank whose communicator did come up gives it back
        assert "rank 1" in res[0]["collective"] and res[0]["closed"] == 1 and res[1]["closed"] == 0


def test_shard_range_partitions():
    from figaroh_plus_amd.dist import shard_range
    for N in (0, 1, 7, 1000, 10 ** 6 + 3):
        for P in (1, 2, 3, 8):
            parts = [shard_range(N, r, P) for r in range(P)]
            assert parts[0][0] == 0 and parts[-1][1] == N
            assert all(parts[i][1] == parts[i + 1][0] for i in range(P - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_rccl_local_setup_failure_fails_fast(tmp_path):
    """a rank that fails BEFORE joining ncclCommInitRank (its peers are blocked inside the rendezvous) must
    not wander off into a collective: it exits non-zero at once and the launcher tears the job down."""
    import torch.multiprocessing as mp
    port = _free_port()
    with pytest.raises(Exception) as e:
        mp.spawn(_setup_worker, args=(2, port, str(tmp_path), "init_local_failure"), nprocs=2, join=True)
    assert "exit code 3" in str(e.value) or "exitcode" in str(e.value).lower()
    assert not os.path.exists(tmp_path / "rank1.json")


def _socket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import json
    from figaroh_plus_amd import dist as fd

    ex, info = fd.exchange_from_env("rccl", rendezvous="socket")  # (no HIP device here: host-staged, decided collectively)
    rng = np.random.default_rng(100 + rank)
    mine = rng.standard_normal(7)
    total = ex.allreduce_sum_host(mine)
    stack = ex.allgather_host(np.full((3, 3), float(rank)))
    objs = ex.control.all_gather_object({"rank": rank})
    top = ex.control.broadcast_object("from %d" % rank, src=world - 1)
    ex.barrier()
    json.dump({"collective": info["collective"], "kind": type(ex).__name__, "total": total.tolist(), "mine": mine.tolist(),
               "stack": stack.tolist(), "objs": objs, "top": top, "torch_loaded": "torch" in sys.modules},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    ex.close()


@pytest.mark.timeout(120)
def test_socket_rendezvous_needs_no_torch(tmp_path):
    """exchange_from_env(rendezvous="socket"): the control plane and the host-staged exchange over plain TCP through rank 0
    (figaroh_plus_amd.dist.SocketGroup) -- three ranks, identical sums and stacks everywhere, PyTorch never imported."""
    import json
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_socket_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    want = np.sum([r["mine"] for r in res], axis=0)
    for r in res:
        assert r["kind"] == "SocketExchange" and "socket control plane host-staged" in r["collective"]
        assert np.array_equal(np.array(r["total"]), np.array(res[0]["total"]))  # bit-identical on every rank
        assert np.abs(np.array(r["total"]) - want).max() <= 1e-15
        assert np.array(r["stack"]).shape == (world, 3, 3) and [s[0][0] for s in r["stack"]] == [0.0, 1.0, 2.0]
        assert r["objs"] == [{"rank": k} for k in range(world)] and r["top"] == "from %d" % (world - 1)
        assert not r["torch_loaded"]


def test_control_plane_codec_is_data_only():
    """The socket control plane carries tagged plain data (no pickle.loads on bytes from a TCP peer): the
    value types of a run round-trip, anything else is refused when ENCODING, and a pickle payload does not decode."""
    import pickle
    from figaroh_plus_amd import dist as fd

    objs = [None, True, 3, -1.5, "why", b"\x00\x01id", (True, "", ("host", "0000:05:00.0")), [b"x" * 128, ""],
            {"rank": 2, "t": [1.0, 2.0]}, np.arange(12.0).reshape(3, 4), np.arange(5, dtype=np.int64)]
    for o in objs:
        buf = bytearray()
        fd._wire_encode(o, buf)
        back, pos = fd._wire_decode(bytes(buf))
        assert pos == len(buf)
        if isinstance(o, np.ndarray):
            assert back.dtype == o.dtype and np.array_equal(back, o)
        else:
            assert back == o and type(back) is type(o)
    with pytest.raises(TypeError):
        fd._wire_encode(object(), bytearray())
    with pytest.raises(TypeError):
        fd._wire_encode(np.zeros(3, dtype=np.float32), bytearray())
    with pytest.raises(ValueError):
        fd._wire_decode(pickle.dumps({"rank": 1}, protocol=4))
    with pytest.raises(ValueError):
        fd._wire_decode(b"s" + (1 << 40).to_bytes(8, "little") + b"abc")  # length beyond the frame
    assert "pickle" not in open(fd.__file__).read().split('"""', 2)[2].replace("no pickle", "").replace("not unpickled", "")


def _rendezvous_worker(rank, world, port, out_dir, secret):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from figaroh_plus_amd import dist as fd

    g = fd.SocketGroup(rank, world, "127.0.0.1", port, timeout=30.0, secret=secret)
    got = g.all_gather_object(rank * 10)
    g.barrier()
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write(repr(got))
    g.close()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("secret", ["", "s3cret"])
def test_socket_rendezvous_rejects_strangers(tmp_path, secret):
    """Rank 0 validates who joins: a stray connection that sends garbage, a frame with a wrong secret, a rank out of range and
    a duplicate rank are dropped and the rendezvous completes with the real peers (before: KeyError / overwritten socket
    and a 120 s hang)."""
    import multiprocessing as mp
    import struct
    import time
    from figaroh_plus_amd import dist as fd

    ctx = mp.get_context("spawn")
    port = _free_port()
    world = 3
    p0 = ctx.Process(target=_rendezvous_worker, args=(0, world, port, str(tmp_path), secret))
    p0.start()

    def connect():
        deadline = time.time() + 20
        while True:
            try:
                return socket.create_connection(("127.0.0.1", port), timeout=5.0)
            except OSError:
                assert time.time() < deadline
                time.sleep(0.05)

    def frame(obj, key):
        import hashlib
        import hmac
        data = bytearray()
        fd._wire_encode(obj, data)
        data = bytes(data)
        mac = hmac.new(key.encode(), data, hashlib.sha256).digest() if key else b""
        return struct.pack("<Q", len(data)) + mac + data

    strays = []
    c = connect()
    c.sendall(b"GET / HTTP/1.0\r\n\r\n")  # not a frame at all
    strays.append(c)
    c = connect()
    c.sendall(frame(("figh-hello", 7, world), secret))  # rank out of range
    strays.append(c)
    c = connect()
    c.sendall(frame(("figh-hello", 1, world), secret + "x"))  # wrong secret (or, without one, an unauthenticated extra MAC)
    strays.append(c)
    # rank 1 is played by this process, frame by frame, so that the duplicate is known to arrive after the real one
    # (rank 0 accepts connections in the order they were established and reads each hello before the next accept)
    real1 = connect()
    real1.sendall(frame(("figh-hello", 1, world), secret))
    c = connect()
    c.sendall(frame(("figh-hello", 1, world), secret))  # rank 1 a second time
    strays.append(c)
    p2 = ctx.Process(target=_rendezvous_worker, args=(2, world, port, str(tmp_path), secret))
    p2.start()
    peer = fd.SocketGroup.__new__(fd.SocketGroup)
    peer._key = secret.encode()
    real1.settimeout(60)
    assert peer._recv(real1) == ("figh-welcome", world)
    peer._send(real1, 10)
    assert peer._recv(real1) == [0, 10, 20]
    peer._send(real1, None)  # the barrier
    assert peer._recv(real1) == [None, None, None]
    for p in (p0, p2):
        p.join(60)
        assert p.exitcode == 0
    real1.close()
    for c in strays:
        c.close()
    for r in (0, 2):
        assert open(tmp_path / ("r%d.txt" % r)).read() == "[0, 10, 20]"


def _normal_terms_worker(rank, world, port, out_dir, plane):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_c
    from figaroh_plus_amd import dist as fd
    from figaroh_plus_amd.identification.identification_tools import relative_stdev_from_normal_terms
    from figaroh_plus_amd.model import Model

    if plane == "torch":
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ex = fd.TorchExchange()
    else:
        ex = fd.SocketExchange(fd.SocketGroup.from_env())
    g = np.load(os.path.join(GOLD, "cfg2_ur10.npz"))
    flat = Model.from_flat(os.path.join(ROOT, "figaroh_plus_amd", "models", "ur10.json")).to_flat()
    om = oracle_c.OracleModel(flat)
    N = len(g["q_big"])
    # unequal shards on purpose (the divisor of the variance is the SUMMED row count)
    cut = [0, 150, N] if world == 2 else [0, 100, 250, N]
    lo, hi = cut[rank], cut[rank + 1]
    W = om.build_regressor_basic(g["q_big"][lo:hi], g["v_big"][lo:hi], g["a_big"][lo:hi], 0, 0)
    tau = np.ascontiguousarray(g["tau"].reshape(6, N)[:, lo:hi]).reshape(-1)
    keep = [i for i in range(W.shape[1]) if i not in set(g["idx_e"].tolist())]
    Wb = W[:, keep][:, g["idx_base"]]
    colsq, G, gv, tt, rows = fd.allreduce_normal_terms(ex, oracle_c.colsq(W), Wb.T @ Wb, Wb.T @ tau, tau @ tau, len(tau))
    tmax = fd.allgather_max(ex, np.max(tau))
    phi = np.linalg.solve(G, gv)
    std = relative_stdev_from_normal_terms(G, gv, tt, rows, phi)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), colsq=colsq, G=G, g=gv, tt=tt, rows=rows, tmax=tmax, phi=phi, std=std)
    ex.barrier()
    if plane == "torch":
        dist.destroy_process_group()
    else:
        ex.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("plane,world", [("torch", 2), ("socket", 3)])
def test_normal_terms_allreduce_matches_single_process(tmp_path, oracle_lib, plane, world):
    """Collective (1) of SURVEY 8e: [colsq | G | W^T tau | tau^T tau | rows] summed over UNEQUAL sample shards in one
    all-reduce (dist.allreduce_normal_terms) equals the single-process quantities on the whole sample set -- bit-identical
    on every rank -- and what its consumers derive from it: the normal-equation solution (the golden pinv solution to 1e-8),
    relative_stdev (identification_tools.py:204-234) and max(tau) of the SIP scaling (:528-531)."""
    import multiprocessing as mp
    import oracle_np
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_normal_terms_worker, args=(r, world, port, str(tmp_path), plane)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(200)
        assert p.exitcode == 0
    res = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    g = np.load(os.path.join(GOLD, "cfg2_ur10.npz"))
    for r in res[1:]:
        for k in ("colsq", "G", "g", "tt", "rows", "tmax", "phi", "std"):
            assert np.array_equal(r[k], res[0][k]), k
    r0 = res[0]
    assert np.abs(r0["colsq"] - g["colsq_big"]).max() <= 1e-12 * g["colsq_big"].max()
    from figaroh_plus_amd.model import Model
    flat = Model.from_flat(os.path.join(ROOT, "figaroh_plus_amd", "models", "ur10.json")).to_flat()
    W = oracle_lib.OracleModel(flat).build_regressor_basic(g["q_big"], g["v_big"], g["a_big"], 0, 0)
    tau = g["tau"]
    assert float(r0["rows"]) == len(tau) and abs(float(r0["tt"]) - tau @ tau) <= 1e-12 * (tau @ tau)
    assert float(r0["tmax"]) == np.max(tau)
    assert np.abs(r0["phi"] - g["phi_pinv"]).max() <= 1e-8 * np.abs(g["phi_pinv"]).max()
    if W is not None:
        keep = [i for i in range(W.shape[1]) if i not in set(g["idx_e"].tolist())]
        Wb = W[:, keep][:, g["idx_base"]]
        assert np.abs(r0["G"] - Wb.T @ Wb).max() <= 1e-12 * np.abs(r0["G"]).max()
        ref_std = oracle_np.relative_stdev(Wb, r0["phi"], tau)
        assert np.abs(r0["std"] - ref_std).max() <= 0.011
