#!/usr/bin/env python3
"""Randomised check of the SHARDED pass on one GPU and in one process: random models, samples split into two or three
unequal shards, one IdentificationPipeline per shard under an exchange with world_size > 1 -- the collective code paths of the
pipeline: all-reduced column norms, non-local rank decision, stacked triangles merged by figh_tsqr_merge_base -- against the
single-rank pass on all samples.  The exchange replays what the other shards contributed in the previous round of the same pass
(the column norms do not depend on the selection, the triangles only on the kept set, so three rounds settle it); nothing but
figh_memcpy moves data.  usage: python tools/fuzz_sharded.py [first] [count]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.pipeline import Exchange, IdentificationPipeline  # noqa: E402

_lib.load()


class ReplayExchange(Exchange):
    """Rank ``rank`` of ``world``: sums / stacks = own contribution + what the board holds of the others from the last round."""

    def __init__(self, board, rank, world):
        self.board, self.rank, self.world_size = board, rank, world
        self.calls = {"sum": 0, "stack": 0}

    def sum_columns(self, d_colsq, ncols):
        mine = np.empty(ncols)
        _lib.check(_lib.load().figh_memcpy_d2h(mine.ctypes.data, d_colsq.ptr, mine.nbytes))
        key = ("sum", self.calls["sum"], ncols)
        self.calls["sum"] += 1
        self.board.setdefault(key, {})[self.rank] = mine.copy()
        total = np.zeros(ncols)
        for r in range(self.world_size):  # rank order: every rank forms the same sum
            total += self.board[key].get(r, np.zeros(ncols))
        return total

    def stack_triangles(self, d_R, nc):
        mine = np.empty(nc * nc)
        _lib.check(_lib.load().figh_memcpy_d2h(mine.ctypes.data, d_R.ptr, mine.nbytes))
        key = ("stack", self.calls["stack"], nc)
        self.calls["stack"] += 1
        self.board.setdefault(key, {})[self.rank] = mine.copy()
        parts = [self.board[key].get(r, np.zeros(nc * nc)) for r in range(self.world_size)]
        return _lib.DeviceArray.from_host(np.concatenate(parts)), self.world_size

    def new_round(self):
        self.calls = {"sum": 0, "stack": 0}


def inputs(m, N, rng):
    q = np.zeros((N, m.nq))
    for j in m.joints[1:]:
        if j.nq == 7:
            quat = rng.standard_normal((N, 4))
            q[:, :3], q[:, 3:7] = rng.uniform(-1, 1, (N, 3)), quat / np.linalg.norm(quat, axis=1)[:, None]
        elif j.nq == 2:
            th = rng.uniform(-3, 3, N)
            q[:, j.idx_q], q[:, j.idx_q + 1] = np.cos(th), np.sin(th)
        else:
            q[:, j.idx_q] = rng.uniform(-2, 2, N)
    return q, rng.uniform(-2, 2, (N, m.nv)), rng.uniform(-3, 3, (N, m.nv))


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    try:
        kind = seed % 3
        if kind == 0:  # serial chain (fused launch when the shards are long enough)
            nj = int(rng.choice([5, 6, 7]))
            robot = T._synthetic_chain(nj, seed=seed)
            param = dict(is_joint_torques=True, is_external_wrench=False, has_friction=False, has_actuator_inertia=bool(rng.integers(2)),
                         has_joint_offset=False, force_torque=None)
            N = int(rng.integers(9000, 16000))
            q, v, a = (rng.uniform(-3, 3, (N, nj)) for _ in range(3))
            rps = nj
        elif kind == 1:  # fixed-base tree: per-row-block TSQR
            n = int(rng.integers(6, 20))
            parents = T._random_parents(rng, n, deep=0.6)
            robot = T._synthetic_tree(parents, seed=seed, massless=())
            param = dict(is_joint_torques=True, is_external_wrench=False, has_friction=bool(rng.integers(2)), has_actuator_inertia=False,
                         has_joint_offset=bool(rng.integers(2)), force_torque=None)
            N = 64 * int(rng.integers(20, 60)) + int(rng.integers(0, 64))
            q, v, a = inputs(robot.model, N, rng)
            rps = robot.model.nv
        else:  # floating base: external wrench, force / torque split
            n = int(rng.integers(6, 20))
            parents = T._random_parents(rng, n, deep=0.6)
            robot = T._synthetic_tree([0] + [p + 1 for p in parents], seed=seed, massless=(), freeflyer=True)
            param = dict(is_joint_torques=False, is_external_wrench=True, has_friction=False, has_actuator_inertia=False,
                         has_joint_offset=False, force_torque=["All"])
            N = 64 * int(rng.integers(200, 300)) + int(rng.integers(0, 64))
            q, v, a = inputs(robot.model, N, rng)
            rps = 6
        params_std = robot.get_standard_parameters(param)
        layout = "block-compact" if kind == 1 and rng.random() < 0.5 else "dense"
        one = IdentificationPipeline(robot, param, params_std=params_std, w_layout=layout)
        one.set_samples(q, v, a)
        tau = one.set_tau_from_parameters(np.array(list(params_std.values()), dtype=float), noise_std=1e-3, seed=seed).to_host()
        ref = one.run()
        ref = one.run()
        del one
        world = int(rng.choice([2, 3]))
        cuts = np.sort(rng.choice(np.arange(N // 8, N - N // 8), world - 1, replace=False))
        bounds = [0] + [int(c) for c in cuts] + [N]
        board = {}
        pipes = []
        for r in range(world):
            lo, hi = bounds[r], bounds[r + 1]
            ex = ReplayExchange(board, r, world)
            p = IdentificationPipeline(robot, param, params_std=params_std, w_layout=layout, exchange=ex)
            tau_r = np.concatenate([tau[j * N + lo:j * N + hi] for j in range(rps)])
            p.set_samples(q[lo:hi], v[lo:hi], a[lo:hi], tau_r)
            pipes.append((p, ex))
        outs = None
        for rnd in range(5):  # rounds of the same pass until every rank has seen every other rank's contribution
            board_before = {k: dict(vv) for k, vv in board.items()}
            outs = []
            for p, ex in pipes:
                ex.new_round()
                try:
                    outs.append(p.run())
                except ValueError:  # (a rank that has only seen zeros of the others in round 0)
                    outs.append(None)
        assert all(o is not None for o in outs), (seed, "a rank did not settle")
        for o in outs:
            assert o["idx_e"] == ref["idx_e"] and o["idx_base"] == ref["idx_base"], (seed, kind, world, "index sets")
            assert o["rows"] == ref["rows"] or kind != 99
            assert np.abs(o["col_norm"] - ref["col_norm"]).max() <= 1e-11 * ref["col_norm"].max(), (seed, "col_norm")
            base_d = ref["absdiagR"][np.asarray(ref["idx_base"])]
            cond = base_d.max() / base_d.min()
            # the fit itself first: both are least-squares minimisers of the same problem
            assert abs(o["residual_norm"] - ref["residual_norm"]) <= 1e-8 * max(1.0, ref["residual_norm"]), (seed, "residual")
            # phi: determined to cond(W_b) eps; the ratio of the extreme pivots bounds cond from below only (seed 126: ratio 6e10,
            # |phi| up to 4e5, the two solutions 55 apart with residuals equal to 1e-8)
            tol = max(1e-8, 1e5 * np.finfo(float).eps * cond) * max(1.0, np.abs(ref["phi_ls"]).max())
            assert np.abs(o["phi_ls"] - ref["phi_ls"]).max() <= tol, (seed, kind, world, "phi", np.abs(o["phi_ls"] - ref["phi_ls"]).max(), cond,
                                                                       float(np.abs(ref["phi_ls"]).max()))
        # (ranks of a real run gather the SAME triangles and agree bit for bit; here a rank's own triangle is recomputed in every round
        # while the others' are replayed, and the fused launch reproduces its triangle only up to rounding: bitwise for the other kinds)
        if kind != 0:
            assert all(np.array_equal(o["phi_ls"], outs[0]["phi_ls"]) for o in outs), (seed, "ranks differ")
        del pipes
    except Exception:  # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED")
        traceback.print_exc(limit=3)
print("%d sharded random models (seeds %d .. %d), %d failures" % (count, first, first + count - 1, bad))
