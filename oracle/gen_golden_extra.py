"""Extra golden vectors (run in the authoring container, where /root/reference exists):

* ``QR_pivoting`` (src/figaroh/tools/qrdecomposition.py:24-86): the reference's own function on the UR10 and TX40
  fixtures (W rebuilt bit-exactly by the NumPy oracle from the committed samples), including the quirk that the rank
  loop leaves numrank_W = 0 when no pivot falls below the tolerance (full-rank input -> empty result);
* ``calculate_first_second_order_differentiation`` (identification_tools.py:334-387) on the human model (free-flyer
  root): the reference's own function with ``pin.difference`` stubbed by oracle_np.joint_difference (Pinocchio is not
  installable here; the SE(3) logarithm behind it is checked against scipy.linalg.logm in tests/test_oracle.py).

Writes tests/golden/qr_pivoting.json and tests/golden/human_differentiation.npz.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import gen_golden as gg  # noqa: E402  (sets up the pinocchio stubs and loads the reference modules by path)
import oracle_np  # noqa: E402
from figaroh_plus_amd.model import Model  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def qr_pivoting_fixture():
    out = {}
    for cfg, mname in (("cfg2_ur10", "ur10"), ("cfg1_tx40", "tx40")):
        meta = json.load(open(os.path.join(GOLD, cfg + ".json")))
        z = np.load(os.path.join(GOLD, cfg + ".npz"))
        flat = Model.from_flat(os.path.join(ROOT, "figaroh_plus_amd", "models", mname + ".json")).to_flat()
        W = oracle_np.build_regressor_basic(flat, z["q_big"], z["v_big"], z["a_big"], meta["param"])
        if meta["coupling"]:
            W = oracle_np.add_coupling_TX40(W, len(z["q_big"]), z["v_big"], z["a_big"])
        keep = [i for i in range(W.shape[1]) if i not in set(z["idx_e"].tolist())]
        W_e, params_r, tau = W[:, keep], meta["params_r"], z["tau"]
        W_b, base_parameters = gg.ref_qr.QR_pivoting(tau, W_e, list(params_r))
        # full-rank input: the reference's loop never reaches its `else`, numrank_W stays 0
        W_full = W_e[:, z["idx_base"]]
        names_full = [params_r[i] for i in z["idx_base"]]
        W_b0, base0 = gg.ref_qr.QR_pivoting(tau, W_full, list(names_full))
        out[cfg] = {"expressions": list(base_parameters.keys()), "phi_b": [float(x) for x in base_parameters.values()],
                    "W_b_shape": list(W_b.shape), "W_b_checksum": [float(W_b.sum()), float(np.abs(W_b).sum())],
                    # tie-break independent invariants (the pivot order among equal trailing norms is roundoff noise:
                    # LAPACK on W_e perturbed by 1e-15 relative already changes P): the fitted torques and the residual
                    "prediction_norm": float(np.linalg.norm(W_b @ np.array(list(base_parameters.values())))),
                    "residual_norm": float(np.linalg.norm(tau - W_b @ np.array(list(base_parameters.values())))),
                    "W_b_colnorm_max": float(np.linalg.norm(W_b, axis=0).max()),
                    "full_rank_result": {"W_b_shape": list(W_b0.shape), "n_parameters": len(base0)}}
        print(cfg, "QR_pivoting:", len(base_parameters), "parameters; full-rank input ->", W_b0.shape, len(base0))
    with open(os.path.join(GOLD, "qr_pivoting.json"), "w") as f:
        json.dump(out, f, indent=1)


def human_differentiation_fixture():
    model = Model.from_flat(os.path.join(ROOT, "figaroh_plus_amd", "models", "human.json"))
    flat = model.to_flat()
    gg.pin.difference = lambda m, q0, q1: oracle_np.joint_difference(flat, q0, q1)
    rng = np.random.default_rng(20250410 + 77)
    T, ts = 14, 0.01
    # a smooth whole-body motion: free-flyer drifting and rotating, joints on sines
    t = np.arange(T) * ts
    q = np.zeros((T, model.nq))
    q[:, :3] = rng.uniform(-1, 1, 3) + np.outer(t, rng.uniform(-1, 1, 3)) + np.outer(t ** 2, rng.uniform(-2, 2, 3))
    ax = rng.standard_normal(3)
    ax /= np.linalg.norm(ax)
    ang = 0.3 + 2.0 * t + 3.0 * t ** 2
    q[:, 3:6] = np.sin(ang / 2)[:, None] * ax
    q[:, 6] = np.cos(ang / 2)
    amp, freq, ph = rng.uniform(0.1, 0.8, model.nq - 7), rng.uniform(1, 6, model.nq - 7), rng.uniform(0, 6, model.nq - 7)
    q[:, 7:] = amp * np.sin(np.outer(t, freq) + ph)
    param = {"is_joint_torques": False, "is_external_wrench": True, "ts": ts}
    q2, dq, ddq = gg.ref_idt.calculate_first_second_order_differentiation(model, q.copy(), param)
    dt = np.full(T - 1, ts) * (1.0 + 0.1 * rng.uniform(-1, 1, T - 1))
    q3, dq3, ddq3 = gg.ref_idt.calculate_first_second_order_differentiation(model, q.copy(), param, dt=dt)
    np.savez_compressed(os.path.join(GOLD, "human_differentiation.npz"), q=q, ts=ts, q_out=q2, dq=dq, ddq=ddq, dt=dt,
                        q_out_dt=q3, dq_dt=dq3, ddq_dt=ddq3)
    print("human differentiation:", q2.shape, dq.shape, ddq.shape, "max |dq|", np.abs(dq).max())


if __name__ == "__main__":
    qr_pivoting_fixture()
    human_differentiation_fixture()
