import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import Golden
from figaroh_plus_amd import _lib as lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
g = Golden("cfg3_tiago")
pipe = IdentificationPipeline(g.robot(), g.param, params_std=g.params_std(), coupling=g.coupling)
pipe.set_samples(g["q_big"], g["v_big"], g["a_big"], g["tau"])
try:
    pipe.run()
except Exception as e:
    print("ERR", e)
W = pipe.W
ncols = W.ref_cols
selw = pipe._sel_words
host = np.empty(ncols + selw)
lib.check(lib.load().figh_memcpy_d2h(host.ctypes.data, pipe._d_pack.ptr, host.nbytes))
sel = host[ncols:].view(np.int32)
n = int(sel[0])
kept_ref = [i for i in range(ncols) if i not in set(g["idx_e"].tolist())]
dev_list = sel[2:2 + n]
exp = (np.array(kept_ref) // 14) * 16 + np.array(kept_ref) % 14
print("n", n, "list ok", np.array_equal(dev_list, exp), "colsq ok", np.abs(host[:ncols] - g["colsq_big"]).max() / g["colsq_big"].max())
# plain R via the host-list path
d_idx = lib.DeviceArray.from_host(exp.astype(np.int32))
nc = n + 1
d_R = lib.DeviceArray((nc * nc,))
lib.tsqr(W.buf, W.rows, W.ld, d_idx, n, pipe.d_tau, None, d_R)
R = d_R.to_host().reshape(nc, nc)
d_ref = np.abs(np.diag(R))
print("host-list base == golden", np.flatnonzero(d_ref[:n] > 1e-8).tolist() == list(g["idx_base"]))
# selected, plain
d_sel = lib.DeviceArray((2 + 2 * ncols,), np.int32)
d_R2 = lib.DeviceArray((nc * nc,))
lib.tsqr_selected(W.buf, W.rows, W.ld, pipe._d_colsq, ncols, 1e-6, 16, 24, n, pipe.d_tau, -1.0, d_sel, d_R2)
R2 = d_R2.to_host().reshape(nc, nc)
print("selected plain vs tsqr: max diff", np.abs(np.abs(R2) - np.abs(R)).max(), "diag diff", np.abs(np.abs(np.diag(R2)) - d_ref).max())
d_rows = lib.DeviceArray(((nc + 1) * nc,))
lib.tsqr_selected(W.buf, W.rows, W.ld, pipe._d_colsq, ncols, 1e-6, 16, 24, n, pipe.d_tau, 1e-8, d_sel, d_rows)
o = d_rows.to_host().reshape(nc + 1, nc)
print("reveal diag vs plain", np.abs(np.abs(o[nc]) - d_ref).max(), "base", np.flatnonzero(np.abs(o[nc])[:n] > 1e-8).tolist() == list(g["idx_base"]))
bad = np.flatnonzero(np.abs(np.abs(o[nc]) - d_ref) > 1e-9)
print("bad idx", bad[:10], np.abs(o[nc])[bad[:5]], d_ref[bad[:5]])
