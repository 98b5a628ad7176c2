"""Diagnostic (not product): do cheap per-instance keys predict the rho-index path of a linear-MPC batch?  (Columns of an MFMA tile
that sit at different rho indices cost one pass of the dense K stream each: k_admm_mfmal averages 1.335 passes per iteration.)"""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import mpc
dev = torch.device("cuda:0")
form = sys.argv[1] if len(sys.argv) > 1 else "sparse"
Ad, Bd = mpc.random_plant(12, 4, seed=0)
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form=form)
B = 4096
x0 = np.random.RandomState(1).randn(B, 12)
g, l, u = ctl.qp_vectors(x0)
m = reluqpth.ReLU_QP()
m.collect_trace = True
m.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float32, eps_abs=1e-3, warm_starting=False)
r = m.solve()
tr = m.last_trace.cpu().numpy()                     # [B][checks][pri, dua, est, ri_before]
it = r.info.iter.cpu().numpy()
ri_fin = r.info.rho_ind.cpu().numpy()
nchk = tr.shape[1]
path = np.full((B, 8), -1)
for c in range(min(8, nchk)):
    v = tr[:, c, 3]
    path[:, c] = np.where(np.isnan(v), -1, v).astype(int)
# index after check c = ri_before of check c + 1 (or the final index)
after = np.concatenate([path[:, 1:], -np.ones((B, 1), int)], 1)
for b in range(B):
    k = it[b] // 25 - 1
    if 0 <= k < 8:
        after[b, k] = ri_fin[b]
def passes(order):
    tot = 0.0; cnt = 0
    for t0 in range(0, B, 16):
        ids = order[t0:t0 + 16]
        for c in range(8):                           # iterations 25c+1 .. 25(c+1): live columns at the index after check c-1
            live = it[ids] > 25 * c
            if not live.any():
                break
            idx = np.full(len(ids), 7) if c == 0 else after[ids, c - 1]
            tot += len(np.unique(idx[live])); cnt += 1
    return tot / cnt
keys = {"grid order": np.arange(B), "|x0|": np.argsort(np.linalg.norm(x0, axis=1)), "|g|inf": np.argsort(np.abs(g).max(1)),
        "|l,u|inf (finite)": np.argsort(np.where(np.isfinite(l), np.abs(l), 0).max(1)),
        "oracle: index after check 1": np.argsort(after[:, 0], kind="stable"), "oracle: iteration count": np.argsort(it, kind="stable")}
print("form %s, kernel %s: index after check 1: %s" % (form, m.kernel, dict(zip(*np.unique(after[:, 0], return_counts=True)))))
for name, order in keys.items():
    print("   tiles formed by %-32s -> %.3f K passes per tile-iteration" % (name, passes(order)))
