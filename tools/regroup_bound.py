"""Diagnostic (not product): upper bound of what grouping the tiles of a COLD sparse-MPC batch by rho path could buy -- the batch is
re-ordered on the host by keys taken from a previous solve of the same problems (index after the first check, final index, iterations)."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import mpc
dev = torch.device("cuda:0")
Ad, Bd = mpc.random_plant(12, 4, seed=0)
form = sys.argv[1] if len(sys.argv) > 1 else "sparse"
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form=form)
B = 4096
x0 = np.random.RandomState(1).randn(B, 12)
g, l, u = ctl.qp_vectors(x0)
def run(perm, tag):
    m = reluqpth.ReLU_QP()
    m.collect_trace = True
    m.setup(ctl.H, g[perm], ctl.A, l[perm], u[perm], device=dev, precision=torch.float32, eps_abs=1e-3, warm_starting=False)
    ks = []
    for _ in range(4):
        r = m.solve(); ks.append(m.last_kernel_time * 1e3)
    print("%-40s kernel %.3f ms (min of 3)  mean it %.1f" % (tag, min(ks[1:]), r.info.iter.float().mean().item()), flush=True)
    return m, r
ident = np.arange(B)
m, r = run(ident, "grid order")
tr = m.last_trace.cpu().numpy()
it = r.info.iter.cpu().numpy(); rf = r.info.rho_ind.cpu().numpy()
after1 = np.where(np.isnan(tr[:, 1, 3]), rf, tr[:, 1, 3]).astype(int)
run(np.argsort(after1, kind="stable"), "sorted by index after check 1")
run(np.lexsort((it, after1)), "by index after check 1, then iterations")
run(np.argsort(it, kind="stable"), "sorted by iteration count")
run(np.lexsort((after1, it)), "by iterations, then index")
