"""Diagnostic (not product): per-segment s_memtime ticks of k_admm_mfmad (RQP_DIAG=1 build) on the condensed config-3 batch, float64."""
import os, sys
os.environ["RQP_DIAG"] = "1"
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import mpc
dev = torch.device("cuda:0")
Ad, Bd = mpc.random_plant(12, 4, seed=0)
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="condensed")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0 = np.random.RandomState(1).randn(B, 12)
g, l, u = ctl.qp_vectors(x0)
m = reluqpth.ReLU_QP()
m.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float64, eps_abs=1e-3, warm_starting=False, kernel="mfma")
print("---- %s B=%d" % (m.kernel, B), file=sys.stderr, flush=True)
m.solve()
torch.cuda.synchronize()
