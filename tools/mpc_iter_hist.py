"""Diagnostic: iteration-count histogram of the bench's MPC workload (cold), MFMA kernel vs per-instance resident kernel times."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
import bench
B = 4096
ctl = bench._mpc_controller()
x0 = np.random.RandomState(1).randn(B, 12)
g, l, u = ctl.qp_vectors(x0)
for kern in ("mfma", "resident"):
    m = reluqpth.ReLU_QP()
    m.setup(ctl.H, g, ctl.A, l, u, device=torch.device("cuda:0"), precision=torch.float32, warm_starting=False, kernel=kern)
    ks = []
    for _ in range(5):
        r = m.solve(); ks.append(m.last_kernel_time * 1e3)
    it = r.info.iter.cpu().numpy()
    print(kern, "kernel ms", ["%.3f" % k for k in ks], "mean", it.mean(), "max", it.max())
    print("  hist", {int(k): int((it == k).sum()) for k in np.unique(it)})
