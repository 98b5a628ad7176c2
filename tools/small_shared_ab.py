"""Diagnostic: shared-matrix batches of SMALL problems -- MFMA kernel vs one-wavefront kernel (dispatch threshold)."""
import os, sys, time
import numpy as np, torch
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import reluqp.reluqpth as reluqpth
from reluqp import mpc
for (nx, nu, N) in [(2, 2, 15), (4, 2, 8), (6, 2, 10)]:
    Ad, Bd = mpc.random_plant(nx, nu, seed=1)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(nx), 0.1 * np.eye(nu), N, 0.5, 10.0, form="condensed")
    for B in (4096, 65536):
        x0 = np.random.RandomState(3).randn(B, nx)
        g, l, u = ctl.qp_vectors(x0)
        for kern in ("mfma", "auto"):
            m = reluqpth.ReLU_QP()
            m.setup(ctl.H, g, ctl.A, l, u, device=torch.device("cuda:0"), precision=torch.float32, warm_starting=False, kernel=kern)
            m.solve(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                r = m.solve()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print("n=%d m=%d B=%d %-9s %.2f M QP/s  kernel %.3f ms  mean it %.1f" % (ctl.H.shape[0], ctl.A.shape[0], B, m.kernel, B / dt / 1e6, m.last_kernel_time * 1e3, float(r.info.iter.double().mean())))
