"""Diagnostic (not product): per-segment s_memtime cycles of k_admm_res2 (RQP_DIAG=1 build) at 1 and 2 workgroups per CU and at
the bench batch.  python tools/res2_diag.py [n n_eq n_ineq]"""
import os, sys, time
os.environ["RQP_DIAG"] = "1"
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
n, n_eq, n_ineq = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 25, 275)
for B in (256, 512, 2048):
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=0, feasible=True, dtype=np.float32)
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=torch.device("cuda:0"), precision=torch.float32, warm_starting=False, kernel="resident")
    print("B=%d kernel=%s" % (B, m.kernel), file=sys.stderr, flush=True)
    r = m.solve()
    torch.cuda.synchronize()
    print("B=%d mean iters %.1f" % (B, float(r.info.iter.double().mean())), file=sys.stderr, flush=True)
