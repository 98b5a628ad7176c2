"""Diagnostic (not product): per-segment s_memtime cycles of k_admm_res64 (RQP_DIAG=1 build).  python tools/res64_diag.py"""
import os, sys
os.environ["RQP_DIAG"] = "1"
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
for B in (256, 1024):
    H, g, A, l, u, _ = utils.rand_qp_batch(B, 100, 25, 275, seed0=0, feasible=True)
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=torch.device("cuda:0"), precision=torch.float64, warm_starting=False)
    print("B=%d kernel=%s" % (B, m.kernel), file=sys.stderr, flush=True)
    r = m.solve()
    torch.cuda.synchronize()
    print("B=%d mean iters %.1f" % (B, float(r.info.iter.double().mean())), file=sys.stderr, flush=True)
