"""Diagnostic: how much of the headline launch is TAIL (the last workgroups running alone)?  Solves the bench batch in natural
order, then with the instances sorted by descending iteration count (longest-processing-time-first), prints both kernel times."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
B = 4096
H, g, A, l, u, _ = utils.rand_qp_batch(B, 100, 25, 275, seed0=0, feasible=True, dtype=np.float32)
dev = torch.device("cuda:0")
def run(order, tag):
    t = [torch.from_numpy(np.ascontiguousarray(a[order])).to(dev) for a in (H, g, A, l, u)]
    m = reluqpth.ReLU_QP()
    m.setup(*t, device=dev, precision=torch.float32, warm_starting=False)
    ks = []
    for _ in range(6):
        r = m.solve(); ks.append(m.last_kernel_time * 1e3)
    it = r.info.iter.cpu().numpy()
    print("%-10s kernel ms: %s  mean it %.1f max %d" % (tag, " ".join("%.3f" % k for k in ks[1:]), it.mean(), it.max()), flush=True)
    return it
it = run(np.arange(B), "natural")
run(np.argsort(-it, kind="stable"), "LPT")
run(np.argsort(it, kind="stable"), "SPT")
