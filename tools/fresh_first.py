"""Diagnostic: kernel time of the FIRST solve of a fresh handle on fresh problems once the code objects are warm
(separates the cold-code cost of a process's first launch from the grid-order cost of a handle without history)."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
dev = torch.device("cuda:0")
for tag, B, n, ne, ni in (("c2", 4096, 100, 25, 275), ("c4", 8192, 32, 8, 56)):
    for rep in range(3):
        H, g, A, l, u, _ = utils.rand_qp_batch(B, n, ne, ni, seed0=rep * B, feasible=True, dtype=np.float32)
        m = reluqpth.ReLU_QP()
        m.setup(*[torch.from_numpy(a).to(dev) for a in (H, g, A, l, u)], device=dev, precision=torch.float32, warm_starting=False)
        ks = []
        for _ in range(4):
            r = m.solve(); ks.append(m.last_kernel_time * 1e3)
        print(tag, "handle", rep, m.kernel, "kernel ms", " ".join("%.3f" % k for k in ks), "mean it %.1f" % r.info.iter.float().mean().item(), flush=True)
        del m
