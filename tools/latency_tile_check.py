"""Diagnostic: the 8-wave latency tile of k_admm_res2 (batches <= #CUs) against the 4-wave throughput tile (forced with low_memory=True,
which costs it ~1 %) on the headline problem size: kernel time per launch, iteration counts, x."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
dev = torch.device("cuda:0")
n, ne, ni = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 25, 275)
for B in (1, 16, 128, 256, 512):
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, ne, ni, seed0=0, feasible=True, dtype=np.float32)
    out = {}
    for tag, kw in (("latency", {}), ("throughput", dict(low_memory=True))):
        m = reluqpth.ReLU_QP()
        m.setup(H, g, A, l, u, device=dev, precision=torch.float32, warm_starting=False, kernel="resident", full_ladder=True, **kw)
        ks = []
        for _ in range(6):
            r = m.solve(); ks.append(m.last_kernel_time * 1e3)
        out[tag] = (min(ks[1:]), r.x.clone(), r.info.iter.clone())
        del m
    (tl, xl, il), (tt, xt, it_) = out["latency"], out["throughput"]
    same = il == it_
    print("B=%4d  default %.3f ms  4-wave tile %.3f ms  ratio %.2f | same iterations %.3f max|dx| %.2e  max it %d" % (
        B, tl, tt, tt / tl, same.float().mean().item(), float((xl - xt)[same].abs().max()), int(il.max())), flush=True)
