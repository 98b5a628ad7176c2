"""Diagnostic: iteration counts per instance of the bench workloads on the GPU (saved under gpurun_out/ for the scheduling
simulator tools/sched_sim.py)."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
dev = torch.device("cuda:0")
os.makedirs(os.path.join(R0, "gpurun_out"), exist_ok=True)
for tag, B, n, ne, ni, prec in (("c2_f32", 4096, 100, 25, 275, torch.float32), ("c4_f32", 8192, 32, 8, 56, torch.float32),
                                ("c2_f64", 4096, 100, 25, 275, torch.float64)):
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, ne, ni, seed0=0, feasible=True, dtype=np.float32 if prec == torch.float32 else np.float64)
    m = reluqpth.ReLU_QP()
    m.setup(*[torch.from_numpy(a).to(dev) for a in (H, g, A, l, u)], device=dev, precision=prec, warm_starting=False)
    ks = []
    for _ in range(4):
        r = m.solve(); ks.append(m.last_kernel_time * 1e3)
    it = r.info.iter.cpu().numpy()
    ri = r.info.rho_ind.cpu().numpy()
    np.save(os.path.join(R0, "gpurun_out", "iters_%s.npy" % tag), it)
    print(tag, m.kernel, "kernel ms", " ".join("%.3f" % k for k in ks), "mean", it.mean(), flush=True)
    print("  iters", dict(zip(*[x.tolist() for x in np.unique(it, return_counts=True)])))
    print("  rho_ind", dict(zip(*[x.tolist() for x in np.unique(ri, return_counts=True)])))
    print("  long ones at", np.nonzero(it > 300)[0].tolist()[:40], flush=True)
    del m
