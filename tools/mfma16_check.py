"""Diagnostic: the bf16-plane MFMA kernel (tile_dtype = RQP_TILE_BF16) against the float32 MFMA kernel on config-3 batches."""
import os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import mpc
dev = torch.device("cuda:0")
Ad, Bd = mpc.random_plant(12, 4, seed=0)
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="condensed")
for B in [int(v) for v in (sys.argv[1:] or ["64", "4096", "65536"])]:
    x0 = np.random.RandomState(1).randn(B, 12)
    g, l, u = ctl.qp_vectors(x0)
    out = {}
    for tile in (None, torch.bfloat16):
        m = reluqpth.ReLU_QP()
        m.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float32, eps_abs=1e-3, warm_starting=False, kernel="mfma", iterate_dtype=tile)
        ks = []
        for _ in range(4):
            r = m.solve(); ks.append(m.last_kernel_time * 1e3)
        out[tile] = (r.x.clone(), r.info.iter.clone(), r.info.status_code.clone(), r.info.pri_res.clone(), r.info.dua_res.clone())
        print("B=%d %-6s kernel %s ms  %.2f M QP/s  mean it %.1f solved %.3f" % (B, m.kernel, " ".join("%.3f" % k for k in ks), B / min(ks[1:]) / 1e3,
              r.info.iter.float().mean().item(), (r.info.status_code == 0).float().mean().item()), flush=True)
        del m
    xa, ita, sa, pa, da = out[None]
    xb, itb, sb, pb, db = out[torch.bfloat16]
    same = ita == itb
    print("   same iterations %.4f  max|dit| %d  max|dx| (same) %.2e  max|x| %.2f  nan %s" % (same.float().mean().item(), int((ita - itb).abs().max()),
          float((xa - xb)[same].abs().max()) if bool(same.any()) else -1.0, float(xa.abs().max()), bool(torch.isnan(xb).any())), flush=True)
