"""Diagnostic: the float64 streamed-operand MFMA kernel (k_admm_mfmad) on the condensed linear-MPC form against the float64
resident kernel and the oracle.  usage: mfmad_check.py [B ...]"""
import os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
sys.path.insert(0, R0)
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import mpc
dev = torch.device("cuda:0")
Ad, Bd = mpc.random_plant(12, 4, seed=0)
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="condensed")
print("n %d m %d" % (ctl.H.shape[0], ctl.A.shape[0]), flush=True)
for B in [int(v) for v in (sys.argv[1:] or ["64", "4096"])]:
    x0 = np.random.RandomState(1).randn(B, 12)
    g, l, u = ctl.qp_vectors(x0)
    out = {}
    for kern in ("mfma", "resident"):
        m = reluqpth.ReLU_QP()
        m.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float64, eps_abs=1e-3, warm_starting=False, kernel=kern)
        ks = []
        for _ in range(3):
            r = m.solve(); ks.append(m.last_kernel_time * 1e3)
        out[kern] = (r.x.clone(), r.info.iter.clone(), r.info.status_code.clone(), r.info.pri_res.clone(), r.info.dua_res.clone(), r.info.rho_ind.clone())
        print("B=%d %-10s kernel %s ms  %.3f M QP/s  mean it %.1f max it %d solved %.3f setup %.1f ms" % (B, m.kernel, " ".join("%.3f" % k for k in ks), B / min(ks[1:]) / 1e3,
              r.info.iter.float().mean().item(), int(r.info.iter.max()), (r.info.status_code == 0).float().mean().item(), m.results.info.setup_time * 1e3), flush=True)
        del m
    xa, ita, sa, pa, da, ra = out["mfma"]
    xb, itb, sb, pb, db, rb = out["resident"]
    same = ita == itb
    print("   same iterations %.4f  max|dit| %d  max|dx| (same) %.2e  max|x| %.2f  nan %s  rho_ind differs %d  pri %.3e/%.3e" % (same.float().mean().item(), int((ita - itb).abs().max()),
          float((xa - xb)[same].abs().max()) if bool(same.any()) else -1.0, float(xb.abs().max()), bool(torch.isnan(xa).any()), int((ra != rb).sum()),
          float(pa.max()), float(pb.max())), flush=True)
    if B <= 64:
        from oracle import reluqp_oracle as O
        ref = O.solve_batch(ctl.H, g[:8], ctl.A, l[:8], u[:8], form="factored", eps_abs=1e-3)
        print("   oracle it", ref["iter"], "mfmad it", ita[:8].cpu().numpy(), "max|dx|", np.abs(xa[:8].cpu().numpy() - ref["x"]).max(), flush=True)
