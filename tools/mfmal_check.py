"""Diagnostic: the large / sparse shared-matrix MFMA kernel (k_admm_mfmal) on the sparse linear-MPC form against the streaming
kernel and the oracle.  usage: mfmal_check.py [B ...]"""
import os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
sys.path.insert(0, R0)
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import mpc
dev = torch.device("cuda:0")
Ad, Bd = mpc.random_plant(12, 4, seed=0)
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="sparse")
print("n %d m %d  nnz(A) %.3f nnz(H) %.3f" % (ctl.H.shape[0], ctl.A.shape[0], (ctl.A != 0).mean(), (ctl.H != 0).mean()), flush=True)
for B in [int(v) for v in (sys.argv[1:] or ["64", "4096"])]:
    x0 = np.random.RandomState(1).randn(B, 12)
    g, l, u = ctl.qp_vectors(x0)
    out = {}
    for kern in ("mfma", "generic"):
        m = reluqpth.ReLU_QP()
        m.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float32, eps_abs=1e-3, warm_starting=False, kernel=kern)
        ks = []
        for _ in range(3):
            r = m.solve(); ks.append(m.last_kernel_time * 1e3)
        out[kern] = (r.x.clone(), r.info.iter.clone(), r.info.status_code.clone(), r.info.pri_res.clone(), r.info.dua_res.clone())
        out[kern + "_ri"] = r.info.rho_ind.clone()
        print("B=%d %-8s kernel %s ms  %.3f M QP/s  mean it %.1f max it %d solved %.3f setup %.1f ms" % (B, m.kernel, " ".join("%.3f" % k for k in ks), B / min(ks[1:]) / 1e3,
              r.info.iter.float().mean().item(), int(r.info.iter.max()), (r.info.status_code == 0).float().mean().item(), m.results.info.setup_time * 1e3), flush=True)
        del m
    xa, ita, sa, pa, da = out["mfma"]
    xb, itb, sb, pb, db = out["generic"]
    same = ita == itb
    print("   same iterations %.4f  max|dit| %d  max|dx| (same) %.2e  max|x| %.2f  nan %s  pri %.2e/%.2e dua %.2e/%.2e" % (same.float().mean().item(), int((ita - itb).abs().max()),
          float((xa - xb)[same].abs().max()) if bool(same.any()) else -1.0, float(xb.abs().max()), bool(torch.isnan(xa).any()),
          float(pa.max()), float(pb.max()), float(da.max()), float(db.max())), flush=True)
    dxi = (xa - xb).abs().amax(1)
    big = (dxi > 1e-3) & same
    ria, rib = out["mfma_ri"], out["generic_ri"]
    print("   instances with same iterations and |dx| > 1e-3: %d; of those with a different final rho index: %d; different rho index overall: %d"
          % (int(big.sum()), int((big & (ria != rib)).sum()), int((ria != rib).sum())), flush=True)
    if B <= 64:
        from oracle import reluqp_oracle as O
        ref = O.solve_batch(ctl.H, g[:8], ctl.A, l[:8], u[:8], form="factored", eps_abs=1e-3)
        print("   oracle it", ref["iter"], "mfmal it", ita[:8].cpu().numpy(), "max|dx|", np.abs(xa[:8].cpu().double().numpy() - ref["x"]).max(), flush=True)
