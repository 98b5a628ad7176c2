"""Diagnostic (not product, not part of the test suite): randomised A/B sweep of the specialised ADMM kernels against
the streaming kernel (k_admm_generic) -- many shapes, batch sizes and settings, exits and solutions must agree.
    python tools/sweep_kernels.py [n_cases]
"""
import os
import sys

import numpy as np
import torch

R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import reluqp.reluqpth as reluqpth  # noqa: E402
from reluqp import mpc, utils  # noqa: E402


def solve(H, g, A, l, u, prec, kernel, **kw):
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=torch.device("cuda:0"), precision=prec, kernel=kernel, **kw)
    return m, m.solve()


def compare(tag, ra, rb, prec):
    ia, ib = ra.info.iter.cpu().numpy(), rb.info.iter.cpu().numpy()
    sa, sb = ra.info.status, rb.info.status
    same = ia == ib
    tol = 1e-9 if prec == torch.float64 else 2e-4
    scale = max(1.0, float(rb.x.abs().max()))
    dx = float((ra.x - rb.x).abs()[torch.from_numpy(same)].max()) if same.any() else 0.0
    # (a residual sitting on the threshold ends one solve a check before the other: allow one such instance in small batches)
    ok = (sa == sb or np.mean(np.array(sa) == np.array(sb)) > 0.97) and (np.mean(same) >= 0.85 or (~same).sum() <= 1) \
        and dx <= tol * scale
    print("%-44s %s  same-iter %.3f  max|dx| %.2e  mean it %.1f" % (tag, "ok " if ok else "FAIL", np.mean(same), dx, ia.mean()))
    return ok


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rs = np.random.RandomState(2024)
    bad = 0
    for case in range(ncase):
        # ---- one-wavefront kernel vs streaming kernel
        n = int(rs.randint(2, 33) if rs.rand() < 0.7 else rs.randint(57, 65))     # 57..64: the two-wavefront variant
        n_eq = int(rs.randint(0, max(1, min(n, 8))))
        m = int(rs.randint(max(n_eq + 1, 2), 65 if rs.rand() < 0.5 else 129))     # both one-wavefront variants
        B = int(rs.choice([1, 3, 17, 64, 300]))
        prec = torch.float32 if (rs.rand() < 0.6 or m > 64 or n > 32) else torch.float64
        eps = float(rs.choice([1e-3, 1e-4]))
        H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, m - n_eq, seed0=1000 * case, feasible=True)
        mw, rw = solve(H, g, A, l, u, prec, "auto", eps_abs=eps)
        mg, rg = solve(H, g, A, l, u, prec, "generic", eps_abs=eps)
        assert mw.kernel == "wave" and mg.kernel == "generic", (mw.kernel, mg.kernel)
        bad += not compare("wave    n=%d m=%d (eq %d) B=%d %s eps %g" % (n, m, n_eq, B, str(prec)[6:], eps), rw, rg, prec)
    for case in range(max(4, ncase // 3)):
        # ---- MFMA kernel vs per-instance kernels on linear-MPC batches
        nx, nu = int(rs.randint(2, 13)), int(rs.randint(1, 5))
        N = int(rs.randint(3, 80 // nu + 1))
        while N * nu > 80 or N * (nx + nu) > 320:
            N -= 1
        B = int(rs.choice([5, 16, 100, 700]))
        Ad, Bd = mpc.random_plant(nx, nu, seed=case)
        try:
            ctl = mpc.LinearMPC(Ad, Bd, np.eye(nx), 0.1 * np.eye(nu), N, 0.5, 10.0, form="condensed")
        except Exception as e:          # the Riccati iteration of the generator gives up on some random plants (as upstream)
            print("mfma    nx=%d nu=%d: plant skipped (%s)" % (nx, nu, e))
            continue
        x0 = rs.randn(B, nx)
        g, l, u = ctl.qp_vectors(x0)
        mm, rm = solve(ctl.H, g, ctl.A, l, u, torch.float32, "mfma", eps_abs=1e-3)
        mr, rr = solve(ctl.H, g, ctl.A, l, u, torch.float32, "resident", eps_abs=1e-3)
        assert mm.kernel == "mfma", mm.kernel
        bad += not compare("mfma    nx=%d nu=%d N=%d (n=%d m=%d) B=%d vs %s" % (nx, nu, N, N * nu, N * (nx + nu), B, mr.kernel), rm, rr, torch.float32)
    print("failures:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
