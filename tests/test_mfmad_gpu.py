"""GPU tests of k_admm_mfmad (csrc/rqp_mfmad.hip): shared-(H, A) batches in FLOAT64 -- the reference's default and only working
precision -- on v_mfma_f64_16x16x4_f64, operands streamed from L2 as non-zero 16 x 16 blocks (n <= 160, m <= 320).
Exact float64 FMA chains: checked against the oracle and the float64 resident / streaming kernels to 1e-9 (the summation order
differs), identical iteration counts and rho trajectories.
"""
import numpy as np
import pytest
import torch

from oracle import reluqp_oracle as O
from reluqp import mpc, utils, _cabi
import reluqp.reluqpth as reluqpth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _c3(B, seed=1, form="condensed"):
    Ad, Bd = mpc.random_plant(12, 4, seed=0)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form=form)
    x0 = np.random.RandomState(seed).randn(B, 12)
    g, l, u = ctl.qp_vectors(x0)
    return ctl, ctl.H, g, ctl.A, l, u


def _solve(H, g, A, l, u, kernel="mfma", **kw):
    m = reluqpth.ReLU_QP()
    m.collect_trace = True
    m.setup(H, g, A, l, u, device=DEV, precision=torch.float64, kernel=kernel, **kw)
    return m, m.solve()


def _shared_dense(n, n_eq, n_ineq, B, seed=5):
    H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=seed, compute_sol=False, feasible=True)
    qs = [utils.update_qp(H, A, n_eq, n_ineq, seed=50 + b, compute_sol=False, feasible=True) for b in range(B)]
    g, l, u = (np.stack([q[i] for q in qs]) for i in (1, 3, 4))
    return H, g, A, l, u


def _same(ra, rb, tol=1e-9):
    assert ra.info.status == rb.info.status
    assert torch.equal(ra.info.iter, rb.info.iter) and torch.equal(ra.info.rho_ind, rb.info.rho_ind)
    scale = max(1.0, float(rb.x.abs().max()))
    for fa, fb in ((ra.x, rb.x), (ra.z, rb.z)):
        np.testing.assert_allclose(fa.cpu().numpy(), fb.cpu().numpy(), rtol=0, atol=tol * scale)
    np.testing.assert_allclose(ra.y.cpu().numpy(), rb.y.cpu().numpy(), rtol=0, atol=tol * max(1.0, float(rb.y.abs().max())) * 1e2)
    np.testing.assert_allclose(ra.info.pri_res.cpu().numpy(), rb.info.pri_res.cpu().numpy(), rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(ra.info.obj_val.cpu().numpy(), rb.info.obj_val.cpu().numpy(), rtol=1e-9, atol=1e-9)


def test_c3_float64_full_batch_vs_resident64_and_oracle():
    """BASELINE config 3 (condensed form, n = 80, m = 320) in float64 at the full batch: default dispatch, exact agreement with
    the float64 resident kernel, the oracle on a subset, the check trace of instance 0."""
    B = 4096
    ctl, H, g, A, l, u = _c3(B)
    ma, ra = _solve(H, g, A, l, u, kernel="auto", eps_abs=1e-3, warm_starting=False)
    assert ma.kernel == "mfmad"
    mr, rr = _solve(H, g, A, l, u, kernel="resident", eps_abs=1e-3, warm_starting=False)
    assert mr.kernel == "resident64"
    _same(ra, rr)
    idx = np.unique(np.concatenate([np.arange(0, B, B // 24)[:24], np.arange(B - 8, B)]))
    ref = O.solve_batch(H, g[idx], A, l[idx], u[idx], form="factored", eps_abs=1e-3)
    assert list(np.array(ra.info.status)[idx]) == ref["status"]
    assert np.array_equal(ra.info.iter.cpu().numpy()[idx], ref["iter"])
    np.testing.assert_allclose(ra.x.cpu().numpy()[idx], ref["x"], rtol=0, atol=1e-9 * max(1.0, np.abs(ref["x"]).max()))
    np.testing.assert_allclose(ra.info.rho_estimate.cpu().numpy()[idx], ref["rho_estimate"], rtol=1e-6)
    ta = ma.last_trace[0].cpu().numpy()
    tr = mr.last_trace[0].cpu().numpy()
    ok = ~np.isnan(tr[:, 3])
    np.testing.assert_allclose(ta[ok], tr[ok], rtol=1e-6, atol=1e-12)
    # a smaller batch of the same shape stays on the per-instance float64 kernel
    m2, _ = _solve(H, g[:1024], A, l[:1024], u[:1024], kernel="auto")
    assert m2.kernel == "resident64"


@pytest.mark.parametrize("n,n_eq,n_ineq,B", [(120, 20, 200, 40), (160, 30, 290, 24), (100, 25, 275, 33), (17, 3, 140, 20), (90, 0, 320, 17)])
def test_dense_shared_float64_problems(n, n_eq, n_ineq, B):
    """Dense shared (H, A): beyond the float64 resident tile (n > 104) the kernel is the default at any batch size; equality rows,
    sizes that are not multiples of 16, ragged tiles, the largest shape (160, 320) -- against the oracle and the streaming kernel."""
    H, g, A, l, u = _shared_dense(n, n_eq, n_ineq, B)
    mm, rm = _solve(H, g, A, l, u, kernel="mfma", eps_abs=1e-4)
    assert mm.kernel == "mfmad"
    if n > 104:
        ma, _ = _solve(H, g, A, l, u, kernel="auto", eps_abs=1e-4)
        assert ma.kernel == "mfmad"
    mg, rg = _solve(H, g, A, l, u, kernel="generic", eps_abs=1e-4)
    _same(rm, rg, tol=1e-8)
    nref = min(B, 8)
    ref = O.solve_batch(H, g[:nref], A, l[:nref], u[:nref], form="factored", eps_abs=1e-4)
    assert np.array_equal(rm.info.iter.cpu().numpy()[:nref], ref["iter"])
    np.testing.assert_allclose(rm.x.cpu().numpy()[:nref], ref["x"], rtol=0, atol=1e-8 * max(1.0, np.abs(ref["x"]).max()))


@pytest.mark.parametrize("max_iter,check_interval", [(30, 25), (50, 25), (0, 25), (64, 10)])
def test_max_iter_paths(max_iter, check_interval):
    ctl, H, g, A, l, u = _c3(20, seed=7)
    mm, rm = _solve(H, g, A, l, u, eps_abs=1e-12, max_iter=max_iter, check_interval=check_interval)
    assert mm.kernel == "mfmad"
    ref = O.solve_batch(H, g, A, l, u, form="factored", eps_abs=1e-12, max_iter=max_iter, check_interval=check_interval)
    assert rm.info.status == ref["status"] and np.array_equal(rm.info.iter.cpu().numpy(), ref["iter"])
    np.testing.assert_allclose(rm.x.cpu().numpy(), ref["x"], rtol=0, atol=1e-9 * max(1.0, np.abs(ref["x"]).max()))
    np.testing.assert_allclose(rm.info.pri_res.cpu().numpy(), ref["pri_res"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(rm.info.rho_estimate.cpu().numpy(), ref["rho_estimate"], rtol=1e-5)


def test_warm_start_updates_scaling_and_mixed_rho_tiles():
    """Closed-loop use in float64: per-instance state and rho index persist (tiles with several rho indices), update(g, l, u),
    update(Hx=), update(Ax=), Ruiz scaling, eps_rel -- step by step against the float64 resident kernel."""
    B = 200
    ctl, H, g, A, l, u = _c3(B, seed=11)
    g = g * np.linspace(0.05, 30.0, B)[:, None]
    for kw in (dict(), dict(scaling=10), dict(eps_rel=1e-4)):
        ms = {}
        for kern in ("mfma", "resident"):
            m = reluqpth.ReLU_QP()
            m.setup(H, g, A, l, u, device=DEV, precision=torch.float64, kernel=kern, eps_abs=1e-4, **kw)
            ms[kern] = m
        assert ms["mfma"].kernel == "mfmad" and ms["resident"].kernel == "resident64"
        for step in range(4):
            rs = {k: m.solve() for k, m in ms.items()}
            _same(rs["mfma"], rs["resident"], tol=1e-8)
            if step == 0:
                assert len(np.unique(rs["mfma"].info.rho_ind.cpu().numpy())) >= 2
                g2, l2, u2 = ctl.qp_vectors(np.random.RandomState(5).randn(B, 12))
                for m in ms.values():
                    m.update(g=g * 0.9, l=l2, u=u2)
            elif step == 1:
                for m in ms.values():
                    m.update(Hx=H * 1.2)
            elif step == 2:
                for m in ms.values():
                    m.update(Ax=A * 1.05)


def test_limits():
    H, g, A, l, u = _shared_dense(170, 0, 100, 4)
    with pytest.raises(_cabi.RqpError):
        reluqpth.ReLU_QP().setup(H, g, A, l, u, device=DEV, precision=torch.float64, kernel="mfma")
    ctl, H, g, A, l, u = _c3(8, form="sparse")                        # n = 320: beyond the float64 tile
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=DEV, precision=torch.float64, kernel="auto")
    assert m.kernel == "generic"
