"""GPU tests of k_admm_mfma16 (rqp_dims.tile_dtype = RQP_TILE_BF16, ``iterate_dtype=torch.bfloat16``): shared-(H, A)
batches on the 16-bit matrix pipe -- every matrix tile and vector operand as two bf16 planes (16 significant bits), three
v_mfma_f32_16x16x32_bf16 per product, float32 accumulation, float32 state / residuals / checks (csrc/rqp_mfma16.hip;
BASELINE config 5 "16-bit tile, float32 residual").

The reference has no 16-bit mode (it only runs float64), so nothing of it can pin this option: parity is
**unpinned** for it.  The checks are (a) against the float32 MFMA kernel -- the same recurrence with exact float32
products, itself checked against the oracle and the reference's goldens -- (b) against the oracle on a subset, and (c)
independent: the KKT residuals of every instance recomputed in float64 on the device against the thresholds the
kernel tested.  Stated tolerance (operand error 2^-16 relative): for eps_abs >= 1e-5, identical iteration counts on
>= 99 % of a batch (a marginal check may fall one check apart), x within 1e-4 * max|x| of the float32 kernel where the
exits agree.
"""
import numpy as np
import pytest
import torch

from oracle import reluqp_oracle as O
from reluqp import mpc, utils, _cabi

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _c3(B, seed=5, scale=1.0):
    Ad, Bd = mpc.random_plant(12, 4, seed=0)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="condensed")
    x0 = scale * np.random.RandomState(seed).randn(B, 12)
    g, l, u = ctl.qp_vectors(x0)
    return ctl.H, g, ctl.A, l, u


def _solve(H, g, A, l, u, tile, **kw):
    import reluqp.reluqpth as reluqpth
    m = reluqpth.ReLU_QP()
    m.prefill_outputs = True
    m.setup(H, g, A, l, u, device=DEV, precision=torch.float32, kernel="mfma", iterate_dtype=tile, warm_starting=False, **kw)
    assert m.kernel == ("mfma16" if tile == torch.bfloat16 else "mfma")
    return m, m.solve()


def _kkt(H, A, g, r):
    Hd, Ad_, gd = (torch.as_tensor(t, device=DEV, dtype=torch.float64) for t in (H, A, g))
    x, z, y = r.x.double(), r.z.double(), r.y.double()
    return (x @ Ad_.T - z).abs().amax(1), (x @ Hd.T + y @ Ad_ + gd).abs().amax(1)


@pytest.mark.parametrize("B,eps", [(64, 1e-3), (4096, 1e-3), (4096, 1e-5), (20000, 1e-3)])
def test_c3_bf16_planes_vs_float32_mfma(B, eps):
    """Config-3 batches: one tile per CU with the straggler hand-off (64, 4096) and the persistent grid with the refill
    queue (20000 > 16 x #CUs)."""
    H, g, A, l, u = _c3(B)
    _, r32 = _solve(H, g, A, l, u, None, eps_abs=eps)
    _, r16 = _solve(H, g, A, l, u, torch.bfloat16, eps_abs=eps)
    assert bool((r16.info.status_code == 0).all()) and bool((r32.info.status_code == 0).all())
    assert not bool(torch.isnan(r16.x).any()) and not bool((r16.info.iter == -7).any())
    i32, i16 = r32.info.iter.cpu().numpy(), r16.info.iter.cpu().numpy()
    same = i32 == i16
    assert same.mean() >= 0.99 and np.all(np.abs(i32 - i16) <= 25), (same.mean(), np.abs(i32 - i16).max())
    scale = float(r32.x.abs().max())
    np.testing.assert_allclose(r16.x.cpu().numpy()[same], r32.x.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    pri, dua = _kkt(H, A, g, r16)                                     # independent: float64 KKT residuals of every instance
    n, m_ = H.shape[0], A.shape[0]
    assert float(pri.max()) < eps * np.sqrt(m_) * 1.05 + 2e-5 and float(dua.max()) < eps * np.sqrt(n) * 1.05 + 2e-5
    np.testing.assert_allclose(r16.info.pri_res.cpu().numpy(), pri.cpu().numpy(), rtol=5e-2, atol=5e-5)


def test_c3_bf16_planes_vs_oracle():
    B = 4096
    H, g, A, l, u = _c3(B)
    _, r = _solve(H, g, A, l, u, torch.bfloat16, eps_abs=1e-3)
    idx = np.unique(np.concatenate([np.arange(0, B, B // 56)[:56], np.arange(B - 8, B)]))
    ref = O.solve_batch(H, g[idx], A, l[idx], u[idx], form="factored", eps_abs=1e-3)
    it = r.info.iter.cpu().numpy()[idx]
    assert list(np.array(r.info.status)[idx]) == ref["status"]
    same = it == ref["iter"]
    assert same.mean() >= 0.85 and np.all(np.abs(it - ref["iter"]) <= 25)
    xg = r.x.cpu().double().numpy()[idx]
    np.testing.assert_allclose(xg[same], ref["x"][same], rtol=0, atol=1e-4 * max(1.0, np.abs(ref["x"]).max()))
    np.testing.assert_allclose(r.info.obj_val.cpu().numpy()[idx][same], ref["obj_val"][same], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("n,n_eq,n_ineq", [(80, 20, 300), (50, 10, 160), (37, 0, 101), (16, 4, 16)])
def test_bf16_planes_on_dense_shared_problems(n, n_eq, n_ineq):
    """Dense random (H, A) shared by the batch -- equality rows (rho x 1e3), sizes that are not multiples of the 16 x 32
    tiles (zero padding, half k-steps), m not a multiple of 4 -- against the float32 MFMA kernel."""
    B = 200
    H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=5, compute_sol=False, feasible=True)
    qs = [utils.update_qp(H, A, n_eq, n_ineq, seed=50 + b, compute_sol=False, feasible=True) for b in range(B)]
    g, l, u = (np.stack([q[i] for q in qs]) for i in (1, 3, 4))
    _, r32 = _solve(H, g, A, l, u, None)
    _, r16 = _solve(H, g, A, l, u, torch.bfloat16)
    assert bool((r16.info.status_code == 0).all())
    i32, i16 = r32.info.iter.cpu().numpy(), r16.info.iter.cpu().numpy()
    same = i32 == i16
    # equality rows put rho x 1e3 into d = H x + g + A' nu: the cancellation amplifies the 2^-16 operand error, more checks
    # fall on their thresholds than on the MPC batches -- stated: >= 90 % identical, none more than four checks apart
    assert same.mean() >= 0.9 and np.all(np.abs(i32 - i16) <= 100), (same.mean(), np.abs(i32 - i16).max())
    scale = float(r32.x.abs().max())
    np.testing.assert_allclose(r16.x.cpu().numpy()[same], r32.x.cpu().numpy()[same], rtol=0, atol=5e-4 * scale)
    # The checks themselves run on the 16-bit planes: a residual is evaluated to ~2^-16 (|H| |x| + |A|' |lam|), which with
    # rho x 1e3 equality multipliers reaches 10-15 % of the eps_abs = 1e-3 thresholds (the MPC batches: < 1 %) -- stated:
    # float64 KKT residuals within 1.25 x the thresholds here; use the float32 MFMA kernel where the certificate must be exact
    pri, dua = _kkt(H, A, g, r16)
    m_ = A.shape[0]
    assert float(pri.max()) < 1e-3 * np.sqrt(m_) * 1.25 and float(dua.max()) < 1e-3 * np.sqrt(n) * 1.25


def test_bf16_planes_warm_start_update_and_rules():
    """State persists across solves (warm starts), update(g, l, u) and update(Hx=) rebuild what they must; the tile request is
    refused where no kernel implements it."""
    import reluqp.reluqpth as reluqpth
    B = 300
    H, g, A, l, u = _c3(B, seed=9)
    ms = {}
    for tile in (None, torch.bfloat16):
        m = reluqpth.ReLU_QP()
        m.setup(H, g, A, l, u, device=DEV, precision=torch.float32, kernel="mfma", iterate_dtype=tile)
        ms[tile] = m
    for step in range(3):
        rs = {t: m.solve() for t, m in ms.items()}
        i32, i16 = rs[None].info.iter.cpu().numpy(), rs[torch.bfloat16].info.iter.cpu().numpy()
        assert bool((rs[torch.bfloat16].info.status_code == 0).all())
        assert np.mean(i32 == i16) >= 0.97, (step, np.mean(i32 == i16))
        if step == 0:
            g2 = g * 1.2
            for m in ms.values():
                m.update(g=g2)
        elif step == 1:
            for m in ms.values():
                m.update(Hx=H * 1.3)
    assert np.median(i16) <= 50                                       # warm starts
    Hb, gb, Ab, lb, ub, _ = utils.rand_qp_batch(8, 20, 5, 15, seed0=1, feasible=True, dtype=np.float32)
    with pytest.raises(_cabi.RqpError) as ei:                         # per-instance matrices: no MFMA kernel, nothing falls back
        reluqpth.ReLU_QP().setup(Hb, gb, Ab, lb, ub, device=DEV, precision=torch.float32, iterate_dtype=torch.bfloat16)
    assert ei.value.code == _cabi.RQP_ERR_UNSUPPORTED
    with pytest.raises(_cabi.RqpError):
        reluqpth.ReLU_QP().setup(H, g, A, l, u, device=DEV, precision=torch.float32, iterate_dtype=torch.bfloat16, kernel="resident")
    with pytest.raises(ValueError):
        reluqpth.ReLU_QP().setup(H, g, A, l, u, device=DEV, precision=torch.float64, iterate_dtype=torch.bfloat16)
