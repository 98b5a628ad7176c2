"""GPU tests of k_admm_mfmal (csrc/rqp_mfmal.hip): shared-(H, A) batches BEYOND the register-resident MFMA tile (n <= 320,
m <= 640) -- the reference's own sparse linear-MPC form (loose_code/RandomLinMPC.py:54-66: n = 320, m = 560) and dense
shared problems with n > 80 or m > 320.  Operands are streamed from L2 as non-zero 16 x 16 blocks; the arithmetic is the exact
float32 of `v_mfma_f32_16x16x4_f32`, the recurrence and the checks are those of every other kernel.

Checked against (a) the oracle (CPU restatement of reluqpth.py:201-305, pinned by the reference's goldens in
tests/test_oracle_golden.py), (b) the streaming kernel on the same inputs, (c) float64 KKT residuals recomputed on the device.
Tolerances (float32 state, different summation order than the oracle's numpy): iteration counts identical on >= 90 % of
a batch and never more than three checks apart; x, z within 2e-4 * max|x| where the exits agree.
"""
import numpy as np
import pytest
import torch

from oracle import reluqp_oracle as O
from reluqp import mpc, utils, _cabi
import reluqp.reluqpth as reluqpth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _c3_sparse(B, seed=1, scale=1.0):
    Ad, Bd = mpc.random_plant(12, 4, seed=0)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="sparse")
    x0 = scale * np.random.RandomState(seed).randn(B, 12)
    g, l, u = ctl.qp_vectors(x0)
    return ctl, ctl.H, g, ctl.A, l, u


def _solve(H, g, A, l, u, kernel="mfma", **kw):
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=DEV, precision=torch.float32, kernel=kernel, **kw)
    return m, m.solve()


def _kkt(H, A, g, r):
    Hd, Ad_, gd = (torch.as_tensor(t, device=DEV, dtype=torch.float64) for t in (H, A, g))
    x, z, y = r.x.double(), r.z.double(), r.y.double()
    return (x @ Ad_.T - z).abs().amax(1), (x @ Hd.T + y @ Ad_ + gd).abs().amax(1)


def _shared_dense(n, n_eq, n_ineq, B, seed=5):
    H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=seed, compute_sol=False, feasible=True)
    qs = [utils.update_qp(H, A, n_eq, n_ineq, seed=50 + b, compute_sol=False, feasible=True) for b in range(B)]
    g, l, u = (np.stack([q[i] for q in qs]) for i in (1, 3, 4))
    return H, g, A, l, u


def test_sparse_c3_full_shape_vs_streaming_kernel_oracle_and_kkt():
    """BASELINE config 3 in the reference's own (sparse) form at the full batch: n = 320, m = 560, B = 4096."""
    B = 4096
    ctl, H, g, A, l, u = _c3_sparse(B)
    assert (H.shape[0], A.shape[0]) == (320, 560)
    ma, ra = _solve(H, g, A, l, u, kernel="auto", eps_abs=1e-3, warm_starting=False)
    assert ma.kernel == "mfmal"                                       # the default for this shape and batch
    mg, rg = _solve(H, g, A, l, u, kernel="generic", eps_abs=1e-3, warm_starting=False)
    assert mg.kernel == "generic"
    assert bool((ra.info.status_code == 0).all()) and not bool(torch.isnan(ra.x).any())
    ia, ig = ra.info.iter.cpu().numpy(), rg.info.iter.cpu().numpy()
    same = ia == ig
    assert same.mean() >= 0.99 and np.all(np.abs(ia - ig) <= 25), (same.mean(), np.abs(ia - ig).max())
    scale = float(rg.x.abs().max())
    np.testing.assert_allclose(ra.x.cpu().numpy()[same], rg.x.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(ra.z.cpu().numpy()[same], rg.z.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(ra.y.cpu().numpy()[same], rg.y.cpu().numpy()[same], rtol=0, atol=2e-3 * max(1.0, float(rg.y.abs().max())))
    # oracle on a subset (56 spread + the ragged end)
    idx = np.unique(np.concatenate([np.arange(0, B, B // 40)[:40], np.arange(B - 8, B)]))
    ref = O.solve_batch(H, g[idx], A, l[idx], u[idx], form="factored", eps_abs=1e-3)
    assert list(np.array(ra.info.status)[idx]) == ref["status"]
    so = ia[idx] == ref["iter"]
    assert so.mean() >= 0.9 and np.all(np.abs(ia[idx] - ref["iter"]) <= 75)
    np.testing.assert_allclose(ra.x.cpu().double().numpy()[idx][so], ref["x"][so], rtol=0, atol=2e-4 * max(1.0, np.abs(ref["x"]).max()))
    np.testing.assert_allclose(ra.info.obj_val.cpu().numpy()[idx][so], ref["obj_val"][so], rtol=1e-3, atol=1e-3)
    # independent: float64 KKT residuals of every instance against the thresholds the kernel tested
    pri, dua = _kkt(H, A, g, ra)
    assert float(pri.max()) < 1e-3 * np.sqrt(560) * 1.02 + 1e-5 and float(dua.max()) < 1e-3 * np.sqrt(320) * 1.02 + 1e-5
    np.testing.assert_allclose(ra.info.pri_res.cpu().numpy(), pri.cpu().numpy(), rtol=2e-2, atol=2e-5)
    np.testing.assert_allclose(ra.info.dua_res.cpu().numpy(), dua.cpu().numpy(), rtol=2e-2, atol=2e-5)
    # dynamics feasibility of the sparse form (equality rows l == u): met to the residual tolerance
    eq = l[0] == u[0]
    assert eq.sum() == 240 and float((ra.z.cpu().double().numpy()[:, eq] - l[:, eq]).__abs__().max()) < 1e-6


def test_regrouped_cold_solve_is_bit_identical_to_the_single_launch():
    """warm_starting=False: a cold solve runs as two launches -- every instance leaves behind its first check with its exact state
    (x, z, lam, float-float A x, carried rho estimate), the slots are re-sorted by the new rho indices, the second launch continues in
    homogeneous tiles (one pass of the dense K stream per distinct index of a tile).  A handle whose state came from the caller
    runs the same problems in ONE launch: every output must agree bit for bit, including the check trace and max_iter exits."""
    B = 1000
    ctl, H, g, A, l, u = _c3_sparse(B, seed=17)
    g = g + np.random.RandomState(3).randn(B, 320) * np.linspace(0.0, 2.0, B)[:, None]       # (spread the rho paths)
    for kw in (dict(eps_abs=1e-3), dict(eps_abs=1e-6, max_iter=150), dict(eps_abs=1e-4, check_interval=10, eps_rel=1e-4)):
        outs = []
        for ws in (False, True):
            m = reluqpth.ReLU_QP()
            m.collect_trace = True
            m.prefill_outputs = True
            m.setup(H, g, A, l, u, device=DEV, precision=torch.float32, kernel="mfma", warm_starting=ws, **kw)
            assert m.kernel == "mfmal"
            if ws:                     # a caller-provided state (here: the same zeros) is not known to be the common cold state:
                m.warm_start(x=np.zeros((B, 320)))        # this handle solves in ONE launch
            r = m.solve()
            outs.append((r.x.clone(), r.z.clone(), r.y.clone(), r.info.iter.clone(), r.info.status_code.clone(), r.info.rho_ind.clone(),
                         r.info.pri_res.clone(), r.info.dua_res.clone(), r.info.rho_estimate.clone(), r.info.obj_val.clone(),
                         torch.nan_to_num(m.last_trace.clone(), nan=-1.0)))
            if not ws:                                                # cold semantics: the state is cleared, the index reset
                xs, ri = m.get_state()
                assert float(xs.abs().max()) == 0.0 and bool((ri == m._rho_ind0()).all())
        for k, (ta, tb) in enumerate(zip(*outs)):
            assert torch.equal(ta, tb), (kw, k)
        assert len(torch.unique(outs[0][5])) >= 2 and not bool((outs[0][3] == -7).any())


def test_more_tiles_than_cus():
    """Three rounds of 16-instance tiles per CU and a ragged tail: every instance must come out as from the streaming kernel --
    same iteration counts, same solution; a warm re-solve; max_iter on and off the check grid at this grid size.  (A persistent
    grid with a refill queue was built and measured, tools/experiments/mfmal_refill_queue.patch: identical results, but slots at
    different stages sit at different rho indices and every index costs a pass of the dense K stream -- +10 % at 65 536
    instances, -2.5 % at 4096: not kept.)"""
    B = 3 * 4096 + 16 * 3 + 5
    ctl, H, g, A, l, u = _c3_sparse(B, seed=21)
    mm, rm = _solve(H, g, A, l, u, kernel="mfma", eps_abs=1e-3)
    mg, rg = _solve(H, g, A, l, u, kernel="generic", eps_abs=1e-3)
    assert mm.kernel == "mfmal" and mg.kernel == "generic"
    assert rm.info.status == rg.info.status and bool((rm.info.status_code == 0).all())
    im, ig = rm.info.iter.cpu().numpy(), rg.info.iter.cpu().numpy()
    assert im.min() > 0 and np.mean(im == ig) >= 0.995 and np.all(np.abs(im - ig) <= 25), (np.mean(im == ig), np.abs(im - ig).max())
    same = im == ig
    scale = max(1.0, float(rg.x.abs().max()))
    # (a marginal rho move at a check changes the path but not the count: a handful of 12 000 -- compared where the indices agree)
    same &= (rm.info.rho_ind.cpu().numpy() == rg.info.rho_ind.cpu().numpy())
    assert same.mean() >= 0.99
    np.testing.assert_allclose(rm.x.cpu().numpy()[same], rg.x.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(rm.z.cpu().numpy()[same], rg.z.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(rm.y.cpu().numpy()[same], rg.y.cpu().numpy()[same], rtol=0, atol=2e-3 * max(1.0, float(rg.y.abs().max())))
    pri, dua = _kkt(H, A, g, rm)
    assert float(pri.max()) < 1e-3 * np.sqrt(560) * 1.02 + 1e-5 and float(dua.max()) < 1e-3 * np.sqrt(320) * 1.02 + 1e-5
    r2 = mm.solve()                                                   # warm-started: state persisted per instance
    assert bool((r2.info.status_code == 0).all())
    assert float(r2.info.iter.double().mean()) <= float(rm.info.iter.double().mean())
    # max_iter on the check grid ends instances inside the persistent grid too; off the grid the tiles are static
    for mi in (50, 60):
        m3, r3 = _solve(H, g, A, l, u, kernel="mfma", eps_abs=1e-9, max_iter=mi, warm_starting=False)
        assert bool((r3.info.iter == mi).all()) and bool((r3.info.status_code == 1).all())
        if mi == 50:
            ref = O.solve_batch(H, g[-6:], A, l[-6:], u[-6:], form="factored", eps_abs=1e-9, max_iter=mi)
            np.testing.assert_allclose(r3.x.cpu().double().numpy()[-6:], ref["x"], rtol=0, atol=2e-4 * max(1.0, np.abs(ref["x"]).max()))


@pytest.mark.parametrize("B", [1, 17, 255, 300])
def test_sparse_c3_ragged_batches_and_dispatch_rule(B):
    """Ragged last tile (padding columns), a single instance; AUTO takes the streamed-operand kernel at any batch size (one
    instance: 1.0 ms against 2.8 ms on the streaming kernel)."""
    ctl, H, g, A, l, u = _c3_sparse(B, seed=3)
    mm, rm = _solve(H, g, A, l, u, kernel="mfma", eps_abs=1e-3)
    assert mm.kernel == "mfmal"
    ma, ra = _solve(H, g, A, l, u, kernel="auto", eps_abs=1e-3)
    assert ma.kernel == "mfmal"
    ref = O.solve_batch(H, g[:24], A, l[:24], u[:24], form="factored", eps_abs=1e-3)
    it = rm.info.iter.cpu().numpy()[:24]
    assert list(np.array(rm.info.status)[:24]) == ref["status"]
    assert np.mean(it == ref["iter"]) >= 0.9 and np.all(np.abs(it - ref["iter"]) <= 75)
    same = it == ref["iter"]
    np.testing.assert_allclose(rm.x.cpu().double().numpy()[:24][same], ref["x"][same], rtol=0, atol=2e-4 * max(1.0, np.abs(ref["x"]).max()))
    mg, rg = _solve(H, g, A, l, u, kernel="generic", eps_abs=1e-3)
    assert mg.kernel == "generic" and np.mean(rm.info.iter.cpu().numpy() == rg.info.iter.cpu().numpy()) >= 0.95


@pytest.mark.parametrize("n,n_eq,n_ineq", [(100, 25, 275), (150, 0, 200), (90, 13, 320), (320, 40, 600), (17, 3, 330)])
def test_dense_shared_problems_beyond_the_resident_tile(n, n_eq, n_ineq):
    """Dense random (H, A) shared by the batch: every block non-zero, equality rows (rho x 1e3), sizes that are not multiples
    of 16 (zero padding of tiles and of the dense K stream), the largest supported shape (320, 640)."""
    B = 48 if n < 320 else 20
    H, g, A, l, u = _shared_dense(n, n_eq, n_ineq, B)
    mm, rm = _solve(H, g, A, l, u, kernel="mfma", eps_abs=1e-3)
    assert mm.kernel == "mfmal"
    nref = B if n < 320 else 6
    ref = O.solve_batch(H, g[:nref], A, l[:nref], u[:nref], form="factored", eps_abs=1e-3)
    it = rm.info.iter.cpu().numpy()[:nref]
    assert list(np.array(rm.info.status)[:nref]) == ref["status"]
    assert np.mean(it == ref["iter"]) >= 0.8 and np.all(np.abs(it - ref["iter"]) <= 75), (it, ref["iter"])
    same = it == ref["iter"]
    scale = max(1.0, np.abs(ref["x"]).max())
    np.testing.assert_allclose(rm.x.cpu().double().numpy()[:nref][same], ref["x"][same], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(rm.z.cpu().double().numpy()[:nref][same], ref["z"][same], rtol=0, atol=2e-4 * scale)
    pri, dua = _kkt(H, A, g, rm)
    m_ = n_eq + n_ineq
    assert float(pri.max()) < 1e-3 * np.sqrt(m_) * 1.05 + 2e-5 and float(dua.max()) < 1e-3 * np.sqrt(n) * 1.05 + 5e-5


@pytest.mark.parametrize("max_iter,check_interval", [(30, 25), (50, 25), (0, 25), (64, 10)])
def test_max_iter_and_check_grid_paths_match_oracle(max_iter, check_interval):
    """max_iter off / on the check grid and 0 (final residual pass; compounded rho estimate on the grid), another interval."""
    ctl, H, g, A, l, u = _c3_sparse(20, seed=7)
    mm, rm = _solve(H, g, A, l, u, kernel="mfma", eps_abs=1e-9, max_iter=max_iter, check_interval=check_interval)
    assert mm.kernel == "mfmal"
    ref = O.solve_batch(H, g, A, l, u, form="factored", eps_abs=1e-9, max_iter=max_iter, check_interval=check_interval)
    assert rm.info.status == ref["status"]
    assert np.array_equal(rm.info.iter.cpu().numpy(), ref["iter"])
    scale = max(1.0, np.abs(ref["x"]).max())
    np.testing.assert_allclose(rm.x.cpu().double().numpy(), ref["x"], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(rm.info.pri_res.cpu().double().numpy(), ref["pri_res"], rtol=5e-2, atol=1e-4)
    # (the estimate is a ratio of residuals: compared where neither sits at the float32 noise floor)
    ok = (ref["pri_res"] > 1e-4) & (ref["dua_res"] > 1e-4)
    assert ok.sum() >= 10 or max_iter != 50
    np.testing.assert_allclose(rm.info.rho_estimate.cpu().double().numpy()[ok], ref["rho_estimate"][ok], rtol=0.1)


def test_warm_start_updates_and_mixed_rho_tiles():
    """Closed-loop use: state and rho index persist per instance (tiles whose columns sit at DIFFERENT rho indices run one
    K pass per index), update(g, l, u) and update(Hx=) rebuild what they must -- against the streaming kernel step by step."""
    B = 300
    ctl, H, g, A, l, u = _c3_sparse(B, seed=11)
    g = g * np.linspace(0.05, 30.0, B)[:, None] + np.random.RandomState(2).randn(B, 320) * np.linspace(0.0, 3.0, B)[:, None]
    ms = {}
    for kern in ("mfma", "generic"):
        m = reluqpth.ReLU_QP()
        m.setup(H, g, A, l, u, device=DEV, precision=torch.float32, kernel=kern, eps_abs=1e-3)
        ms[kern] = m
    assert ms["mfma"].kernel == "mfmal"
    for step in range(4):
        rs = {k: m.solve() for k, m in ms.items()}
        ia, ig = rs["mfma"].info.iter.cpu().numpy(), rs["generic"].info.iter.cpu().numpy()
        assert bool((rs["mfma"].info.status_code == 0).all())
        assert np.mean(ia == ig) >= 0.95 and np.all(np.abs(ia - ig) <= 50), (step, np.mean(ia == ig))
        same = ia == ig
        scale = max(1.0, float(rs["generic"].x.abs().max()))
        np.testing.assert_allclose(rs["mfma"].x.cpu().numpy()[same], rs["generic"].x.cpu().numpy()[same], rtol=0, atol=2e-4 * scale)
        ri = rs["mfma"].info.rho_ind.cpu().numpy()
        if step == 0:
            assert len(np.unique(ri)) >= 2                            # the batch really spreads over rho indices
            x1 = np.random.RandomState(5).randn(B, 12)
            g2, l2, u2 = ctl.qp_vectors(x1)
            for m in ms.values():
                m.update(g=g * 0.9, l=l2, u=u2)
        elif step == 1:
            for m in ms.values():
                m.update(Hx=H * 1.2)
        elif step == 2:
            A2 = A.copy()
            A2[A2 != 0] *= 1.05
            for m in ms.values():
                m.update(Ax=A2)
    assert np.median(ia) <= 100


def test_ruiz_scaling_and_eps_rel_and_trace():
    """Settings the kernel shares with the others: Ruiz scaling (caller-unit termination), eps_rel, the check trace."""
    ctl, H, g, A, l, u = _c3_sparse(64, seed=13)
    for kw in (dict(scaling=10), dict(eps_rel=1e-3), dict()):
        mm, rm = _solve(H, g, A, l, u, kernel="mfma", eps_abs=1e-3, **kw)
        mg, rg = _solve(H, g, A, l, u, kernel="generic", eps_abs=1e-3, **kw)
        assert mm.kernel == "mfmal" and mg.kernel == "generic"
        assert bool((rm.info.status_code == 0).all())
        ia, ig = rm.info.iter.cpu().numpy(), rg.info.iter.cpu().numpy()
        assert np.mean(ia == ig) >= 0.9 and np.all(np.abs(ia - ig) <= 50), (kw, ia, ig)
        same = ia == ig
        scale = max(1.0, float(rg.x.abs().max()))
        np.testing.assert_allclose(rm.x.cpu().numpy()[same], rg.x.cpu().numpy()[same], rtol=0, atol=3e-4 * scale)
    mm = reluqpth.ReLU_QP()
    mm.collect_trace = True
    mm.setup(H, g, A, l, u, device=DEV, precision=torch.float32, kernel="mfma", eps_abs=1e-3)
    mm.solve()
    for b in range(4):
        qp = O.OracleQP(form="factored", quirks=False)
        qp.setup(H, g[b], A, l[b], u[b], eps_abs=1e-3)
        qp.solve()
        rt = np.asarray(qp.trace, dtype=np.float64)
        tr = mm.last_trace[b].cpu().double().numpy()
        tr = tr[~np.isnan(tr[:, 3])]
        k = min(len(tr), len(rt))
        assert k >= 1 and abs(len(tr) - len(rt)) <= 1
        np.testing.assert_allclose(tr[:k, 3], rt[:k, 3], atol=0)      # rho index before each move
        np.testing.assert_allclose(tr[:k - 1, 0], rt[:k - 1, 0], rtol=5e-2, atol=1e-4)


def test_limits_are_refused_loudly():
    """Beyond (320, 640), float64, per-instance matrices: nothing falls back silently."""
    H, g, A, l, u = _shared_dense(330, 0, 100, 4)
    with pytest.raises(_cabi.RqpError):
        reluqpth.ReLU_QP().setup(H, g, A, l, u, device=DEV, precision=torch.float32, kernel="mfma")
    ctl, H, g, A, l, u = _c3_sparse(8)
    with pytest.raises(_cabi.RqpError):
        reluqpth.ReLU_QP().setup(H, g, A, l, u, device=DEV, precision=torch.float64, kernel="mfma")
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=DEV, precision=torch.float64, kernel="auto")
    assert m.kernel == "generic"
