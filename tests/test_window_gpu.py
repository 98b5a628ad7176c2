"""GPU tests of the rho-ladder WINDOW (RQP_WINDOW K(rho) entries per matrix instead of the whole ladder the reference
builds, reluqpth.py:52-78) and its exit-and-continue protocol (csrc/rqp_abi.hip: rqp_solve; kernels k_admm_res2 and
k_admm_generic).

The bar is bit-identity with a handle that holds the whole ladder (``full_ladder=True`` = RQP_FLAG_FULL_LADDER) -- whose
parity with the reference the golden tests pin: an instance that leaves its window stops with its exact state
(x, z, lam, A x, carried rho estimate, iteration count), gets a new window factored around its index and continues.
Cases are chosen so that instances DO leave the initial window [rho_ind0 - 1, rho_ind0 + 3] = [6, 10]:
  * the reference's own (signed-slack) generator at m = 2n: infeasible problems ratchet rho up the ladder to its last
    entries -- several window moves per instance, every instance ends max_iters_reached;
  * eps_abs = 1e-6 float64 solves (golden G4 `e6_` trajectory: rho index 7 -> 11);
  * warm starts with rho far from the window, k plain iterations (rqp_iterate) there, K_j outside the window.
"""
import numpy as np
import pytest
import torch

from reluqp import utils, _cabi

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _solver(H, g, A, l, u, prec, full, **kw):
    import reluqp.reluqpth as reluqpth
    m = reluqpth.ReLU_QP()
    m.collect_trace = True
    m.prefill_outputs = True
    m.setup(H, g, A, l, u, device=DEV, precision=prec, full_ladder=full, **kw)
    return m


def _snap(res, model):
    i = res.info
    d = dict(x=res.x.clone(), z=res.z.clone(), y=res.y.clone(), iter=i.iter.clone(), status=i.status_code.clone(),
             rho_ind=i.rho_ind.clone(), pri=i.pri_res.clone(), dua=i.dua_res.clone(), rho=i.rho_estimate.clone(),
             obj=i.obj_val.clone(), trace=model.last_trace.clone())
    st, ri = model.get_state()
    d["state"], d["state_ri"] = st.clone(), ri.clone()
    return d


def _same(a, b, what):
    for k in a:
        x, y = a[k], b[k]
        if x.is_floating_point():
            ok = torch.equal(torch.nan_to_num(x, nan=12345.0), torch.nan_to_num(y, nan=12345.0))
        else:
            ok = torch.equal(x, y)
        assert ok, "%s: %s differs between the windowed and the full-ladder handle" % (what, k)


CASES = [  # (precision, kernel, n, n_eq, n_ineq, feasible, settings)
    (torch.float32, "resident", 10, 5, 15, False, dict(max_iter=600)),
    (torch.float64, "generic", 10, 5, 15, False, dict(max_iter=600)),
    (torch.float32, "generic", 12, 4, 20, False, dict(max_iter=500, check_interval=10)),
    (torch.float32, "resident", 40, 10, 70, False, dict(max_iter=400)),
    (torch.float64, "generic", 30, 8, 50, True, dict(eps_abs=1e-7, max_iter=2000)),
    (torch.float32, "resident", 100, 25, 275, True, dict(eps_abs=1e-4)),
    (torch.float32, "resident", 72, 18, 150, True, dict(eps_abs=1e-4)),              # the (80, 320) tile: direct K image, borrowed A
    (torch.float32, "resident", 33, 8, 60, False, dict(max_iter=400)),               # padded rows (ldn = 36 != n): direct K image, copied A
    (torch.float64, "resident", 10, 5, 15, False, dict(max_iter=600)),               # k_admm_res64
    (torch.float64, "resident", 40, 10, 70, False, dict(max_iter=400, check_interval=10)),
    (torch.float64, "resident", 100, 25, 275, True, dict(eps_abs=1e-6)),
    (torch.float32, "wave", 10, 5, 15, False, dict(max_iter=600)),                   # k_admm_wave: one wavefront per QP
    (torch.float64, "wave", 12, 4, 20, False, dict(max_iter=500, check_interval=10)),
    (torch.float32, "wave", 30, 8, 100, False, dict(max_iter=400)),                  # two rows per lane
    (torch.float32, "wave", 60, 10, 90, True, dict(eps_abs=1e-5)),                   # two wavefronts per QP
    (torch.float32, "wave", 32, 8, 56, True, dict(eps_abs=1e-5)),                    # BASELINE config 4 shape
]


@pytest.mark.parametrize("prec,kernel,n,n_eq,n_ineq,feasible,st", CASES)
def test_window_bit_identical_to_full_ladder(prec, kernel, n, n_eq, n_ineq, feasible, st):
    B = 48
    dt = np.float32 if prec == torch.float32 else np.float64
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=11, feasible=feasible, dtype=dt)
    mw = _solver(H, g, A, l, u, prec, False, kernel=kernel, **st)
    mf = _solver(H, g, A, l, u, prec, True, kernel=kernel, **st)
    assert mw.kernel == mf.kernel
    nrho = len(mw._rhos)
    assert mw.get_window()[0] == 5 and mf.get_window() == (nrho, None)
    wb0 = mw.get_window()[1].cpu().numpy()
    assert np.all(wb0 == 6)                                           # rho_ind0 - 1 (rho_ind0 = 7 for the default ladder)
    sw, sf = _snap(mw.solve(), mw), _snap(mf.solve(), mf)
    assert not bool((sw["iter"] == -7).any()) and not bool(torch.isnan(sw["x"]).any())
    _same(sw, sf, "cold solve")
    ri = sw["rho_ind"].cpu().numpy()
    wb1 = mw.get_window()[1].cpu().numpy()
    if not feasible:                                                  # the ratchet: rho ends far above the initial window
        assert np.mean(ri > 10) > 0.25, "this case is meant to leave the window"
        assert np.any(wb1 != 6)
    # every instance's final index lies inside its final window, or it left at its last check (then the next solve re-centres)
    # warm re-solve: starts at the persisted indices (some outside the windows) and states
    sw2, sf2 = _snap(mw.solve(), mw), _snap(mf.solve(), mf)
    _same(sw2, sf2, "warm re-solve")
    # new vectors (reference update(), reluqpth.py:159-183), then a cold solve after clear_primal_dual
    g2 = g * 1.25
    for m_ in (mw, mf):
        m_.update(g=g2)
    _same(_snap(mw.solve(), mw), _snap(mf.solve(), mf), "after update(g)")
    for m_ in (mw, mf):
        m_.clear_primal_dual()
    _same(_snap(mw.solve(), mw), _snap(mf.solve(), mf), "after clear_primal_dual")


def test_window_follows_golden_e6_trajectory(golden):
    """The reference's own eps_abs = 1e-6 run of G4 seed 0 walks the rho index 7 -> 11, out of the window [6, 10]; as
    instance 0 of a windowed float64 batch it must reproduce that run's iteration count, trajectory and solution."""
    gold = golden("g4_c2_feasible.npz")
    B = 40
    H, g, A, l, u, _ = utils.rand_qp_batch(B, 100, 25, 275, seed0=0, feasible=True)
    p = "s0_e6_"
    gt = gold[p + "trace"]
    assert gt[:, 3].max() > 10, "fixture no longer leaves the window"
    snaps = {}
    for full in (False, True):
        m = _solver(H, g, A, l, u, torch.float64, full, kernel="generic", eps_abs=1e-6)
        assert m.get_window()[0] == (len(m._rhos) if full else 5)
        res = m.solve()
        snaps[full] = _snap(res, m)
        # 2200 iterations at eps_abs = 1e-6: the terminating check is marginal (K by Gauss-Jordan here, torch.inverse there);
        # one check of slack on the count, exact rho-index trajectory on the first two thirds
        assert abs(int(res.info.iter[0]) - int(gold[p + "iter"])) <= 25
        tr = m.last_trace[0].cpu().numpy()
        tr = tr[~np.isnan(tr[:, 3])]
        k = (2 * min(len(tr), len(gt))) // 3                         # (late moves of this long run sit on their thresholds)
        assert np.array_equal(tr[:k, 3], gt[:k, 3]) and tr[:, 3].max() > 10
        np.testing.assert_allclose(tr[:k, :2], gt[:k, :2], rtol=1e-2, atol=1e-7)
        np.testing.assert_allclose(res.x[0].cpu().numpy(), gold[p + "x"], rtol=0, atol=1e-6 * max(1.0, np.abs(gold[p + "x"]).max()))
        if not full:
            assert int(m.get_window()[1][0]) > 6                      # instance 0's window moved up
    _same(snaps[False], snaps[True], "eps_abs = 1e-6 batch")


@pytest.mark.parametrize("prec,xtol,k64", [(torch.float64, 1e-8, "generic"), (torch.float32, 2e-5, None), (torch.float64, 1e-8, "resident")])
def test_window_g1_chain_below_the_window(golden, prec, xtol, k64):
    """The reference's built-in QP (reluqpth.py:342-346) as a windowed batch of 40 copies: cold solve 7 -> 6, warm re-solve
    ends at index 5 -- below the window -- so the next solve starts outside it (golden G1 / G5 values of the reference)."""
    gold = golden("g1_builtin.npz")
    B = 40
    rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)
    H, g, A, l, u = (rep(gold[k]) for k in ("H", "g", "A", "l", "u"))
    m = _solver(H, g, A, l, u, prec, False, kernel="resident" if prec == torch.float32 else k64)
    assert m.get_window()[0] == 5
    r1 = m.solve()
    assert bool((r1.info.iter == int(gold["iter"])).all()) and bool((r1.info.status_code == 0).all())
    np.testing.assert_allclose(r1.x.cpu().double().numpy(), rep(gold["x"]), rtol=0, atol=xtol * 3)
    r2 = m.solve()                                                    # warm: index 6 -> 5 at its terminating check
    assert bool((r2.info.iter == int(gold["warm_iter"])).all())
    if prec == torch.float64:       # (float32: the last estimate is a ratio of rounding noise, test_hip_parity.test_g1_solve_*)
        assert bool((r2.info.rho_ind == int(gold["warm_rho_ind_final"])).all()) and int(gold["warm_rho_ind_final"]) == 5
    np.testing.assert_allclose(r2.x.cpu().double().numpy(), rep(gold["warm_x"]), rtol=0, atol=xtol * 3)
    r3 = m.solve()                                                    # starts at index 5: outside every window
    assert bool((r3.info.status_code == 0).all())
    if prec == torch.float64:
        assert bool((m.get_window()[1] < 6).all())
    mf = _solver(H, g, A, l, u, prec, True, kernel="resident" if prec == torch.float32 else k64)
    mf.solve(); mf.solve()
    r3f = mf.solve()
    assert torch.equal(r3.x, r3f.x) and torch.equal(r3.info.iter, r3f.info.iter) and torch.equal(r3.info.rho_ind, r3f.info.rho_ind)


@pytest.mark.parametrize("prec,kernel", [(torch.float32, "resident"), (torch.float64, "generic"), (torch.float64, "resident"),
                                         (torch.float32, "wave")])
def test_window_warm_start_iterate_and_K_outside(prec, kernel):
    B, n, n_eq, n_ineq = 40, 20, 5, 35
    dt = np.float32 if prec == torch.float32 else np.float64
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=3, feasible=True, dtype=dt)
    mw = _solver(H, g, A, l, u, prec, False, kernel=kernel)
    mf = _solver(H, g, A, l, u, prec, True, kernel=kernel)
    # K_j far outside the window = the full-ladder entry (factored on demand)
    for j in (0, 3, 7, 13, 17):
        for b in (0, B - 1):
            assert torch.equal(mw.layers.K(j, instance=b), mf.layers.K(j, instance=b))
    # warm start at rho = 1e3 (index 13) and rho = 1e-5 (index 1): the solve starts outside every window
    for rho in (1e3, 1.3e-5):
        for m_ in (mw, mf):
            m_.warm_start(rho=rho)
        _same(_snap(mw.solve(), mw), _snap(mf.solve(), mf), "warm_start(rho=%g)" % rho)
    # k plain iterations (ReLU_Layer.forward, reluqpth.py:80-89) at an index outside the window
    for m_ in (mw, mf):
        m_.clear_primal_dual()
        m_.warm_start(rho=2e2)
    sw, sf = mw.iterate(30), mf.iterate(30)
    assert torch.equal(sw, sf)
    # matrix update (rqp_update_mats) after windows moved: new H on the moved windows
    H2 = H * 1.1
    for m_ in (mw, mf):
        m_.update(Hx=H2)
    _same(_snap(mw.solve(), mw), _snap(mf.solve(), mf), "after update(Hx)")
    # new A alone, then both (a windowed float32 resident handle keeps no row-major copy of A -- rqp_handle.borrow_A: with a new
    # H only, G = A'cA and the register image of A stand; with a new A both are rebuilt from the caller's matrix)
    A2 = (A * 1.02).astype(dt)
    for m_ in (mw, mf):
        m_.update(Ax=A2)
    _same(_snap(mw.solve(), mw), _snap(mf.solve(), mf), "after update(Ax)")
    for m_ in (mw, mf):
        m_.update(Hx=H, Ax=A)
        m_.clear_primal_dual()
    _same(_snap(mw.solve(), mw), _snap(mf.solve(), mf), "after update(Hx, Ax)")
    for m_ in (mw, mf):
        m_.update(Hx=H2)
    _same(_snap(mw.solve(), mw), _snap(mf.solve(), mf), "after a second update(Hx)")


def test_window_rules():
    """Which handles are windowed, and the calls a window refuses."""
    B, n, n_eq, n_ineq = 40, 10, 3, 12
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=5, feasible=True, dtype=np.float32)
    mw = _solver(H, g, A, l, u, torch.float32, False, kernel="resident")
    assert mw.get_window()[0] == 5
    with pytest.raises(_cabi.RqpError):                              # the certificate pass reads K at the final index
        mw.update_settings(check_infeasibility=True)
    mc = _solver(H, g, A, l, u, torch.float32, False, kernel="resident", check_infeasibility=True)
    assert mc.get_window()[1] is None                                # asked for at setup: the whole ladder
    ms = _solver(H[:8], g[:8], A[:8], l[:8], u[:8], torch.float32, False, kernel="resident")
    assert ms.get_window()[1] is None                                # small batches: the whole ladder
    mh = _solver(H[0], g, A[0], l, u, torch.float32, False)          # shared (H, A): one ladder for the batch
    assert mh.get_window()[1] is None
    mv = _solver(H, g, A, l, u, torch.float32, False)                # one-wavefront kernel: windowed too
    assert mv.kernel == "wave" and mv.get_window()[0] == 5
    # a windowed solve cannot be captured into a HIP graph (it synchronises); full_ladder can
    graph = torch.cuda.CUDAGraph()
    mw.synchronous = False
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        mw.solve()
    torch.cuda.current_stream(DEV).wait_stream(side)
    dummy = torch.zeros(8, device=DEV)
    err = None
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        dummy.add_(1.0)                                              # (a non-empty capture)
        try:
            mw.solve()
        except _cabi.RqpError as e:
            err = e
    torch.cuda.synchronize()
    assert err is not None and err.code == _cabi.RQP_ERR_UNSUPPORTED and "FULL_LADDER" in str(err)
