"""Linear-MPC path on the GPU: shared (H, A) batches through the C-ABI against the oracle."""
import numpy as np
import pytest
import torch

from oracle import reluqp_oracle as O
from reluqp import mpc

pytestmark = pytest.mark.gpu


def _setup(form, nx=6, nu=2, N=10, seed=7, B=24):
    Ad, Bd = mpc.random_plant(nx, nu, seed=seed)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(nx), 0.1 * np.eye(nu), N, u_max=0.4, x_max=8.0, form=form,
                        device=torch.device("cuda:0"), precision=torch.float32, eps_abs=1e-3)
    x0 = 1.5 * np.random.RandomState(seed).randn(B, nx)
    return ctl, x0


@pytest.mark.parametrize("form", ["condensed", "sparse"])
def test_mpc_batch_matches_oracle(form):
    import reluqp.reluqpth as reluqpth
    ctl, x0 = _setup(form)
    g, l, u = ctl.qp_vectors(x0)
    model = reluqpth.ReLU_QP()
    model.setup(ctl.H, g, ctl.A, l, u, device=torch.device("cuda:0"), precision=torch.float32, eps_abs=1e-3)
    assert model.QP.shared_mats and model.kernel in ("wave", "resident2")     # condensed: n=20, m=80 -> one-wavefront kernel
    res = model.solve()
    ref = O.solve_batch(ctl.H, g, ctl.A, l, u, form="factored", eps_abs=1e-3)
    assert res.info.status == ref["status"]            # incl. any instance the reference algorithm cannot solve
    assert np.mean([s == "solved" for s in res.info.status]) >= 0.9
    it = res.info.iter.cpu().numpy()
    assert np.mean(it == ref["iter"]) >= 0.85
    same = it == ref["iter"]
    x = res.x.cpu().double().numpy()
    np.testing.assert_allclose(x[same], ref["x"][same], rtol=0, atol=1e-4 * max(1.0, np.abs(ref["x"]).max()))
    u0 = ctl.first_input(x, x0)
    assert np.all(np.abs(u0) <= 0.4 + 2e-2)                           # eps_abs-level constraint satisfaction


def test_mpc_closed_loop_update_and_warm_start():
    ctl, x0 = _setup("condensed", B=16)
    xs, us, its = ctl.simulate(x0, steps=40)
    nrm = np.linalg.norm(xs, axis=2)
    assert np.all(nrm[-1] < 0.5 * nrm[0])                             # regulated
    assert np.all(np.abs(us) <= 0.4 + 2e-2)
    assert its[10:].mean() < its[0].mean()                           # warm starts pay
    # the same loop with the float64 kernels follows the same trajectory (tolerance: eps_abs-level inputs)
    Ad, Bd = ctl.Ad, ctl.Bd
    ctl64 = mpc.LinearMPC(Ad, Bd, np.eye(6), 0.1 * np.eye(2), 10, u_max=0.4, x_max=8.0, form="condensed",
                          device=torch.device("cuda:0"), precision=torch.float64, eps_abs=1e-3)
    xs64, us64, _ = ctl64.simulate(x0, steps=40)
    assert np.abs(xs64 - xs).max() < 5e-2 * np.abs(xs).max()


@pytest.mark.parametrize("B", [16, 40])
def test_mfma_kernel_matches_resident_and_oracle(B):
    """Shared-(H,A) batches on the MFMA kernel (forced with RQP_MFMA=1): same exits as the per-instance resident
    kernel and the oracle, x within float32 tolerance; ragged last tile (B=40 = 2.5 tiles)."""
    import reluqp.reluqpth as reluqpth
    ctl, x0 = _setup("condensed", nx=6, nu=2, N=10, seed=11, B=B)
    g, l, u = ctl.qp_vectors(x0)
    dev = torch.device("cuda:0")
    mm = reluqpth.ReLU_QP()
    mm.collect_trace = True
    mm.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float32, eps_abs=1e-3, kernel="mfma")
    mr = reluqpth.ReLU_QP()
    mr.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float32, eps_abs=1e-3)
    assert mm.kernel == "mfma" and mr.kernel in ("resident2", "wave")
    rm, rr = mm.solve(), mr.solve()
    ref = O.solve_batch(ctl.H, g, ctl.A, l, u, form="factored", eps_abs=1e-3)
    itm, itr = rm.info.iter.cpu().numpy(), rr.info.iter.cpu().numpy()
    assert rm.info.status == ref["status"]
    assert np.mean(itm == ref["iter"]) >= 0.85 and np.mean(itm == itr) >= 0.85
    same = itm == ref["iter"]
    scale = max(1.0, np.abs(ref["x"]).max())
    np.testing.assert_allclose(rm.x.cpu().double().numpy()[same], ref["x"][same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(rm.z.cpu().double().numpy()[same], ref["z"][same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(rm.y.cpu().double().numpy()[same], ref["lam"][same], rtol=0, atol=2e-3 * max(1.0, np.abs(ref["lam"]).max()))
    np.testing.assert_allclose(rm.info.obj_val.cpu().double().numpy()[same], ref["obj_val"][same], rtol=1e-3, atol=1e-3)
    assert np.array_equal(rm.info.rho_ind.cpu().numpy()[same], ref["rho_ind"][same])
    # warm re-solve after update() goes through the same kernel and persists its state
    g2, l2, u2 = ctl.qp_vectors(0.9 * x0)
    mm.update(g=g2, l=l2, u=u2)
    r2 = mm.solve()
    ref2 = O.solve_batch(ctl.H, g2, ctl.A, l2, u2, form="factored", eps_abs=1e-3)
    assert all(s == "solved" for s in r2.info.status)
    assert r2.info.iter.double().mean() <= ref2["iter"].mean()          # warm start: no slower than cold


def _solve_with_env(val, ctl, g, l, u, **settings):
    """setup()+solve() with a kernel request through the C ABI: "1" -> the MFMA kernel, "0" -> a per-instance kernel
    (the resident tile where the default dispatch would pick MFMA), None -> the default dispatch."""
    import reluqp.reluqpth as reluqpth
    n_, m_ = ctl.H.shape[-1], ctl.A.shape[-2]
    kernel = {"1": "mfma", None: "auto"}.get(val)
    if kernel is None:
        kernel = "resident" if (g.shape[0] >= 2048 and (n_ > 56 or m_ > 128)) else "auto"
    m = reluqpth.ReLU_QP()
    m.setup(ctl.H, g, ctl.A, l, u, device=torch.device("cuda:0"), precision=torch.float32, kernel=kernel, **settings)
    return m, m.solve()


def test_mfma_is_default_for_large_shared_batches_full_shape():
    """BASELINE config-3 shape (nx=12, nu=4, N=20 -> n=80, m=320, the largest tile of the MFMA kernel) at B=2064 (129
    tiles): the default dispatch picks the MFMA kernel and it agrees with the per-instance resident kernel."""
    ctl, x0 = _setup("condensed", nx=12, nu=4, N=20, seed=5, B=2064)
    g, l, u = ctl.qp_vectors(x0)
    mm, rm = _solve_with_env(None, ctl, g, l, u, eps_abs=1e-3)
    mr, rr = _solve_with_env("0", ctl, g, l, u, eps_abs=1e-3)
    assert mm.kernel == "mfma" and mr.kernel in ("resident2", "wave")
    assert rm.info.status == rr.info.status
    itm, itr = rm.info.iter.cpu().numpy(), rr.info.iter.cpu().numpy()
    assert np.mean(itm == itr) >= 0.9
    same = itm == itr
    scale = max(1.0, float(rr.x.abs().max()))
    np.testing.assert_allclose(rm.x.cpu().numpy()[same], rr.x.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(rm.z.cpu().numpy()[same], rr.z.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    assert np.array_equal(rm.info.rho_ind.cpu().numpy()[same], rr.info.rho_ind.cpu().numpy()[same])
    # solved instances meet the termination test on recomputed residuals
    pri, dua = mm.compute_residuals(0.1)[:2]                # on the persisted (warm-start) state
    ok = np.array([s == "solved" for s in rm.info.status])
    thr = 1e-3 * np.sqrt(ctl.A.shape[0])
    assert float(pri.cpu().numpy()[ok].max()) < 1.05 * thr


@pytest.mark.parametrize("max_iter", [30, 50, 0])
def test_mfma_max_iter_paths_match_oracle(max_iter):
    """max_iter off / on the check grid and 0: the max-iter exits of the MFMA kernel (final residual pass, compounded
    rho estimate when max_iter is a multiple of check_interval) match the oracle's."""
    ctl, x0 = _setup("condensed", nx=6, nu=2, N=10, seed=3, B=24)
    g, l, u = ctl.qp_vectors(x0)
    mm, rm = _solve_with_env("1", ctl, g, l, u, eps_abs=1e-9, max_iter=max_iter)
    assert mm.kernel == "mfma"
    ref = O.solve_batch(ctl.H, g, ctl.A, l, u, form="factored", eps_abs=1e-9, max_iter=max_iter)
    assert rm.info.status == ref["status"]
    assert np.array_equal(rm.info.iter.cpu().numpy(), ref["iter"])
    scale = max(1.0, np.abs(ref["x"]).max())
    np.testing.assert_allclose(rm.x.cpu().double().numpy(), ref["x"], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(rm.info.pri_res.cpu().double().numpy(), ref["pri_res"], rtol=5e-2, atol=1e-4)
    np.testing.assert_allclose(rm.info.rho_estimate.cpu().double().numpy(), ref["rho_estimate"], rtol=0.1)


def test_mfma_persistent_grid_refills_slots():
    """More 16-instance tiles than CUs: the MFMA kernel runs a persistent grid whose slots take the next unsolved instance
    when theirs exits.  Every instance must come out exactly as from the per-instance resident kernel (same iteration
    counts -- each instance keeps its own check schedule -- and the same solution), including the ragged tail."""
    B = 256 * 16 * 2 + 16 * 3 + 5
    ctl, x0 = _setup("condensed", nx=6, nu=2, N=10, seed=9, B=B)
    g, l, u = ctl.qp_vectors(x0)
    mm, rm = _solve_with_env("1", ctl, g, l, u, eps_abs=1e-3)      # (forced: at n=20, m=80 the default stays per-instance)
    mr, rr = _solve_with_env("0", ctl, g, l, u, eps_abs=1e-3)
    assert mm.kernel == "mfma" and mr.kernel in ("resident2", "wave")
    assert rm.info.status == rr.info.status
    itm, itr = rm.info.iter.cpu().numpy(), rr.info.iter.cpu().numpy()
    assert itm.min() > 0 and np.mean(itm == itr) >= 0.9
    same = itm == itr
    scale = max(1.0, float(rr.x.abs().max()))
    np.testing.assert_allclose(rm.x.cpu().numpy()[same], rr.x.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(rm.z.cpu().numpy()[same], rr.z.cpu().numpy()[same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(rm.y.cpu().numpy()[same], rr.y.cpu().numpy()[same], rtol=0,
                               atol=2e-3 * max(1.0, float(rr.y.abs().max())))
    # (the index move at the terminating check is a threshold test on a noise-level estimate: a handful of 8000 differ)
    assert np.mean(rm.info.rho_ind.cpu().numpy()[same] == rr.info.rho_ind.cpu().numpy()[same]) >= 0.995
    # a second (warm-started) solve through the same persistent grid: state persisted per instance
    r2 = mm.solve()
    assert all(s == "solved" for s in r2.info.status)
    assert float(r2.info.iter.double().mean()) <= float(rm.info.iter.double().mean())


def test_mfma_dense_shared_matrices_with_equality_rows():
    """Dense random shared (H, A) with equality rows (rho x 1e3 on them) and per-instance (g, l, u): no zero operand
    tiles to skip, every n/m padding case of the <5,5> tile; MFMA kernel vs the oracle."""
    from reluqp import utils
    B, n, n_eq, n_ineq = 40, 70, 10, 189          # m = 199: not a multiple of 4 (scalar load/store path)
    H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=5, compute_sol=False, feasible=True)
    qs = [utils.update_qp(H, A, n_eq, n_ineq, seed=50 + b, compute_sol=False, feasible=True) for b in range(B)]
    g = np.stack([q[1] for q in qs])
    l = np.stack([q[3] for q in qs])
    u = np.stack([q[4] for q in qs])

    class _Ctl:            # duck-typed holder for _solve_with_env
        pass
    ctl = _Ctl()
    ctl.H, ctl.A = H, A
    mm, rm = _solve_with_env("1", ctl, g, l, u, eps_abs=1e-3)
    assert mm.kernel == "mfma"
    ref = O.solve_batch(H, g, A, l, u, form="factored", eps_abs=1e-3)
    assert rm.info.status == ref["status"]
    itm = rm.info.iter.cpu().numpy()
    assert np.mean(itm == ref["iter"]) >= 0.8 and np.all(np.abs(itm - ref["iter"]) <= 75)
    same = itm == ref["iter"]
    scale = max(1.0, np.abs(ref["x"]).max())
    np.testing.assert_allclose(rm.x.cpu().double().numpy()[same], ref["x"][same], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(rm.z.cpu().double().numpy()[same], ref["z"][same], rtol=0, atol=2e-4 * scale)
    # equality rows are met to the residual tolerance
    z = rm.z.cpu().double().numpy()
    assert np.abs(z[:, :n_eq] - l[:, :n_eq]).max() < 1e-2


@pytest.mark.parametrize("prec", [torch.float32, torch.float64])
def test_update_affine_matches_update(prec):
    """rqp_update_affine (g = G x0, l/u = l_add/u_add + LU x0 on the device) leaves the solver in the same state as
    update(g, l, u) with host-built vectors: identical iteration counts, solutions within rounding."""
    import reluqp.reluqpth as reluqpth
    ctl, x0 = _setup("condensed", nx=6, nu=2, N=10, seed=4, B=48)
    dev = torch.device("cuda:0")
    g, l, u = ctl.qp_vectors(x0)
    x1 = 0.8 * x0 + 0.05
    g1, l1, u1 = ctl.qp_vectors(x1)
    ma, mb = reluqpth.ReLU_QP(), reluqpth.ReLU_QP()
    for mdl in (ma, mb):
        mdl.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=prec, eps_abs=1e-4)
        mdl.solve()
    ma.update(g=g1, l=l1, u=u1)
    mb.update_affine(x1, ctl.g_x0, ctl.lu_x0, ctl.l_add, ctl.u_add)
    ra, rb = ma.solve(), mb.solve()
    assert ra.info.status == rb.info.status
    ita, itb = ra.info.iter.cpu().numpy(), rb.info.iter.cpu().numpy()
    assert np.mean(ita == itb) >= 0.95
    same = ita == itb
    tol = 1e-9 if prec == torch.float64 else 2e-5
    scale = max(1.0, float(ra.x.abs().max()))
    np.testing.assert_allclose(rb.x.cpu().double().numpy()[same], ra.x.cpu().double().numpy()[same], rtol=0, atol=tol * scale)
    np.testing.assert_allclose(rb.z.cpu().double().numpy()[same], ra.z.cpu().double().numpy()[same], rtol=0, atol=tol * scale)
    with pytest.raises(ValueError):
        mb.update_affine(x1[:, :3], ctl.g_x0, ctl.lu_x0, ctl.l_add, ctl.u_add)


def test_closed_loop_graph_replay_matches_eager():
    """The control step captured in a HIP graph and replayed gives the same trajectory as the eager device loop."""
    dev = torch.device("cuda:0")
    ctl_a, x0 = _setup("condensed", nx=6, nu=2, N=10, seed=6, B=64)
    ctl_b, _ = _setup("condensed", nx=6, nu=2, N=10, seed=6, B=64)
    xa, _ = ctl_a.simulate_device(x0, 1, dev, torch.float32)
    xb, _ = ctl_b.simulate_device(x0, 1, dev, torch.float32)
    assert torch.equal(xa, xb)
    xa, ita = ctl_a.simulate_device(xa.cpu().numpy(), 12, dev, torch.float32)
    xb, itb = ctl_b.simulate_graph(xb.cpu().numpy(), 12, dev, torch.float32)
    assert abs(ita - itb) < 1e-9
    np.testing.assert_allclose(xb.cpu().numpy(), xa.cpu().numpy(), rtol=0, atol=1e-6)


def test_closed_loop_sparse_form_on_device_matches_host_loop_and_graph():
    """The reference's own (sparse) MPC form in closed loop: the device loop (affine x0 maps, rqp_update_affine) follows the
    host loop (qp_vectors + update(g, l, u) per step, the path of reluqpth.py:159-183) and the HIP-graph replay; the
    streamed-operand MFMA kernel serves it at the full config-3 shape."""
    dev = torch.device("cuda:0")
    B, steps = 300, 6
    ctls = []
    for _ in range(3):
        Ad, Bd = mpc.random_plant(12, 4, seed=0)
        ctls.append(mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="sparse", device=dev,
                                  precision=torch.float32, eps_abs=1e-4))
    x0 = 0.5 * np.random.RandomState(4).randn(B, 12)
    xa, _ = ctls[0].simulate_device(x0, 1, dev, torch.float32)
    xb, _ = ctls[1].simulate_device(x0, 1, dev, torch.float32)
    assert ctls[0].solver.kernel == "mfmal" and torch.equal(xa, xb)
    xa, ita = ctls[0].simulate_device(xa.cpu().numpy(), steps, dev, torch.float32)
    xb, itb = ctls[1].simulate_graph(xb.cpu().numpy(), steps, dev, torch.float32)
    assert abs(ita - itb) < 1e-9
    np.testing.assert_allclose(xb.cpu().numpy(), xa.cpu().numpy(), rtol=0, atol=1e-6)
    # host loop: step() = qp_vectors + update + warm-started solve, plant step in numpy
    x = x0.copy()
    for _ in range(steps + 1):
        u0, res = ctls[2].step(x)
        assert all(s == "solved" for s in res.info.status)
        x = x @ ctls[2].Ad.T + u0 @ ctls[2].Bd.T
    np.testing.assert_allclose(xa.cpu().double().numpy(), x, rtol=0, atol=2e-3 * max(1.0, np.abs(x).max()))
    assert np.linalg.norm(x, axis=1).mean() < np.linalg.norm(x0, axis=1).mean()      # the loop regulates


def test_solver_destroyed_during_graph_capture_keeps_the_capture_valid():
    """A solver object dies (rqp_destroy -> hipFree of its workspace) while this thread captures the control step of ANOTHER
    solver: the capture must stay valid and replay correctly (free_ws frees under the relaxed capture mode; the capture
    itself runs in the thread-local error mode).  Round-2 finding: a finaliser's hipFree invalidated a global-mode capture."""
    import reluqp.reluqpth as reluqpth
    dev = torch.device("cuda:0")
    ctl, x0 = _setup("condensed", nx=6, nu=2, N=10, seed=8, B=32)
    ctl.simulate_device(x0, 1, dev, torch.float32)                    # sets the solver up
    g, l, u = ctl.qp_vectors(x0)
    victim = reluqpth.ReLU_QP()
    victim.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=torch.float32)
    victim.solve()
    solver = ctl.solver
    solver.synchronous = False
    out = {}
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        out["r"] = solver.solve()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        victim._destroy()                                             # hipFree x ~25 in the middle of the capture
        out["r"] = solver.solve()
    del victim
    graph.replay()
    torch.cuda.synchronize()
    solver.synchronous = True
    ref = solver.solve()                                              # eager re-solve of the same (warm) state
    assert bool((out["r"].info.status_code == 0).all()) and bool((ref.info.status_code == 0).all())
    assert float((out["r"].x - ref.x).abs().max()) < 1e-3
