"""GPU tests of the rows SURVEY.md 8(f) marks "next" and of the round-1 ADVICE items, through the C-ABI:

  f2  benchmarks/random_qps.py harness (reference random_qps.py:47-81) with the oracle as the cross-solver check
  f4  update(Hx=, Ax=)  (rejected upstream, reluqpth.py:176-177) against an oracle re-setup with the carried state
  shared-matrix batches with heterogeneous equality rows are refused (K is built per matrix)
  non-symmetric H is symmetrised once at pack time: every kernel solves the same problem
  adaptive_rho=False still checks convergence (Q3 fixed) -- iteration count / status pinned against the oracle
  fixed-k parity (rqp_iterate / rqp_compute_residuals) of the big resident tile
"""
import numpy as np
import pytest
import torch

from oracle import reluqp_oracle as O
from reluqp import _cabi, utils

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _np(t):
    return t.detach().cpu().double().numpy()


def _solver(H, g, A, l, u, **kw):
    import reluqp.reluqpth as reluqpth
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=DEV, **kw)
    return m


# ------------------------------------------------------------------------------------------ f2
def test_random_qps_harness_vs_oracle():
    """The reference's benchmark loop (rand_qp(nx, nx/4, nx/4), status must be "solved", float64, tol 1e-4) with the
    oracle (tight tolerance, float64) standing in for the OSQP agreement check of random_qps.py:68."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rqp_bench_random_qps",
                                                  os.path.join(here, "reluqp-py_amd", "benchmarks", "random_qps.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    checked = []

    def check(nx, seed, tol, x):
        H, g, A, l, u, _ = utils.rand_qp(nx=nx, n_eq=nx // 4, n_ineq=nx // 4, seed=seed, compute_sol=False)
        qp = O.OracleQP(form="factored")
        qp.setup(H, g, A, l, u, eps_abs=1e-9, max_iter=20000)
        r = qp.solve()
        assert r.info.status == "solved"
        assert np.linalg.norm(_np(x) - r.x, ord=np.inf) < 10 * tol           # random_qps.py:68 with the oracle as "OSQP"
        checked.append((nx, seed))

    bench = mod.Random_QP_benchmark(precision=torch.float64)
    rows = bench.random_initial_solve(nx_min=10, nx_max=100, n_sample=4, n_seeds=3, tol=1e-4, check=check)
    assert [r["nx"] for r in rows] == [10, 21, 46, 100] and len(checked) == 12
    assert all(r["reluqpth_mean_s"] > 0 for r in rows)
    # float32 column (the MI355X default precision) on the same sweep
    bench32 = mod.Random_QP_benchmark(precision=torch.float32)
    rows32 = bench32.random_initial_solve(nx_min=10, nx_max=100, n_sample=4, n_seeds=3, tol=1e-4, check=check)
    assert len(rows32) == 4
    # the batched extension: all seeds of one size in ONE call
    out = bench.batched_solve(46, 64, tol=1e-4)
    assert out["solved_frac"] == 1.0


# ------------------------------------------------------------------------------------------ f4
@pytest.mark.parametrize("prec,tol", [(torch.float64, 1e-7), (torch.float32, 5e-5)])
@pytest.mark.parametrize("n,n_eq,n_ineq,B", [(10, 3, 12, 6), (100, 25, 275, 4)])
def test_update_matrices_vs_oracle(prec, tol, n, n_eq, n_ineq, B):
    """update(Hx=, Ax=): pack -> gram -> factor -> kernel images re-run on the device, state kept.  The oracle defines the
    semantics (the reference asserts): new matrices, K ladder rebuilt with the equality pattern of setup, carried state."""
    H, g, A, l, u, xs = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=40, feasible=True)
    rs = np.random.RandomState(1)
    M = 0.1 * rs.randn(B, n, n)
    H2 = H + np.einsum("bij,bkj->bik", M, M)                        # still SPD
    # perturb A orthogonally to the planted point: A2 xs = A xs, so the planted point stays feasible with the same active
    # set (a free perturbation of the degenerate planted vertex -- more active rows than free dimensions -- is infeasible)
    dA = 0.05 * rs.randn(*A.shape)
    dA -= np.einsum("bmn,bn->bm", dA, xs)[:, :, None] * xs[:, None, :] / np.einsum("bn,bn->b", xs, xs)[:, None, None]
    A2 = A + dA
    m = _solver(H, g, A, l, u, precision=prec)

    def snap():                       # solve() returns the solver's ONE Results object (as the reference): copy what is compared
        r = m.solve()
        return dict(x=_np(r.x), it=r.info.iter.cpu().numpy().copy(), status=list(r.info.status))

    r0 = snap()
    assert all(s == "solved" for s in r0["status"])
    m.update(Hx=H2)                                                  # H only
    r1 = snap()
    m.update(Ax=A2)                                                  # then A only
    r2 = snap()
    m.update(Hx=H, Ax=A)                                             # both, back to the original
    r3 = snap()
    refs = []
    for b in range(B):
        qp = O.OracleQP(form="factored")
        qp.setup(H[b], g[b], A[b], l[b], u[b])
        a0 = qp.solve()
        qp.update(Hx=H2[b])
        a1 = qp.solve()
        a1 = (a1.x.copy(), a1.info.iter, a1.info.status)
        qp.update(Ax=A2[b])
        a2 = qp.solve()
        a2 = (a2.x.copy(), a2.info.iter, a2.info.status)
        qp.update(Hx=H[b], Ax=A[b])
        a3 = qp.solve()
        refs.append((a1, a2, (a3.x.copy(), a3.info.iter, a3.info.status)))
    for k, r in enumerate((r1, r2, r3)):
        it = r["it"]
        it_ref = np.array([refs[b][k][1] for b in range(B)])
        x_ref = np.stack([refs[b][k][0] for b in range(B)])
        assert r["status"] == [refs[b][k][2] for b in range(B)]
        if prec == torch.float64:
            assert np.array_equal(it, it_ref)
        else:
            assert np.all(np.abs(it - it_ref) <= 25)
        same = it == it_ref
        np.testing.assert_allclose(r["x"][same], x_ref[same], rtol=0, atol=tol * max(1.0, np.abs(x_ref).max()))
    # warm start pays: the re-solves after a small matrix change are not slower than the cold solve
    assert r1["it"].mean() <= r0["it"].mean()
    with pytest.raises(ValueError):
        m.update(Hx=np.eye(n + 1))


def test_update_matrices_shared_batch_mfma_and_wave():
    """Shared (H, A): one matrix update serves the whole batch; the MFMA images and the resident images are re-packed."""
    from reluqp import mpc
    Ad, Bd = mpc.random_plant(6, 2, seed=7)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(6), 0.1 * np.eye(2), 10, 0.4, 8.0, form="condensed")
    x0 = 1.5 * np.random.RandomState(7).randn(48, 6)
    g, l, u = ctl.qp_vectors(x0)
    H2 = ctl.H + 0.3 * np.eye(ctl.H.shape[0])
    ref = O.solve_batch(H2, g, ctl.A, l, u, form="factored", eps_abs=1e-3)
    for kern in ("mfma", "auto", "resident", "generic"):
        m = _solver(ctl.H, g, ctl.A, l, u, precision=torch.float32, kernel=kern, warm_starting=False)
        m.solve()
        m.update(Hx=H2)
        r = m.solve()                                                # cold (warm_starting=False): comparable with a fresh oracle
        it = r.info.iter.cpu().numpy()
        assert list(r.info.status) == ref["status"], kern
        assert np.mean(it == ref["iter"]) >= 0.85, kern
        same = it == ref["iter"]
        np.testing.assert_allclose(_np(r.x)[same], ref["x"][same], rtol=0, atol=1e-4 * max(1.0, np.abs(ref["x"]).max()))


# ------------------------------------------------------------------- ADVICE: shared c pattern
def test_shared_matrices_heterogeneous_equality_rows_refused():
    n, m_, B = 8, 12, 5
    rs = np.random.RandomState(0)
    M = rs.randn(n, n)
    H, A = M.T @ M + np.eye(n), rs.randn(m_, n)
    g = rs.randn(B, n)
    l = -np.ones((B, m_))
    u = np.ones((B, m_))
    l[3, 2] = u[3, 2] = 0.25                                         # instance 3 alone has row 2 as an equality
    import reluqp.reluqpth as reluqpth
    mdl = reluqpth.ReLU_QP()
    with pytest.raises(_cabi.RqpError) as ei:
        mdl.setup(H, g, A, l, u, device=DEV, precision=torch.float32)
    assert ei.value.code == _cabi.RQP_ERR_UNSUPPORTED and "equalit" in str(ei.value)
    # batched matrices carry a K per instance: accepted, and equal to the oracle
    Hb, Ab = np.broadcast_to(H, (B, n, n)).copy(), np.broadcast_to(A, (B, m_, n)).copy()
    mdl.setup(Hb, g, Ab, l, u, device=DEV, precision=torch.float64)
    r = mdl.solve()
    ref = O.solve_batch(Hb, g, Ab, l, u, form="factored")
    assert np.array_equal(r.info.iter.cpu().numpy(), ref["iter"])
    np.testing.assert_allclose(_np(r.x), ref["x"], rtol=1e-6, atol=1e-7)
    # a homogeneous pattern (same equality rows everywhere) is fine when shared
    l[:, 2] = u[:, 2] = 0.25
    mdl.setup(H, g, A, l, u, device=DEV, precision=torch.float32)
    assert all(s == "solved" for s in mdl.solve().info.status)


# ------------------------------------------------------------------- ADVICE: non-symmetric H
@pytest.mark.parametrize("n,m_,kernels", [(12, 20, ("generic", "wave", "resident")), (60, 100, ("generic", "resident"))])
def test_nonsymmetric_H_is_symmetrised(n, m_, kernels):
    """x'Hx only sees sym(H) = (H + H')/2: the pack kernel stores sym(H), so every kernel (whatever the dispatch picks)
    and the oracle on sym(H) agree.  (The reference would iterate with the raw H: not a QP gradient.)"""
    rs = np.random.RandomState(2)
    B = 4
    M = rs.randn(n, n)
    Hs = M.T @ M + np.eye(n)
    N = rs.randn(n, n)
    H = Hs + 0.3 * (N - N.T)                                         # same symmetric part, non-symmetric matrix
    A = rs.randn(m_, n)
    g = rs.randn(B, n)
    l, u = -np.ones((B, m_)), np.ones((B, m_))
    ref = O.solve_batch(Hs, g, A, l, u, form="factored")
    for kern in kernels:
        for shared in (True, False):
            Hin = H if shared else np.broadcast_to(H, (B, n, n)).copy()
            Ain = A if shared else np.broadcast_to(A, (B, m_, n)).copy()
            m = _solver(Hin, g, Ain, l, u, precision=torch.float32, kernel=kern)
            r = m.solve()
            it = r.info.iter.cpu().numpy()
            assert np.all(np.abs(it - ref["iter"]) <= 25), kern
            same = it == ref["iter"]
            assert same.sum() >= B - 1
            np.testing.assert_allclose(_np(r.x)[same], ref["x"][same], rtol=0, atol=5e-5 * max(1.0, np.abs(ref["x"]).max()))
            np.testing.assert_allclose(_np(r.info.obj_val)[same], ref["obj_val"][same], rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------- ADVICE: adaptive_rho=False
@pytest.mark.parametrize("prec", [torch.float64, torch.float32])
def test_adaptive_rho_false_checks_convergence(golden, prec):
    """Q3 (deliberate deviation): with adaptive_rho=False the reference never checks convergence (reluqpth.py:218) and
    returns its initial zero x after max_iter iterations; the build checks every check_interval iterations on the single
    rho of the ladder.  Pinned against the oracle with the same fix (quirks=False) and contrasted with quirks=True."""
    g3 = golden("g3_c1_feasible.npz")
    H, g, A, l, u = (g3["s0_" + k] for k in ("H", "g", "A", "l", "u"))
    ref = O.OracleQP(form="factored", quirks=False)
    ref.setup(H, g, A, l, u, adaptive_rho=False, rho=1.0)
    rr = ref.solve()
    m = _solver(H, g, A, l, u, precision=prec, adaptive_rho=False, rho=1.0)
    r = m.solve()
    assert len(m.layers.rhos) == 1 and r.info.rho_ind == 0
    assert r.info.status == rr.info.status == "solved"
    assert r.info.iter == rr.info.iter and r.info.iter < 4000
    np.testing.assert_allclose(_np(r.x), rr.x, rtol=0, atol=(1e-8 if prec == torch.float64 else 2e-5) * max(1.0, np.abs(rr.x).max()))
    quirk = O.OracleQP(form="factored", quirks=True)
    quirk.setup(H, g, A, l, u, adaptive_rho=False, rho=1.0, max_iter=200)
    rq = quirk.solve()
    assert rq.info.status == "max_iters_reached" and rq.info.iter == 200      # what the reference does


# ------------------------------------------------------------------- big resident tile, fixed k
def test_resident_big_tile_fixed_k_and_residuals_vs_oracle():
    """rqp_iterate / rqp_compute_residuals (modes 1 / 2 of k_admm_res2) on the n <= 104, m <= 320 tile against the
    oracle's fixed-k states (forward, reluqpth.py:80-89) and compute_residuals (:307-318) at n=100, m=300."""
    H, g, A, l, u, _ = utils.rand_qp(nx=100, n_eq=25, n_ineq=275, seed=1, feasible=True)
    qp = O.OracleQP(form="factored")
    qp.setup(H, g, A, l, u)
    m = _solver(H, g, A, l, u, precision=torch.float32, kernel="resident")
    assert m.kernel == "resident2"
    for k in (1, 4, 20):
        s_ref = qp.iterate(k).copy()
        s = _np(m.iterate(k))
        np.testing.assert_allclose(s, s_ref, rtol=0, atol=5e-5 * max(1.0, np.abs(s_ref).max()))
    n, mm = 100, 300
    x, z, lam = s_ref[:n], s_ref[n:n + mm], s_ref[n + mm:]
    pri, dua, rho = O.compute_residuals(H, A, g, x, z, lam, 0.7, 1e-6, 1e6)
    p, d, r, J = [float(v) for v in m.compute_residuals(0.7)]
    np.testing.assert_allclose(p, pri, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(d, dua, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(r, rho, rtol=5e-3)
    np.testing.assert_allclose(J, O.compute_J(H, g, x), rtol=1e-4)


# ------------------------------------------------------------------- fp16 K tile (BASELINE config 5)
@pytest.mark.parametrize("shape", ["c3", "dense", "c3_mfma"])
def test_fp16_tile_same_exits_as_float32(shape):
    """iterate_dtype=float16 (C-ABI rqp_dims.tile_dtype = RQP_TILE_F16): K(rho) stored as fp16 in the resident kernel.
    K only preconditions the residual correction, so the fixed point is the float32 one: every instance solved, the
    same iteration counts on >= 85 % of the batch (stated tolerance; fp16 rounding of K shifts marginal checks), x within
    eps_abs-level of the float32-tile run where the exits agree, KKT residuals re-derived in float64 under the thresholds."""
    import reluqp.reluqpth as reluqpth
    if shape in ("c3", "c3_mfma"):
        from reluqp import mpc
        Ad, Bd = mpc.random_plant(12, 4, seed=0)
        ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="condensed")
        x0 = np.random.RandomState(3).randn(96 if shape == "c3" else 2064, 12)
        g, l, u = ctl.qp_vectors(x0)
        H, A = ctl.H, ctl.A
    else:
        H, g, A, l, u, _ = utils.rand_qp_batch(48, 100, 25, 275, seed0=500, feasible=True)
    out = {}
    for tile in (None, torch.float16):
        m = reluqpth.ReLU_QP()
        m.setup(H, g, A, l, u, device=DEV, precision=torch.float32, eps_abs=1e-3, iterate_dtype=tile,
                kernel="auto" if shape == "c3_mfma" else "resident")
        assert m.kernel == ("mfma" if shape == "c3_mfma" else "resident2")      # (the MFMA image takes the same rounded K)
        out[tile] = m.solve()
    r32, r16 = out[None], out[torch.float16]
    assert all(s == "solved" for s in r16.info.status) and all(s == "solved" for s in r32.info.status)
    i32, i16 = r32.info.iter.cpu().numpy(), r16.info.iter.cpu().numpy()
    assert np.mean(i32 == i16) >= 0.85 and np.all(np.abs(i32 - i16) <= 50)
    same = i32 == i16
    scale = max(1.0, float(r32.x.abs().max()))
    np.testing.assert_allclose(_np(r16.x)[same], _np(r32.x)[same], rtol=0, atol=2e-3 * scale)
    Hd, Ad_, gd = (torch.as_tensor(t, device=DEV, dtype=torch.float64) for t in (H, A, g))
    x, z, y = r16.x.double(), r16.z.double(), r16.y.double()
    if Hd.dim() == 2:
        pri, dua = (x @ Ad_.T - z).abs().amax(1), (x @ Hd.T + y @ Ad_ + gd).abs().amax(1)
    else:
        pri = (torch.einsum("bmn,bn->bm", Ad_, x) - z).abs().amax(1)
        dua = (torch.einsum("bij,bj->bi", Hd, x) + torch.einsum("bmn,bm->bn", Ad_, y) + gd).abs().amax(1)
    mm, nn = A.shape[-2], H.shape[-1]
    assert float(pri.max()) < 1e-3 * np.sqrt(mm) * 1.02 and float(dua.max()) < 1e-3 * np.sqrt(nn) * 1.05
    # the tile request is refused where no kernel implements it (nothing falls back silently)
    with pytest.raises(_cabi.RqpError) as ei:
        reluqpth.ReLU_QP().setup(H, g, A, l, u, device=DEV, precision=torch.float32, iterate_dtype=torch.float16, kernel="generic")
    assert ei.value.code == _cabi.RQP_ERR_UNSUPPORTED
    with pytest.raises(ValueError):
        reluqpth.ReLU_QP().setup(H, g, A, l, u, device=DEV, precision=torch.float64, iterate_dtype=torch.float16)


# ------------------------------------------------------------------------------------------ f3
def _badly_scaled(B, n, n_eq, n_ineq, seed0):
    """rand_qp with the variables rescaled by 1e-2 .. 1e2 and the rows by 1e-1 .. 1e1: same optimum x / s."""
    H, g, A, l, u, xs = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=seed0, feasible=True)
    s = np.logspace(-2, 2, n)
    r = np.logspace(-1, 1, n_eq + n_ineq)
    Hb = H * s[None, :, None] * s[None, None, :]
    Ab = A * s[None, None, :] * r[None, :, None]
    return Hb, g * s[None], Ab, l * r[None], u * r[None], xs / s[None]


@pytest.mark.parametrize("prec", [torch.float64, torch.float32])
@pytest.mark.parametrize("kernel,n,n_eq,n_ineq", [("generic", 20, 5, 30), ("wave", 20, 5, 30), ("resident", 60, 15, 100)])
def test_eps_rel_matches_oracle(prec, kernel, n, n_eq, n_ineq):
    """eps_rel > 0 (C-ABI rqp_settings.eps_rel): OSQP-style thresholds eps_abs sqrt(dim) + eps_rel * scale.  Same exits as
    the oracle with the same extension; eps_rel = 0 is bit-for-bit the reference's test (every other test of this suite)."""
    B = 6
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=70, feasible=True)
    kw = dict(eps_abs=1e-7, eps_rel=1e-3)
    m = _solver(H, g, A, l, u, precision=prec, kernel=kernel, **kw)
    r = m.solve()
    ref = O.solve_batch(H, g, A, l, u, form="factored", **kw)
    ref0 = O.solve_batch(H, g, A, l, u, form="factored", eps_abs=1e-7)
    assert np.all(ref["iter"] < ref0["iter"])                         # the relative term ends these solves earlier
    it = r.info.iter.cpu().numpy()
    assert list(r.info.status) == ref["status"]
    if prec == torch.float64:
        assert np.array_equal(it, ref["iter"])
    else:
        assert np.mean(it == ref["iter"]) >= 0.8 and np.all(np.abs(it - ref["iter"]) <= 50)
    same = it == ref["iter"]
    np.testing.assert_allclose(_np(r.x)[same], ref["x"][same], rtol=0,
                               atol=(1e-7 if prec == torch.float64 else 5e-5) * max(1.0, np.abs(ref["x"]).max()))
    m.update_settings(eps_rel=0.0, eps_abs=1e-3)                      # changeable after setup
    assert all(s == "solved" for s in m.solve().info.status)


def test_eps_rel_mfma_shared_batch():
    from reluqp import mpc
    Ad, Bd = mpc.random_plant(6, 2, seed=7)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(6), 0.1 * np.eye(2), 10, 0.4, 8.0, form="condensed")
    x0 = 1.5 * np.random.RandomState(7).randn(40, 6)
    g, l, u = ctl.qp_vectors(x0)
    kw = dict(eps_abs=1e-6, eps_rel=1e-3)
    ref = O.solve_batch(ctl.H, g, ctl.A, l, u, form="factored", **kw)
    m = _solver(ctl.H, g, ctl.A, l, u, precision=torch.float32, kernel="mfma", **kw)
    r = m.solve()
    it = r.info.iter.cpu().numpy()
    assert list(r.info.status) == ref["status"]
    assert np.mean(it == ref["iter"]) >= 0.8


@pytest.mark.parametrize("prec,tol", [(torch.float64, 1e-6), (torch.float32, 2e-4)])
@pytest.mark.parametrize("kernel,n,n_eq,n_ineq", [("generic", 20, 5, 30), ("wave", 20, 5, 30), ("resident", 60, 15, 100)])
def test_ruiz_scaling_vs_oracle(prec, tol, kernel, n, n_eq, n_ineq):
    """settings.scaling = k Ruiz passes at setup (the reference's `scaling` is an unused TODO, reluqpth.py:105): the
    equilibrated problem is solved, every ABI boundary converts.  Against the oracle with the same scaling: same exits,
    same un-scaled x, z, y, objective; warm_start / get_state / update round-trip in the caller's space."""
    B = 5
    H, g, A, l, u, xs = _badly_scaled(B, n, n_eq, n_ineq, seed0=90)
    # (float32: the caller-space residual of these problems -- variables over four decades -- is the scaled residual times up to
    #  1e2: its float32 noise floor sits near 1e-4 * sqrt(dim), so the float32 runs certify eps_abs = 1e-3)
    kw = dict(eps_abs=1e-5 if prec == torch.float64 else 1e-3, max_iter=20000, scaling=10)
    m = _solver(H, g, A, l, u, precision=prec, kernel=kernel, **kw)
    r = m.solve()
    x, z, y = _np(r.x), _np(r.z), _np(r.y)
    it, obj = r.info.iter.cpu().numpy().copy(), _np(r.info.obj_val)
    ref = O.solve_batch(H, g, A, l, u, form="factored", **kw)
    assert list(r.info.status) == ref["status"] and all(s == "solved" for s in ref["status"])
    if prec == torch.float64:
        assert np.array_equal(it, ref["iter"])
    else:
        assert np.all(np.abs(it - ref["iter"]) <= 50)
    same = it == ref["iter"]
    sx = np.abs(ref["x"]).max(axis=1, keepdims=True)                  # per instance: the variables span 4 decades
    np.testing.assert_allclose(x[same] / sx[same], ref["x"][same] / sx[same], rtol=0, atol=tol)
    np.testing.assert_allclose(obj[same], ref["obj_val"][same], rtol=max(tol, 1e-5) * 10, atol=tol)
    # primal feasibility of the UN-scaled outputs in the caller's space
    Ax = np.einsum("bmn,bn->bm", A, x)
    rown = np.abs(A).max(axis=2)
    assert np.all(Ax >= l - 5e-2 * rown * np.abs(x).max()) and np.all(Ax <= u + 5e-2 * rown * np.abs(x).max())
    np.testing.assert_allclose(z / np.maximum(1.0, np.abs(Ax)), Ax / np.maximum(1.0, np.abs(Ax)), atol=5e-2)
    # state round trip in caller space, warm start from the solution: first check
    st, ri = m.get_state()
    st = _np(st)
    np.testing.assert_allclose(st[:, :n], x, rtol=1e-5, atol=1e-9)
    m2 = _solver(H, g, A, l, u, precision=prec, kernel=kernel, **kw)
    m2.warm_start(x=x, z=z, lam=y, rho=float(m.layers.rhos[int(ri[0])]))
    r2 = m2.solve()
    assert float(r2.info.iter.double().mean()) < 0.6 * it.mean()
    # update(g) goes through the scaling as well: compare with a fresh oracle on the new g
    g2 = g * 1.1
    m.update(g=g2)
    r3 = m.solve()
    ref3 = O.solve_batch(H, g2, A, l, u, form="factored", eps_abs=1e-9, max_iter=50000, scaling=10)
    np.testing.assert_allclose(_np(r3.x) / sx, ref3["x"] / sx, rtol=0, atol=50 * tol)
    # update(Hx=) with scaling: new D, E, c; vectors and state move to the new scaled space (needs both raw matrices: the
    # wrapper passes its copies).  Against the oracle doing the same from the same state.
    H3 = H * 1.5
    m.update(Hx=H3)
    r4 = m.solve()
    refs = []
    for b in range(B):
        qp = O.OracleQP(form="factored")
        qp.setup(H[b], g[b], A[b], l[b], u[b], **kw)
        qp.solve()
        qp.update(g=g2[b])
        qp.solve()
        qp.update(Hx=H3[b])
        rb = qp.solve()
        refs.append((rb.x.copy(), rb.info.iter, rb.info.status))
    assert list(r4.info.status) == [t[2] for t in refs] == ["solved"] * B
    it4, it4r = r4.info.iter.cpu().numpy(), np.array([t[1] for t in refs])
    assert np.all(np.abs(it4 - it4r) <= (0 if prec == torch.float64 else 50))
    same4 = it4 == it4r
    x4r = np.stack([t[0] for t in refs])
    np.testing.assert_allclose((_np(r4.x) / sx)[same4], (x4r / sx)[same4], rtol=0, atol=10 * tol)


@pytest.mark.parametrize("prec", [torch.float64, torch.float32])
@pytest.mark.parametrize("kernel,n,n_eq,n_ineq,shared", [("generic", 20, 5, 30, False), ("wave", 20, 5, 30, False),
                                                          ("resident", 60, 15, 100, False), ("mfma", 40, 10, 80, True)])
def test_scaled_solve_certifies_tolerances_in_caller_units(prec, kernel, n, n_eq, n_ineq, shared):
    """With Ruiz scaling the kernels iterate on the equilibrated problem, but every term of compute_residuals is taken back
    to the caller's space before its norm (OSQP's un-scaled termination): a "solved" instance meets eps_abs * sqrt(m) /
    eps_abs * sqrt(n) on the KKT residuals of the ORIGINAL problem, recomputed here in float64, and pri_res / dua_res
    report those numbers.  (Round-2 advisor finding: termination in the scaled space certified nothing in the caller's
    units -- E^-1 and (c D)^-1 reach 1e4.)"""
    if kernel == "mfma" and prec == torch.float64:
        pytest.skip("float32 kernel")
    B, eps = 6, (1e-4 if prec == torch.float64 else 2e-3)           # (float32 noise floor of the caller-space residual: see above)
    if shared:
        H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=5, compute_sol=False, feasible=True)
        qs = [utils.update_qp(H, A, n_eq, n_ineq, seed=50 + b, compute_sol=False, feasible=True) for b in range(B)]
        g, l, u = (np.stack([q[i] for q in qs]) for i in (1, 3, 4))
        sc = np.logspace(-2, 2, n)                                    # badly scaled variables: x = S y
        H, A, g = H * np.outer(sc, sc), A * sc[None, :], g * sc[None, :]
    else:
        H, g, A, l, u, _ = _badly_scaled(B, n, n_eq, n_ineq, seed0=190)
    m = _solver(H, g, A, l, u, precision=prec, kernel=kernel, eps_abs=eps, max_iter=40000, scaling=10)
    r = m.solve()
    assert all(s == "solved" for s in r.info.status)
    x, z, y = _np(r.x), _np(r.z), _np(r.y)
    if shared:
        pri = np.abs(x @ A.T - z).max(axis=1)
        dua = np.abs(x @ H.T + y @ A + g).max(axis=1)
    else:
        pri = np.abs(np.einsum("bmn,bn->bm", A, x) - z).max(axis=1)
        dua = np.abs(np.einsum("bij,bj->bi", H, x) + np.einsum("bmn,bm->bn", A, y) + g).max(axis=1)
    mm = A.shape[-2]
    slack = 1.02 if prec == torch.float64 else 1.5                    # (float32: the residual of a 1e-7-precision iterate)
    assert np.all(pri < eps * np.sqrt(mm) * slack), (pri, eps * np.sqrt(mm))
    assert np.all(dua < eps * np.sqrt(n) * slack), (dua, eps * np.sqrt(n))
    rt = 1e-6 if prec == torch.float64 else 0.5                      # (float32: noise of the scaled residual times 1 / E, 1 / (c D))
    np.testing.assert_allclose(_np(r.info.pri_res), pri, rtol=rt, atol=eps * (1e-2 if prec == torch.float64 else 0.3))
    np.testing.assert_allclose(_np(r.info.dua_res), dua, rtol=rt, atol=eps * (1e-2 if prec == torch.float64 else 0.3))


def test_scaling_helps_badly_scaled_problems():
    """Why the option exists: on a badly scaled batch the un-scaled ADMM needs far more iterations (both runs stop by the same
    rule: residuals in the caller's units)."""
    H, g, A, l, u, xs = _badly_scaled(16, 20, 5, 30, seed0=300)
    it = {}
    for sc in (0, 10):
        m = _solver(H, g, A, l, u, precision=torch.float64, eps_abs=1e-5, max_iter=40000, scaling=sc)
        r = m.solve()
        it[sc] = float(r.info.iter.double().mean())
        if sc:
            assert all(s == "solved" for s in r.info.status)
            np.testing.assert_allclose(_np(r.x) / np.abs(xs).max(axis=1, keepdims=True),
                                       xs / np.abs(xs).max(axis=1, keepdims=True), atol=5e-3)
    assert it[10] < 0.7 * it[0]


@pytest.mark.parametrize("prec", [torch.float64, torch.float32])
@pytest.mark.parametrize("kernel", ["auto", "generic", "wave", "resident", "mfma"])
def test_infeasibility_certificates(prec, kernel):
    """check_infeasibility (C-ABI rqp_settings): OSQP certificates.  The streaming kernel tests them at every check (same exit
    as the oracle) and is what kernel="auto" dispatches to with this option; the register-resident / MFMA kernels, on explicit
    request, run their budget and a certificate pass labels them afterwards."""
    if kernel == "mfma" and prec == torch.float64:
        pytest.skip("float32 kernel")
    n, m_ = 6, 9
    rs = np.random.RandomState(4)
    Mx = rs.randn(n, n)
    H = Mx.T @ Mx + np.eye(n)
    A = np.vstack([np.eye(n), rs.randn(m_ - n, n)])
    A[n] = A[0]                                                       # row n repeats x_0
    B = 4
    g = rs.randn(B, n)
    l = np.tile(np.r_[-np.ones(n), -2 * np.ones(m_ - n)], (B, 1))
    u = np.tile(np.r_[np.ones(n), 2 * np.ones(m_ - n)], (B, 1))
    # instances 1 and 3: x_0 <= 1 (row 0) against x_0 >= 3 (row n) -> primal infeasible; 0 and 2 stay feasible
    for b in (1, 3):
        l[b, n], u[b, n] = 3.0, np.inf
    kw = dict(check_infeasibility=True, max_iter=400)
    mdl = _solver(H, g, A, l, u, precision=prec, kernel=kernel, **kw)
    if kernel == "auto":
        assert mdl.kernel == "generic"
        kernel = "generic"
    r = mdl.solve()
    ref = O.solve_batch(H, g, A, l, u, form="factored", **kw)
    assert ref["status"] == ["solved", "primal_infeasible", "solved", "primal_infeasible"]
    assert list(r.info.status) == ref["status"]
    it = r.info.iter.cpu().numpy()
    if kernel == "generic":
        assert np.all(it[[1, 3]] < 400)                              # early exit at the check where the certificate holds
        if prec == torch.float64:
            assert np.array_equal(it, ref["iter"])
    else:
        assert np.all(it[[1, 3]] == 400)                             # labelled after the budget is spent
    assert np.array_equal(it[[0, 2]], ref["iter"][[0, 2]]) or prec == torch.float32
    # dual infeasible (unbounded below): zero curvature along x_0, cost pushes it to +inf, no upper bound
    H2 = H.copy()
    H2[0, :] = 0.0
    H2[:, 0] = 0.0
    A2 = np.eye(n)
    g2 = rs.randn(B, n)
    g2[:, 0] = -1.0
    l2 = -np.ones((B, n))
    u2 = np.ones((B, n))
    u2[:, 0] = np.inf
    mdl2 = _solver(H2, g2, A2, l2, u2, precision=prec, kernel=kernel, **kw)
    r2 = mdl2.solve()
    ref2 = O.solve_batch(H2, g2, A2, l2, u2, form="factored", **kw)
    assert ref2["status"] == ["dual_infeasible"] * B
    assert list(r2.info.status) == ref2["status"]
    # off (the default): the reference's behaviour -- the budget is spent, "max_iters_reached"
    mdl3 = _solver(H, g, A, l, u, precision=prec, kernel=kernel, max_iter=100)
    assert list(mdl3.solve().info.status)[1] == "max_iters_reached"
    # warm_starting=False + certificate pass: the state the pass needed is cleared afterwards
    mdl4 = _solver(H, g, A, l, u, precision=prec, kernel=kernel, warm_starting=False, **kw)
    mdl4.solve()
    st, ri = mdl4.get_state()
    assert float(st.abs().max()) == 0.0 and int(ri.min()) == int(ri.max()) == 7


@pytest.mark.parametrize("kernel,prec,shape,expect", [
    ("generic", torch.float32, (12, 3, 17), "generic"), ("wave", torch.float32, (12, 3, 17), "wave"),
    ("resident", torch.float32, (12, 3, 17), "resident2"),
    # the register-level check reductions (rqp_lanes.h: NaN flags as a bit mask beside v_max): float64 wavefront kernel,
    # float64 resident kernel, and the two-wavefront instantiation, whose maxima also cross waves through LDS
    ("wave", torch.float64, (12, 3, 17), "wave"), ("resident", torch.float64, (12, 3, 17), "resident64"),
    ("wave", torch.float32, (60, 10, 90), "wave")])
def test_nan_status(kernel, prec, shape, expect):
    """A NaN in the data poisons the residuals: the reference keeps iterating and reports max_iters_reached with NaN
    residuals (Q17); the build labels it nan_detected (status code 2) -- same loop control, only the label differs."""
    H, g, A, l, u, _ = utils.rand_qp_batch(3, *shape, seed0=5, feasible=True)
    g[1, 4] = np.nan
    m = _solver(H, g, A, l, u, precision=prec, kernel=kernel, max_iter=75)
    assert m.kernel == expect
    r = m.solve()
    st = list(r.info.status)
    assert st[1] == "nan_detected" and st[0] != "nan_detected" and st[2] != "nan_detected"
    assert int(r.info.iter[1]) == 75 and bool(torch.isnan(r.info.pri_res[1]) | torch.isnan(r.info.dua_res[1]))
    ref = O.solve_batch(H, g, A, l, u, form="factored", max_iter=75)
    assert ref["status"][1] == "nan_detected" and ref["iter"][1] == 75


# ------------------------------------------------------------------- devices=[...]
@pytest.mark.parametrize("shared", [False, True])
def test_devices_list_splits_the_batch(shared):
    """setup(devices=[0, 0, 0]): three shards (one handle + stream each; on this one-GPU box all on device 0), results
    gathered in batch order -- bit-identical to the single-handle solve; update / warm_start / get_state go to the shards."""
    B, n, n_eq, n_ineq = 50, 20, 5, 30                                # 50 = 17 + 17 + 16: ragged split
    if shared:
        H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=5, compute_sol=False, feasible=True)
        qs = [utils.update_qp(H, A, n_eq, n_ineq, seed=50 + b, compute_sol=False, feasible=True) for b in range(B)]
        g, l, u = (np.stack([q[i] for q in qs]) for i in (1, 3, 4))
    else:
        H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=11, feasible=True)
    one = _solver(H, g, A, l, u, precision=torch.float32)
    r1 = one.solve()
    x1, it1, y1 = r1.x.clone(), r1.info.iter.clone(), r1.y.clone()
    import reluqp.reluqpth as reluqpth
    multi = reluqpth.ReLU_QP()
    multi.setup(H, g, A, l, u, precision=torch.float32, devices=[0, 0, 0])
    assert [sz for _, sz in multi._shards.ranges] == [17, 17, 16]
    rm = multi.solve()
    assert torch.equal(rm.x, x1) and torch.equal(rm.info.iter, it1) and torch.equal(rm.y, y1)
    assert list(rm.info.status) == list(r1.info.status)
    for f in ("pri_res", "dua_res", "rho_estimate", "obj_val"):      # the same dtypes on both paths (float64, as the kernels write them)
        assert getattr(rm.info, f).dtype == getattr(r1.info, f).dtype == torch.float64
        assert torch.equal(getattr(rm.info, f), getattr(r1.info, f))
    assert multi.results.info.setup_time > 0
    g2 = g * 0.9
    one.update(g=g2)
    multi.update(g=g2)
    assert torch.equal(multi.solve().x, one.solve().x)
    sm, rim = multi.get_state()
    s1, ri1 = one.get_state()
    assert torch.equal(sm, s1) and torch.equal(rim, ri1)
    multi.clear_primal_dual()
    assert float(multi.get_state()[0].abs().max()) == 0.0
    with pytest.raises(ValueError):
        reluqpth.ReLU_QP().setup(H if shared else H[0], g[0], A if shared else A[0], l[0], u[0], devices=[0])   # un-batched


def test_devices_list_with_windowed_shards():
    """Shards large enough for the rho-ladder window (>= 32 per-instance matrices each): every shard's rqp_solve reads its
    continue count back on its own host thread; results equal the single-handle solve bit for bit, also for problems that
    walk out of their windows (the reference's signed-slack generator at m = 2n ratchets rho upwards)."""
    import reluqp.reluqpth as reluqpth
    B, n, n_eq, n_ineq = 130, 20, 10, 30
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=21, feasible=False, dtype=np.float32)
    one = reluqpth.ReLU_QP()
    one.setup(H, g, A, l, u, precision=torch.float32, device=DEV, kernel="resident", max_iter=500)
    multi = reluqpth.ReLU_QP()
    multi.setup(H, g, A, l, u, precision=torch.float32, devices=[0, 0, 0], kernel="resident", max_iter=500)
    assert one.get_window()[0] == 5 and all(c.get_window()[0] == 5 for c in multi._shards.children)
    r1, rm = one.solve(), multi.solve()
    assert int((r1.info.rho_ind > 10).sum()) > 10                     # the windows moved
    assert torch.equal(rm.x, r1.x) and torch.equal(rm.info.iter, r1.info.iter) and torch.equal(rm.info.rho_ind, r1.info.rho_ind)
    r1, rm = one.solve(), multi.solve()                               # warm re-solve from the moved indices
    assert torch.equal(rm.x, r1.x) and torch.equal(rm.info.iter, r1.info.iter)


# ------------------------------------------------------------------- float64 resident kernel (the reference's precision)
def test_resident64_equals_streaming_float64():
    """k_admm_res64 (A, K in registers, one CU per QP) against k_admm_generic<double> (streaming) at the headline size in the
    reference's default precision: identical exits, iteration counts and rho indices; x, z, y to rounding (1e-9 stated:
    same recurrence, different summation order of the matrix-vector products)."""
    B, n, n_eq, n_ineq = 24, 100, 25, 275
    H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=800, feasible=True)
    mr = _solver(H, g, A, l, u, precision=torch.float64)
    mg = _solver(H, g, A, l, u, precision=torch.float64, kernel="generic")
    assert mr.kernel == "resident64" and mg.kernel == "generic"
    rr, rg = mr.solve(), mg.solve()
    assert list(rr.info.status) == list(rg.info.status) and all(s == "solved" for s in rg.info.status)
    assert torch.equal(rr.info.iter, rg.info.iter)
    assert torch.equal(rr.info.rho_ind, rg.info.rho_ind)
    for a_, b_, w in ((rr.x, rg.x, 1.0), (rr.z, rg.z, 1.0), (rr.y, rg.y, 100.0)):
        np.testing.assert_allclose(_np(a_), _np(b_), rtol=0, atol=w * 1e-9 * max(1.0, float(b_.abs().max())))
    np.testing.assert_allclose(_np(rr.info.obj_val), _np(rg.info.obj_val), rtol=1e-9)
    np.testing.assert_allclose(_np(rr.info.pri_res), _np(rg.info.pri_res), rtol=1e-5, atol=1e-12)
    # ragged sizes inside the tile, shared matrices, warm re-solve, max_iter off the check grid
    H2, g0, A2, l0, u0, _ = utils.rand_qp(77, 11, 200, seed=5, compute_sol=False, feasible=True)
    qs = [utils.update_qp(H2, A2, 11, 200, seed=50 + b, compute_sol=False, feasible=True) for b in range(6)]
    g2, l2, u2 = (np.stack([q[i] for q in qs]) for i in (1, 3, 4))
    ms = _solver(H2, g2, A2, l2, u2, precision=torch.float64, max_iter=130)
    mt = _solver(H2, g2, A2, l2, u2, precision=torch.float64, max_iter=130, kernel="generic")
    assert ms.kernel == "resident64"
    for _ in range(2):
        rs, rt = ms.solve(), mt.solve()
        assert torch.equal(rs.info.iter, rt.info.iter) and list(rs.info.status) == list(rt.info.status)
        np.testing.assert_allclose(_np(rs.x), _np(rt.x), rtol=0, atol=1e-9 * max(1.0, float(rt.x.abs().max())))
        np.testing.assert_allclose(_np(rs.info.rho_estimate), _np(rt.info.rho_estimate), rtol=1e-6)


# ------------------------------------------------------------------- sizes beyond every resident tile
@pytest.mark.parametrize("prec", [torch.float64, torch.float32])
@pytest.mark.parametrize("n,n_eq,n_ineq", [(120, 30, 200), (136, 30, 250), (200, 40, 360)])
def test_large_sizes_streaming_kernel_and_factor_fallbacks(prec, n, n_eq, n_ineq):
    """n > 104: the streaming ADMM kernel, and the three factorisation paths behind the register-resident one -- LDS
    Gauss-Jordan with 128 columns (n <= 128), LDS Gauss-Jordan (n*n doubles fit LDS: n <= 141), global-scratch (larger)."""
    B = 2
    H, g, A, l, u, xs = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=600 + n, feasible=True)
    m = _solver(H, g, A, l, u, precision=prec)
    assert m.kernel == "generic"
    r = m.solve()
    ref = O.solve_batch(H, g, A, l, u, form="factored")
    it = r.info.iter.cpu().numpy()
    assert list(r.info.status) == ref["status"] == ["solved"] * B
    if prec == torch.float64:
        assert np.array_equal(it, ref["iter"])
        np.testing.assert_allclose(_np(r.x), ref["x"], rtol=0, atol=1e-8 * max(1.0, np.abs(ref["x"]).max()))
        K = _np(m.layers.K(9, 1))
        rv = O.rho_vector(float(m.layers.rhos[9]), l[1], u[1], 1e-6)
        Kref = np.linalg.inv(H[1] + 1e-6 * np.eye(n) + A[1].T @ (rv[:, None] * A[1]))
        np.testing.assert_allclose(K, Kref, rtol=1e-7, atol=1e-9 * np.abs(Kref).max())
    else:
        assert np.all(np.abs(it - ref["iter"]) <= 50)
        same = it == ref["iter"]
        np.testing.assert_allclose(_np(r.x)[same], ref["x"][same], rtol=0, atol=5e-5 * max(1.0, np.abs(ref["x"]).max()))
    np.testing.assert_allclose(_np(r.x), xs, rtol=0, atol=2e-2 * max(1.0, np.abs(xs).max()))


@pytest.mark.parametrize("kernel", ["mfma", "wave", "resident"])
def test_ruiz_scaling_shared_matrices(kernel):
    """Shared (H, A): ONE set of Ruiz factors for the whole batch (the cost scaling ignores g for that reason)."""
    from reluqp import mpc
    Ad, Bd = mpc.random_plant(6, 2, seed=7)
    ctl = mpc.LinearMPC(Ad, Bd, np.diag([100.0, 1, 1, 0.01, 1, 1]), 0.1 * np.eye(2), 10, 0.4, 8.0, form="condensed")
    x0 = 1.5 * np.random.RandomState(7).randn(40, 6)
    g, l, u = ctl.qp_vectors(x0)
    kw = dict(eps_abs=1e-4, scaling=10, max_iter=20000)
    ref = O.solve_batch(ctl.H, g, ctl.A, l, u, form="factored", **kw)
    m = _solver(ctl.H, g, ctl.A, l, u, precision=torch.float32, kernel=kernel, **kw)
    r = m.solve()
    it = r.info.iter.cpu().numpy()
    assert list(r.info.status) == ref["status"] == ["solved"] * 40
    assert np.mean(it == ref["iter"]) >= 0.8 and np.all(np.abs(it - ref["iter"]) <= 75)
    same = it == ref["iter"]
    np.testing.assert_allclose(_np(r.x)[same], ref["x"][same], rtol=0, atol=2e-4 * max(1.0, np.abs(ref["x"]).max()))
    np.testing.assert_allclose(_np(r.info.obj_val)[same], ref["obj_val"][same], rtol=1e-3, atol=1e-3)


def test_fp16_tile_survives_matrix_update():
    """update(Hx=) with the fp16 K tile: the packed tile and its power-of-two scales are rebuilt."""
    H, g, A, l, u, _ = utils.rand_qp_batch(6, 60, 15, 100, seed0=31, feasible=True)
    m = _solver(H, g, A, l, u, precision=torch.float32, iterate_dtype=torch.float16)
    assert m.kernel == "resident2"
    m.solve()
    H2 = H * 40.0                                                     # K shrinks by ~40: a stale scale would flush it
    m.update(Hx=H2, g=g * 40.0)
    r = m.solve()
    ref = O.solve_batch(H2, g * 40.0, A, l, u, form="factored", eps_abs=1e-9, max_iter=20000)
    assert all(s == "solved" for s in r.info.status)
    np.testing.assert_allclose(_np(r.x), ref["x"], rtol=0, atol=2e-2 * max(1.0, np.abs(ref["x"]).max()))


# ------------------------------------------------------------------- low_memory (rqp_dims.flags RQP_FLAG_LOW_MEMORY)
@pytest.mark.parametrize("n,n_eq,n_ineq,B,shared", [(100, 25, 275, 40, False), (13, 3, 20, 9, False), (60, 0, 128, 5, False),
                                                    (80, 20, 300, 2100, True)])
def test_low_memory_reads_K_from_the_table_bit_identical(n, n_eq, n_ineq, B, shared):
    """low_memory=True drops the packed copy of K(rho) of the resident float32 kernel and loads K from the factor kernel's
    row-major table (padding guarded per element): same K values, same kernel arithmetic -> bit-identical solves, on every
    resident tile, through update(Hx, Ax), and on the straggler hand-off of an MFMA batch (shared matrices, B >= 2048)."""
    if shared:
        H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=31, compute_sol=False, feasible=True)
        upd = [utils.update_qp(H, A, n_eq, n_ineq, seed=32 + b, compute_sol=False, feasible=True) for b in range(64)]
        idx = np.arange(B) % 64
        g, l, u = (np.stack([x[k] for x in upd])[idx] for k in (1, 3, 4))
        kern = "auto"
    else:
        H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=4100, feasible=True)
        kern = "resident"
    out = []
    for low in (False, True):
        m = _solver(H, g, A, l, u, precision=torch.float32, kernel=kern, low_memory=low)
        assert m.kernel == ("mfma" if shared else "resident2")
        r = m.solve()
        first = (r.x.clone(), r.z.clone(), r.y.clone(), r.info.iter.clone(), list(r.info.status))
        if not shared:
            m.update(Hx=H * 1.01)
            r = m.solve()
        out.append(first + (r.x.clone(), r.info.iter.clone()))
    a, b = out
    for k in (0, 1, 2, 3, 5, 6):
        assert torch.equal(a[k], b[k]), k
    assert a[4] == b[4] and all(s == "solved" for s in a[4])


# ------------------------------------------------------------------- the C ABI from a host program without Python / torch
def test_c_abi_example_program(tmp_path):
    """reluqp-py_amd/examples/c_abi_solve.cpp: hipMalloc'd buffers, rqp_create / rqp_setup / rqp_solve / rqp_destroy through
    include/rqp_abi.h only -- built with hipcc against librqp_hip.so and run as its own process (it checks its three
    closed-form minimisers itself and prints `ok`)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "c_abi_solve")
    lib = os.path.join(root, "reluqp-py_amd", "reluqp", "lib")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-I", os.path.join(root, "include"),
                    os.path.join(root, "reluqp-py_amd", "examples", "c_abi_solve.cpp"), "-L", lib, "-lrqp_hip",
                    "-Wl,-rpath," + lib, "-o", exe], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    assert out.strip().endswith("ok") and out.count("status=0") == 3, out
