"""CPU tests of the oracle's restatement of the build's extensions: eps_rel, Ruiz scaling,
infeasibility certificates, nan status, matrix updates.  These options are absent from the reference (TODOs at
reluqpth.py:105,176-177,233,335), so there is no reference output to pin them against: "parity unpinned" for them -- the
oracle is checked here against independent facts (planted optima, hand-built infeasible problems, invariance under
rescaling), and the HIP kernels are then checked against the oracle (tests/test_extensions_gpu.py).  With every extension
off the oracle is the pinned one (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from oracle import reluqp_oracle as O
from reluqp import utils


def test_eps_rel_zero_is_the_reference_test():
    H, g, A, l, u, _ = utils.rand_qp(10, 5, 15, seed=0, feasible=True)
    a = O.OracleQP(form="factored"); a.setup(H, g, A, l, u); ra = a.solve()
    b = O.OracleQP(form="factored"); b.setup(H, g, A, l, u, eps_rel=0.0); rb = b.solve()
    assert ra.info.iter == rb.info.iter and np.array_equal(ra.x, rb.x)
    c = O.OracleQP(form="factored"); c.setup(H, g, A, l, u, eps_abs=1e-9, eps_rel=1e-3); rc = c.solve()
    d = O.OracleQP(form="factored"); d.setup(H, g, A, l, u, eps_abs=1e-9); rd = d.solve()
    assert rc.info.status == "solved" and rc.info.iter < rd.info.iter


def test_ruiz_scaling_equilibrates_and_preserves_the_solution():
    H, g, A, l, u, xs = utils.rand_qp(20, 5, 30, seed=3, feasible=True)
    s = np.logspace(-2, 2, 20)
    Hb, Ab, gb = H * s[:, None] * s[None, :], A * s[None, :], g * s
    D, E, c, Hs, As = O.ruiz_scale(Hb, Ab, 10)
    kkt_col = np.maximum(np.abs(Hs / c).max(axis=0), np.abs(As).max(axis=0))
    assert kkt_col.max() / kkt_col.min() < 1.5 and np.abs(As).max(axis=1).max() / np.abs(As).max(axis=1).min() < 1.5
    np.testing.assert_allclose(Hs, c * (Hb * D[:, None] * D[None, :]), rtol=1e-12)
    np.testing.assert_allclose(As, Ab * E[:, None] * D[None, :], rtol=1e-12)
    it = {}
    for sc in (0, 10):
        qp = O.OracleQP(form="factored")
        qp.setup(Hb, gb, Ab, l, u, scaling=sc, eps_abs=1e-6, max_iter=20000)
        r = qp.solve()
        assert r.info.status == "solved"
        np.testing.assert_allclose(r.x * s, xs, atol=1e-4)                       # same optimum in the caller's space
        np.testing.assert_allclose(r.info.obj_val, 0.5 * r.x @ Hb @ r.x + gb @ r.x, rtol=1e-9)
        it[sc] = r.info.iter
        # warm start in caller space: the solution is a fixed point
        q2 = O.OracleQP(form="factored")
        q2.setup(Hb, gb, Ab, l, u, scaling=sc, eps_abs=1e-6, max_iter=20000)
        q2.warm_start(x=r.x, z=r.z, lam=r.y)
        assert q2.solve().info.iter <= 50
    assert it[10] <= it[0]


def test_certificates_and_nan_status():
    H = np.eye(2); g = np.zeros(2)
    A = np.array([[1.0, 0], [1, 0], [0, 1]])
    l = np.array([1.0, -np.inf, -1]); u = np.array([np.inf, 0, 1])                # x0 >= 1 and x0 <= 0
    qp = O.OracleQP(form="factored"); qp.setup(H, g, A, l, u, check_infeasibility=True)
    assert qp.solve().info.status == "primal_infeasible"
    qp = O.OracleQP(form="factored"); qp.setup(H, g, A, l, u, max_iter=100)       # off: the reference's behaviour
    assert qp.solve().info.status == "max_iters_reached"
    H2 = np.diag([0.0, 1]); g2 = np.array([-1.0, 0]); A2 = np.eye(2)
    qp = O.OracleQP(form="factored"); qp.setup(H2, g2, A2, np.array([0.0, -1]), np.array([np.inf, 1]), check_infeasibility=True)
    assert qp.solve().info.status == "dual_infeasible"
    Hf, gf, Af, lf, uf, _ = utils.rand_qp(10, 3, 12, seed=1, feasible=True)
    qp = O.OracleQP(form="factored"); qp.setup(Hf, gf, Af, lf, uf, check_infeasibility=True)
    assert qp.solve().info.status == "solved"
    gf2 = gf.copy(); gf2[2] = np.nan
    qp = O.OracleQP(form="factored"); qp.setup(Hf, gf2, Af, lf, uf, max_iter=50)
    assert qp.solve().info.status == "nan_detected"
    qq = O.OracleQP(form="factored", quirks=True); qq.setup(Hf, gf2, Af, lf, uf, max_iter=50)
    assert qq.solve().info.status == "max_iters_reached"                          # what the reference reports (Q17)
