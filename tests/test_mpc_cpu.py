"""Linear-MPC generators and closed-loop driver on the CPU.

The reference's loose_code/RandomLinMPC.py raises as written, so there is no
reference output to pin against: "parity unpinned" by the reference, pinned here by construction
properties -- Riccati fixed point, dynamics feasibility, sparse == condensed optimum, closed-loop
stability -- with the oracle (oracle/reluqp_oracle.py) as the QP solver."""
import numpy as np
import pytest

from oracle import reluqp_oracle as O
from reluqp import mpc


class OracleBatch(object):
    """setup/update/solve facade over one OracleQP per instance (shared H, A), mirroring ReLU_QP's batch API."""

    class _R(object):
        pass

    def setup(self, H, g, A, l, u, **kw):
        self.qps = []
        for b in range(g.shape[0]):
            q = O.OracleQP(form="factored")
            q.setup(H, g[b], A, l[b], u[b], **kw)
            self.qps.append(q)

    def update(self, g=None, l=None, u=None):
        for b, q in enumerate(self.qps):
            q.update(g=None if g is None else g[b], l=None if l is None else l[b], u=None if u is None else u[b])

    def solve(self):
        rs = [q.solve() for q in self.qps]
        out = self._R()
        out.x = np.stack([r.x for r in rs])
        out.info = self._R()
        out.info.iter = np.array([r.info.iter for r in rs])
        out.info.status = [r.info.status for r in rs]
        return out


def _plant(nx=4, nu=2, seed=1):
    Ad, Bd = mpc.random_plant(nx, nu, seed=seed)
    return Ad, Bd, np.eye(nx), 0.1 * np.eye(nu)


def test_ihlqr_riccati_fixed_point():
    Ad, Bd, Q, R = _plant(6, 2, seed=3)
    K, P = mpc.ihlqr(Ad, Bd, Q, R, Q)
    # discrete algebraic Riccati equation and the gain it implies
    Pn = Q + Ad.T @ P @ Ad - Ad.T @ P @ Bd @ np.linalg.solve(R + Bd.T @ P @ Bd, Bd.T @ P @ Ad)
    np.testing.assert_allclose(Pn, P, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(K, np.linalg.solve(R + Bd.T @ P @ Bd, Bd.T @ P @ Ad), rtol=1e-6, atol=1e-8)   # K lags P by one sweep (tol 1e-8)
    assert np.max(np.abs(np.linalg.eigvals(Ad - Bd @ K))) < 1.0          # stabilising


def test_sparse_form_structure():
    nx, nu, N = 3, 2, 4
    Ad, Bd, Q, R = _plant(nx, nu, seed=2)
    H, g, A, l, u = mpc.gen_sparse_mpc_qp(Ad, Bd, Q, R, 2 * Q, N)
    assert H.shape == (N * (nx + nu),) * 2 and A.shape == (N * nx, N * (nx + nu))
    np.testing.assert_array_equal(H[:nu, :nu], R)
    np.testing.assert_array_equal(H[nu:nu + nx, nu:nu + nx], Q)
    np.testing.assert_array_equal(H[-nx:, -nx:], 2 * Q)                 # terminal cost
    # a simulated trajectory satisfies the dynamics rows for its x0
    rs = np.random.RandomState(0)
    x0 = rs.randn(nx)
    us = rs.randn(N, nu)
    xs = [x0]
    for k in range(N):
        xs.append(Ad @ xs[-1] + Bd @ us[k])
    y = np.concatenate([np.concatenate([us[k], xs[k + 1]]) for k in range(N)])
    lb, ub = mpc.sparse_x0_update(Ad, nx, l, u, x0)
    np.testing.assert_allclose(A @ y, lb[0], atol=1e-12)
    np.testing.assert_array_equal(lb, ub)


@pytest.mark.parametrize("use_gain", [False, True])
def test_condensed_equals_sparse(use_gain):
    nx, nu, N = 4, 2, 5
    Ad, Bd, Q, R = _plant(nx, nu, seed=4)
    K, P = mpc.ihlqr(Ad, Bd, Q, R, Q)
    A_add, l_add, u_add = mpc.box_constraints(nx, nu, N, u_max=0.6, x_max=3.0)
    x0 = np.array([1.5, -1.0, 0.5, 2.0])
    Hs, gs, As, ls, us = mpc.gen_sparse_mpc_qp(Ad, Bd, Q, R, P, N, A_add, l_add, u_add)
    lb, ub = mpc.sparse_x0_update(Ad, nx, ls, us, x0)
    qs = O.OracleQP(form="factored")
    qs.setup(Hs, gs, As, lb[0], ub[0], eps_abs=1e-8, max_iter=20000)
    rs_ = qs.solve()
    Hc, gc, Ac, lc, uc, g_x0, lu_x0 = mpc.gen_condensed_mpc_qp(Ad, Bd, Q, R, P, N, A_add, l_add, u_add,
                                                                K=K if use_gain else None)
    g, l, u = mpc.condensed_x0_update(g_x0, lu_x0, l_add, u_add, x0)
    qc = O.OracleQP(form="factored")
    qc.setup(Hc, g[0], Ac, l[0], u[0], eps_abs=1e-8, max_iter=20000)
    rc = qc.solve()
    assert rs_.info.status == rc.info.status == "solved"
    u0_sparse = rs_.x[:nu]
    u0_cond = rc.x[:nu] - (K @ x0 if use_gain else 0.0)
    np.testing.assert_allclose(u0_cond, u0_sparse, atol=2e-5)
    assert np.any(np.abs(u0_sparse) > 0.59)                              # the input bound is active: a real QP


def test_closed_loop_stabilises_with_oracle_solver():
    nx, nu, N = 4, 2, 6
    Ad, Bd, Q, R = _plant(nx, nu, seed=5)
    ctl = mpc.LinearMPC(Ad, Bd, Q, R, N, u_max=0.5, x_max=5.0, form="condensed", solver=OracleBatch(), eps_abs=1e-4)
    x0 = np.array([[2.0, -1.5, 1.0, 0.5], [-1.0, 2.0, 0.0, -2.0]])
    xs, us, its = ctl.simulate(x0, steps=60)
    assert xs.shape == (61, 2, nx) and us.shape == (60, 2, nu)
    assert np.all(np.abs(us) <= 0.5 + 1e-3)                              # input box respected
    nrm = np.linalg.norm(xs, axis=2)
    assert np.all(nrm[-1] < 0.25 * nrm[0]) and np.all(nrm[-1] < nrm[30])   # regulated towards the origin
    # warm-started re-solves need fewer iterations than the cold first step
    assert its[5:].mean() < its[0].mean()
