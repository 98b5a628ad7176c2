"""Linear-MPC problem generators and a closed-loop driver.

The reference ships these as ``loose_code/RandomLinMPC.py``: ``ihlqr`` works (``:6-15``) but
``gen_sparse_mpc_qp`` (``:54-66``) and ``gen_condensed_mpc_qp`` (``:76-90``) raise ``ValueError``
as written (mistranslated block-diagonal at ``:56``, ``-eye(nu)`` for ``-I_nx`` at ``:58``, 3 of 5
return values unpacked at ``:80``) and there is no driver.  This module implements their documented
INTENT with the same function names, argument order and return tuples; since the reference code does
not run, parity is pinned by construction properties instead (tests/test_mpc_cpu.py): the Riccati
fixed point, dynamics feasibility of the sparse form, and sparse == condensed optima.

Variable order of the sparse form (``:54`` intent): y = [u_0, x_1, u_1, x_2, ..., u_{N-1}, x_N].
Condensed form: y = F v + G x0 with pre-stabilising feedback u_k = -K x_k + v_k, so that
H = F'H_sp F, g = (F'H_sp G) x0, A = A_add F, l/u = l_add/u_add - (A_add G) x0 -- per step only
(g, l, u) change: exactly the ``update(g, l, u)`` + warm-started ``solve()`` path of the solver
(reluqpth.py:159-183), with H and A shared by every instance of a batch.
"""
from copy import deepcopy

import numpy as np


def ihlqr(A, B, Q, R, Qf, max_iters=1000, tol=1e-8):
    """Infinite-horizon LQR by Riccati fixed-point iteration; returns (K, P).
    Same recursion and stopping rule as RandomLinMPC.py:6-15."""
    P = Qf
    K = np.zeros((B.shape[1], A.shape[0]))
    for _ in range(max_iters):
        P_prev = deepcopy(P)
        K = np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
        P = Q + A.T @ P @ (A - B @ K)
        if np.linalg.norm(P - P_prev, 2) < tol:
            return K, P
    raise Exception("ihlqr didn't converge")


def _blkdiag(blocks):
    n = sum(b.shape[0] for b in blocks)
    out = np.zeros((n, n))
    i = 0
    for b in blocks:
        k = b.shape[0]
        out[i:i + k, i:i + k] = b
        i += k
    return out


def gen_sparse_mpc_qp(Ad, Bd, Q, R, Qf, horizon, A_add=None, l_add=None, u_add=None):
    """Sparse-form MPC QP (intent of RandomLinMPC.py:54-66).  Returns (H, g, A, l, u) for x0 = 0;
    the x0-dependence is l[:nx] = u[:nx] = -Ad x0 (first dynamics block)."""
    nx, nu = Ad.shape[0], Bd.shape[1]
    N = horizon
    H = _blkdiag([R, Q] * (N - 1) + [R, Qf])
    g = np.zeros(H.shape[0])
    # dynamics rows k: B u_k - x_{k+1} + A x_k = 0
    A = np.kron(np.eye(N), np.hstack([Bd, -np.eye(nx)]))
    if N > 1:
        A[nx:, nu:nu + (N - 1) * (nx + nu)] += np.kron(np.eye(N - 1), np.hstack([Ad, np.zeros((nx, nu))]))
    l = np.zeros(A.shape[0])
    u = np.zeros(A.shape[0])
    if A_add is not None:
        A = np.vstack([A, A_add])
        l = np.hstack([l, l_add])
        u = np.hstack([u, u_add])
    return H, g, A, l, u


def sparse_x0_update(Ad, nx, l, u, x0):
    """(l, u) of the sparse form for an initial state x0 (batched x0 [B, nx] -> [B, m])."""
    x0 = np.atleast_2d(x0)
    lb = np.repeat(l[None], x0.shape[0], 0)
    ub = np.repeat(u[None], x0.shape[0], 0)
    lb[:, :nx] = ub[:, :nx] = -(x0 @ Ad.T)
    return lb, ub


def gen_condensed_mpc_qp(Ad, Bd, Q, R, Qf, horizon, A_add, l_add, u_add, K=None):
    """Condensed MPC QP (intent of RandomLinMPC.py:76-90): states eliminated through
    y = F v + G x0 with u_k = -K x_k + v_k.  Returns (H, g, A, l, u, g_x0, lu_x0) with
    g = g_x0 @ x0, l = l_add + lu_x0 @ x0, u = u_add + lu_x0 @ x0 (returned for x0 = 0)."""
    nx, nu = Ad.shape[0], Bd.shape[1]
    N = horizon
    if K is None:
        K = np.zeros((nu, nx))
    H_sp, g_sp, _, _, _ = gen_sparse_mpc_qp(Ad, Bd, Q, R, Qf, N)
    Acl = Ad - Bd @ K
    pw = [np.eye(nx)]
    for _ in range(N):
        pw.append(Acl @ pw[-1])
    blk = nu + nx
    F = np.zeros((N * blk, N * nu))
    G = np.zeros((N * blk, nx))
    for k in range(N):
        G[k * blk:k * blk + nu] = -K @ pw[k]
        G[k * blk + nu:(k + 1) * blk] = pw[k + 1]
        F[k * blk:k * blk + nu, k * nu:(k + 1) * nu] = np.eye(nu)
        F[k * blk + nu:(k + 1) * blk, k * nu:(k + 1) * nu] = Bd
        for j in range(k):
            F[k * blk:k * blk + nu, j * nu:(j + 1) * nu] = -K @ pw[k - 1 - j] @ Bd
            F[k * blk + nu:(k + 1) * blk, j * nu:(j + 1) * nu] = pw[k - j] @ Bd
    H = F.T @ H_sp @ F
    H = 0.5 * (H + H.T)
    g_x0 = F.T @ H_sp @ G
    g = g_x0 @ np.zeros(nx) + F.T @ g_sp
    A = A_add @ F
    lu_x0 = -A_add @ G
    return H, g, A, l_add, u_add, g_x0, lu_x0


def condensed_x0_update(g_x0, lu_x0, l_add, u_add, x0):
    """(g, l, u) of the condensed form for (a batch of) initial states x0 [B, nx]."""
    x0 = np.atleast_2d(x0)
    shift = x0 @ lu_x0.T
    return x0 @ g_x0.T, l_add[None] + shift, u_add[None] + shift


def random_plant(nx=12, nu=4, seed=0, dt=0.1):
    """Random controllable, marginally stable discrete plant:
    Ad = I + dt*S with S = (W - W')/2 - 0.05 W W'/nx (lightly damped rotation), rescaled to spectral
    radius <= 0.999 so that box-constrained MPC problems stay feasible; Bd = dt*randn."""
    rs = np.random.RandomState(seed)
    W = rs.randn(nx, nx)
    S = 0.5 * (W - W.T) - 0.05 * (W @ W.T) / nx
    Ad = np.eye(nx) + dt * S
    rad = np.max(np.abs(np.linalg.eigvals(Ad)))
    Ad = Ad / max(1.0, rad / 0.999)
    Bd = dt * rs.randn(nx, nu)
    return Ad, Bd


def box_constraints(nx, nu, horizon, u_max, x_max):
    """A_add = I on y = [u_0, x_1, ...]: |u| <= u_max, |x| <= x_max."""
    blk = nu + nx
    A_add = np.eye(horizon * blk)
    hi = np.tile(np.hstack([np.full(nu, u_max), np.full(nx, x_max)]), horizon)
    return A_add, -hi, hi


class LinearMPC(object):
    """Closed-loop linear MPC on a batch of independent plants/initial states sharing (H, A).

    setup: one ``ReLU_QP.setup`` with un-batched (shared) H, A and batched (g, l, u);
    step(x): ``update(g, l, u)`` from the current states, warm-started ``solve()``, returns the
    first input of every instance (the path of reluqpth.py:159-183 + :201-249 per control step)."""

    def __init__(self, Ad, Bd, Q, R, horizon, u_max, x_max, form="condensed", solver=None, **solver_kw):
        self.Ad, self.Bd, self.horizon, self.form = Ad, Bd, horizon, form
        self.nx, self.nu = Ad.shape[0], Bd.shape[1]
        self.K, self.P = ihlqr(Ad, Bd, Q, R, Q)
        A_add, l_add, u_add = box_constraints(self.nx, self.nu, horizon, u_max, x_max)
        self.l_add, self.u_add = l_add, u_add
        if form == "condensed":
            (self.H, _, self.A, _, _, self.g_x0, self.lu_x0) = gen_condensed_mpc_qp(
                Ad, Bd, Q, R, self.P, horizon, A_add, l_add, u_add, K=self.K)
        else:
            self.H, _, self.A, self.l0, self.u0 = gen_sparse_mpc_qp(Ad, Bd, Q, R, self.P, horizon, A_add, l_add, u_add)
        self.solver = solver
        self.solver_kw = solver_kw
        self._ready = False

    def qp_vectors(self, x):
        """(g, l, u) [B, .] of the QPs for the current states x [B, nx]."""
        x = np.atleast_2d(x)
        if self.form == "condensed":
            return condensed_x0_update(self.g_x0, self.lu_x0, self.l_add, self.u_add, x)
        lb, ub = sparse_x0_update(self.Ad, self.nx, self.l0, self.u0, x)
        return np.zeros((x.shape[0], self.H.shape[0])), lb, ub

    def first_input(self, sol, x):
        """u_0 of every instance from the QP solution (condensed: u_0 = -K x_0 + v_0)."""
        x = np.atleast_2d(x)
        if self.form == "condensed":
            return sol[:, :self.nu] - x @ self.K.T
        return sol[:, :self.nu]

    def step(self, x):
        g, l, u = self.qp_vectors(x)
        if not self._ready:
            import reluqp.reluqpth as reluqpth
            self.solver = self.solver or reluqpth.ReLU_QP()
            self.solver.setup(self.H, g, self.A, l, u, **self.solver_kw)
            self._ready = True
        else:
            self.solver.update(g=g, l=l, u=u)
        res = self.solver.solve()
        sol = res.x
        if hasattr(sol, "detach"):
            sol = sol.detach().cpu().double().numpy()
        sol = np.asarray(sol, dtype=np.float64)
        if sol.ndim == 1:
            sol = sol[None]
        return self.first_input(sol, x), res

    def _device_maps(self, device, dtype):
        """Constant maps of the closed loop on the device.  With u0 = v0 - K x the plant step is
        x+ = Ad x + Bd u0 = (Ad - Bd K) x + Bd v0: two GEMMs per step (Acl' and Bd' are what they multiply from the right)."""
        import torch
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device, dtype=dtype)
        if self.form == "sparse":
            # the reference's own form (RandomLinMPC.py:54-66): g does not depend on x0; l = u = -Ad x0 on the first dynamics
            # block; the first input is the first nu variables, the plant step x+ = Ad x + Bd u0
            n, m = self.H.shape[0], self.A.shape[0]
            lumap = np.zeros((m, self.nx))
            lumap[:self.nx] = -self.Ad
            return dict(gmap=t(np.zeros((n, self.nx))), lumap=t(lumap), ladd=t(self.l0), uadd=t(self.u0),
                        Aclt=t(self.Ad.T), Bdt=t(self.Bd.T))
        return dict(gmap=t(self.g_x0), lumap=t(self.lu_x0), ladd=t(self.l_add), uadd=t(self.u_add),
                    Aclt=t((self.Ad - self.Bd @ self.K).T), Bdt=t(self.Bd.T))

    def simulate_device(self, x0, steps, device, dtype):
        """Closed loop with every per-step map on the device (no host round trip per control step):
        g = G x, l/u = l_add/u_add + LU x (rqp_update_affine), warm solve, x+ = (Ad - Bd K) x + Bd v0 with v0 the first
        input block of the solution.  torch is used here for the caller-side plant step only; the QP path is the HIP library.
        Returns (final states [B, nx] tensor, mean ADMM iterations per solve)."""
        import torch
        mp = self._device_maps(device, dtype)
        x = torch.as_tensor(np.atleast_2d(x0), device=device, dtype=dtype)
        it_acc = None
        for k in range(steps):
            fresh = False
            if not self._ready:
                shift = x @ mp["lumap"].T
                g, l, u = x @ mp["gmap"].T, mp["ladd"] + shift, mp["uadd"] + shift
                import reluqp.reluqpth as reluqpth
                self.solver = self.solver or reluqpth.ReLU_QP()
                self.solver.setup(self.H, g, self.A, l, u, **self.solver_kw)
                self._ready = True
                fresh = True                         # setup() already holds this step's vectors
            sync = self.solver.synchronous
            self.solver.synchronous = False          # enqueue only: the steps chain on the stream, no host wait per step
            try:
                if not fresh:                        # g = G x, l/u = l_add/u_add + LU x in one device pass
                    self.solver.update_affine(x, mp["gmap"], mp["lumap"], mp["ladd"], mp["uadd"])
                res = self.solver.solve()
            finally:
                self.solver.synchronous = sync
            x = torch.addmm(x @ mp["Aclt"], res.x[:, :self.nu], mp["Bdt"])
            it_acc = res.info.iter.clone() if it_acc is None else it_acc.add_(res.info.iter)
        return x, float(it_acc.sum()) / (steps * x.shape[0])

    def simulate_graph(self, x0, steps, device, dtype):
        """simulate_device with the control step (x0 update, warm-started solve, plant step) captured ONCE in a HIP graph
        and replayed ``steps`` times: no host work per step at all (launch-bound small batches gain the most).
        The solver must be set up (run simulate_device for one step first).  Returns as simulate_device."""
        import torch
        assert self._ready
        mp = self._device_maps(device, dtype)
        x = torch.as_tensor(np.atleast_2d(x0), device=device, dtype=dtype).clone()   # static buffer: rewritten in place
        it_acc = torch.zeros(x.shape[0], device=device, dtype=torch.int32)
        sync = self.solver.synchronous
        self.solver.synchronous = False
        try:
            def step():
                self.solver.update_affine(x, mp["gmap"], mp["lumap"], mp["ladd"], mp["uadd"])
                res = self.solver.solve()
                x.copy_(torch.addmm(x @ mp["Aclt"], res.x[:, :self.nu], mp["Bdt"]))
                it_acc.add_(res.info.iter)
            side = torch.cuda.Stream(device=device)           # warm-up on a side stream, as torch's capture rules ask
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream(device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread-local capture error mode: frees / allocations on OTHER threads cannot invalidate the capture; on this
            # thread rqp_destroy (a finaliser, an explicit del) frees its workspace under the relaxed capture mode
            # (csrc/rqp_abi.hip: free_ws), so destroying a solver mid-capture is safe too (tests/test_mpc_gpu.py)
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                step()
            done = 1                                          # the warm-up ran one step (capturing only records)
            for _ in range(max(0, steps - done)):
                graph.replay()
        finally:
            self.solver.synchronous = sync
        torch.cuda.synchronize(device)
        return x, float(it_acc.sum()) / (max(steps, done) * x.shape[0])

    def simulate(self, x0, steps, noise=0.0, seed=0):
        """Closed loop x+ = Ad x + Bd u (+ noise); returns (states [steps+1, B, nx], inputs, iteration counts)."""
        rs = np.random.RandomState(seed)
        x = np.atleast_2d(np.array(x0, dtype=np.float64))
        xs, us, its = [x.copy()], [], []
        for _ in range(steps):
            u0, res = self.step(x)
            x = x @ self.Ad.T + u0 @ self.Bd.T + noise * rs.randn(*x.shape)
            xs.append(x.copy())
            us.append(u0)
            it = res.info.iter
            its.append(it.cpu().numpy() if hasattr(it, "cpu") else np.atleast_1d(np.asarray(it)))
        return np.stack(xs), np.stack(us), np.stack(its)
