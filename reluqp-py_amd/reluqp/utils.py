"""Problem generators for the ReLU-QP hot path (inputs only, numpy).

Mirrors the reference's ``reluqp.utils`` module surface (``rand_qp``,
``update_qp``; reference ``ReLU-QP-py/reluqp/utils.py:11-70``) so that
``import reluqp.utils as utils`` keeps working, and adds

* ``feasible=True``: the benchmark generator of SURVEY.md §8(d).  The reference
  draws a *signed* slack (``utils.py:27``) and a signed inequality multiplier
  (``utils.py:24,29``), which makes its problems infeasible / not optimal at the
  planted point once ``n_eq + n_ineq`` approaches ``nx``.  The
  feasible variant keeps the *same draw order* but uses ``|slack|`` and
  ``|lamb|`` with the sign of ``C'lamb`` fixed, so the planted ``x`` is the
  exact optimum (KKT holds by construction).
* ``rand_qp_batch``: stacked instances, instance ``i`` seeded ``seed0 + i``.

cvxpy is optional here (it is not installed in the build image):
``compute_sol=True`` returns the planted optimum in feasible mode and ``None`` in
reference-compat mode when cvxpy is missing.
"""
import numpy as np


def _draw(nx, n_eq, n_ineq, rs, H=None, A=None, C=None):
    """Common draw sequence (reference utils.py:13-27 / :46-56)."""
    if H is None:
        M = rs.randn(nx, nx)
        H = M.T @ M + np.eye(nx)
        H = H + H.T
        A = rs.randn(n_eq, nx)
        C = rs.randn(n_ineq, nx)
    active = rs.randn(n_ineq) > 0.5
    mu = rs.randn(n_eq)
    lamb = rs.randn(n_ineq) * active
    x = rs.randn(nx)
    b = A @ x
    slack = rs.randn(n_ineq)
    return H, A, C, active, mu, lamb, x, b, slack


def _assemble(H, A, C, active, mu, lamb, x, b, slack, feasible):
    n_ineq = C.shape[0]
    if feasible:
        d = C @ x - np.abs(slack) * (~active)
        g = -H @ x - A.T @ mu + C.T @ np.abs(lamb)
    else:  # reference verbatim (utils.py:27,29)
        d = C @ x - slack * (~active)
        g = -H @ x - A.T @ mu - C.T @ lamb
    return (H, g, np.vstack((A, C)), np.concatenate((b, d)),
            np.concatenate((b, np.full(n_ineq, np.inf))))


def _maybe_solution(H, g, Aeq, b, C, d, x_planted, compute_sol, feasible):
    if not compute_sol:
        return None
    if feasible:
        return x_planted
    try:  # reference behaviour (utils.py:31-34) when cvxpy exists
        import cvxpy as cp
    except ImportError:
        return None
    xv = cp.Variable(H.shape[0])
    prob = cp.Problem(cp.Minimize(0.5 * cp.quad_form(xv, np.array(H)) + g.T @ xv),
                      [Aeq @ xv == b, C @ xv >= d])
    prob.solve()
    return xv.value


def rand_qp(nx=10, n_eq=5, n_ineq=5, seed=1, compute_sol=True, feasible=False):
    """Random dense QP ``min 1/2 x'Hx + g'x  s.t. l <= Ax <= u``.

    Same signature, draw order and return tuple ``(H, g, A, l, u, x_sol)`` as
    the reference (utils.py:11-39).  ``np.random.RandomState(seed)`` yields the
    same stream as the reference's global ``np.random.seed(seed)``.
    """
    rs = np.random.RandomState(seed)
    H, A, C, active, mu, lamb, x, b, slack = _draw(nx, n_eq, n_ineq, rs)
    H, g, Afull, l, u = _assemble(H, A, C, active, mu, lamb, x, b, slack, feasible)
    x_sol = _maybe_solution(H, g, A, b, C, l[n_eq:], x, compute_sol, feasible)
    return H, g, Afull, l, u, x_sol


def update_qp(H, A, n_eq, n_ineq, seed=1, compute_sol=True, feasible=False):
    """New (g, l, u) for the same (H, A) (reference utils.py:42-70)."""
    rs = np.random.RandomState(seed)
    nx = H.shape[0]
    C = A[n_eq:]
    Aeq = A[:n_eq]
    H, Aeq, C, active, mu, lamb, x, b, slack = _draw(nx, n_eq, n_ineq, rs, H=H, A=Aeq, C=C)
    H, g, Afull, l, u = _assemble(H, Aeq, C, active, mu, lamb, x, b, slack, feasible)
    x_sol = _maybe_solution(H, g, Aeq, b, C, l[n_eq:], x, compute_sol, feasible)
    return H, g, Afull, l, u, x_sol


def _shared_empty(shape, dtype):
    """numpy array on an anonymous shared mapping: children forked afterwards write into the parent's memory."""
    import mmap
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = mmap.mmap(-1, max(nbytes, 1))
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def _fill_chunk(out, a, b, nx, n_eq, n_ineq, seed0, feasible):
    try:                                      # one BLAS thread per child: the processes are the parallelism
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except ImportError:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        for i in range(a, b):
            Hi, gi, Ai, li, ui, xi = rand_qp(nx, n_eq, n_ineq, seed=seed0 + i, compute_sol=feasible, feasible=feasible)
            out[0][i], out[1][i], out[2][i], out[3][i], out[4][i] = Hi, gi, Ai, li, ui
            if xi is not None:
                out[5][i] = xi


def rand_qp_batch(batch, nx, n_eq, n_ineq, seed0=0, feasible=True, dtype=np.float64, workers=0):
    """Stack ``batch`` independent instances; instance i uses seed ``seed0 + i``.

    Returns ``H[B,n,n], g[B,n], A[B,m,n], l[B,m], u[B,m], x_sol[B,n]`` (x_sol is
    the planted optimum in feasible mode, NaN otherwise).  ``workers > 1`` draws
    contiguous chunks in that many forked child processes writing into shared
    memory (same instances, same seeds; call it before the process initialises a GPU).
    """
    m = n_eq + n_ineq
    if workers and workers > 1 and batch >= 4 * workers:
        import multiprocessing as mp
        shapes = ((batch, nx, nx), (batch, nx), (batch, m, nx), (batch, m), (batch, m), (batch, nx))
        out = [_shared_empty(sh, dtype) for sh in shapes]
        out[5][:] = np.nan
        edges = np.linspace(0, batch, workers + 1).astype(int)
        ctx = mp.get_context("fork")
        ps = [ctx.Process(target=_fill_chunk, args=(out, int(a), int(b), nx, n_eq, n_ineq, seed0, feasible))
              for a, b in zip(edges[:-1], edges[1:])]
        for p in ps:
            p.start()
        for p in ps:
            p.join()
            if p.exitcode != 0:
                raise RuntimeError("rand_qp_batch worker failed (exit code %s)" % p.exitcode)
        return tuple(out)
    m = n_eq + n_ineq
    H = np.empty((batch, nx, nx), dtype)
    g = np.empty((batch, nx), dtype)
    A = np.empty((batch, m, nx), dtype)
    l = np.empty((batch, m), dtype)
    u = np.empty((batch, m), dtype)
    xs = np.full((batch, nx), np.nan, dtype)
    for i in range(batch):
        Hi, gi, Ai, li, ui, xi = rand_qp(nx, n_eq, n_ineq, seed=seed0 + i,
                                         compute_sol=feasible, feasible=feasible)
        H[i], g[i], A[i], l[i], u[i] = Hi, gi, Ai, li, ui
        if xi is not None:
            xs[i] = xi
    return H, g, A, l, u, xs
