#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation itself.

Run (in the build container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported unmodified from /root/reference/ReLU-QP-py and driven
through the external harness of SURVEY.md §8(c) (no reference file is edited or
copied):
  1. qp.start/qp.end CUDA events  -> perf_counter stand-ins (reluqpth.py:99-100
     raise "No HIP GPUs are available" on a GPU-less host),
  2. torch.cuda.synchronize       -> no-op,
  3. ReLU_Layer.forward           -> non-aliased statement of reluqpth.py:84-89
     (the shipped `matmul(W, input, out=input)` aliases and is wrong on CPU, Q1),
  4. device=cpu, precision=float64 (the only precision the reference runs, Q2),
  5. an empty `cvxpy` module so that reluqp.utils imports; compute_sol=False.

Fixtures are DATA ONLY: inputs and the reference's outputs.
"""
import hashlib
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/ReLU-QP-py"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules.setdefault("cvxpy", types.ModuleType("cvxpy"))

import reluqp.reluqpth as R  # noqa: E402  (the reference)
import reluqp.utils as RU  # noqa: E402  (the reference)

# our own generator (feasible variant); loaded by path so that it does not
# collide with the reference's `reluqp` package name
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "amd_utils", os.path.join(REPO, "reluqp-py_amd", "reluqp", "utils.py"))
amd_utils = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(amd_utils)


class _Ev:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


torch.cuda.Event = lambda **k: _Ev()
torch.cuda.synchronize = lambda *a, **k: None


def _forward(self, inp, idx):
    out = self.W_ks[idx] @ inp + self.b_ks[idx]
    i0, i1 = self.clamp_inds
    out[i0:i1] = torch.clamp(out[i0:i1], self.QP.l, self.QP.u)
    return out


R.ReLU_Layer.forward = _forward
CPU = torch.device("cpu")


def ref_setup(H, g, A, l, u, **kw):
    qp = R.ReLU_QP()
    qp.setup(H=H, g=g, A=A, l=l, u=u, device=CPU, **kw)
    return qp


def ref_solve_traced(qp):
    """solve() while recording every compute_residuals call (k = 25, 50, ...)."""
    trace = []
    orig = R.ReLU_QP.compute_residuals

    def rec(*a):
        out = orig(*a)
        trace.append((float(out[0]), float(out[1]), float(out[2]), int(qp.rho_ind)))
        return out

    qp.compute_residuals = rec
    res = qp.solve()
    del qp.compute_residuals
    return res, np.array(trace, dtype=np.float64).reshape(-1, 4)


def pack_result(prefix, qp, res, trace, out):
    out[prefix + "x"] = res.x.numpy().copy()
    out[prefix + "z"] = res.z.numpy().copy()
    out[prefix + "state"] = qp.output.numpy().copy()
    out[prefix + "iter"] = np.int64(res.info.iter)
    out[prefix + "status"] = np.array(res.info.status)
    out[prefix + "obj_val"] = np.float64(res.info.obj_val)
    out[prefix + "pri_res"] = np.float64(res.info.pri_res)
    out[prefix + "dua_res"] = np.float64(res.info.dua_res)
    out[prefix + "rho_estimate"] = np.float64(res.info.rho_estimate)
    out[prefix + "rho_ind_final"] = np.int64(qp.rho_ind)
    out[prefix + "trace"] = trace  # rows: pri, dua, rho_est, rho_ind before the move


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return np.array(h.hexdigest())


def builtin_qp():
    # the values of reluqpth.py:342-346 (the reference's only known-answer test)
    H = np.array([[6, 2, 1], [2, 5, 2], [1, 2, 4.0]])
    g = np.array([-8.0, -3, -3])
    A = np.array([[1, 0, 1], [0, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    l = np.array([3.0, 0, -10.0, -10, -10])
    u = np.array([3.0, 0, np.inf, np.inf, np.inf])
    return H, g, A, l, u


def g1():
    H, g, A, l, u = builtin_qp()
    out = dict(H=H, g=g, A=A, l=l, u=u)
    qp = ref_setup(H, g, A, l, u)
    out["rhos"] = qp.layers.rhos.numpy().copy()
    out["rho_ind0"] = np.int64(qp.rho_ind)
    out["W7"] = qp.layers.W_ks[7].numpy().copy()
    out["b7"] = qp.layers.b_ks[7].numpy().copy()
    s = qp.output.clone()
    states = {}
    for k in range(1, 26):
        s = qp.layers(s, qp.rho_ind)
        if k in (1, 2, 25):
            states[k] = s.numpy().copy()
    out["state_k1"], out["state_k2"], out["state_k25"] = states[1], states[2], states[25]
    res, trace = ref_solve_traced(qp)
    pack_result("", qp, res, trace, out)
    # second solve on the same object: warm-started (state + rho_ind persist)
    res2, trace2 = ref_solve_traced(qp)
    pack_result("warm_", qp, res2, trace2, out)
    # warm_starting=False: state cleared after the solve
    qp3 = ref_setup(H, g, A, l, u, warm_starting=False)
    res3, _ = ref_solve_traced(qp3)
    out["cold_state_after"] = qp3.output.numpy().copy()
    out["cold_rho_ind_after"] = np.int64(qp3.rho_ind)
    out["cold_iter"] = np.int64(res3.info.iter)
    # adaptive_rho=False ladder (1 element)
    qp4 = ref_setup(H, g, A, l, u, adaptive_rho=False)
    out["rhos_noadapt"] = qp4.layers.rhos.numpy().copy()
    # non-default ladder parameters
    qp5 = ref_setup(H, g, A, l, u, rho=0.4, rho_min=1e-3, rho_max=1e3, adaptive_rho_tolerance=3)
    out["rhos_alt"] = qp5.layers.rhos.numpy().copy()
    out["rho_ind0_alt"] = np.int64(qp5.rho_ind)
    # max_iters_reached path (max_iter not a multiple of check_interval, Q11)
    qp6 = ref_setup(H, g, A, l, u, max_iter=10)
    res6, trace6 = ref_solve_traced(qp6)
    out["mi10_status"] = np.array(res6.info.status)
    out["mi10_iter"] = np.int64(res6.info.iter)
    out["mi10_state"] = qp6.output.numpy().copy()
    # tight tolerance on the builtin QP
    qp7 = ref_setup(H, g, A, l, u, eps_abs=1e-8)
    res7, trace7 = ref_solve_traced(qp7)
    pack_result("tight_", qp7, res7, trace7, out)
    np.savez_compressed(os.path.join(HERE, "g1_builtin.npz"), **out)
    print("G1", res.x.numpy(), int(res.info.iter), res.info.status, "rho_ind", int(qp.rho_ind))


def g2():
    out = {}
    for seed in range(5):
        H, g, A, l, u, _ = RU.rand_qp(nx=10, n_eq=5, n_ineq=5, seed=seed, compute_sol=False)
        # our compat generator must reproduce the reference generator bit-for-bit
        H2, g2_, A2, l2, u2, _ = amd_utils.rand_qp(nx=10, n_eq=5, n_ineq=5, seed=seed, compute_sol=False)
        for a, b in ((H, H2), (g, g2_), (A, A2), (l, l2), (u, u2)):
            assert np.array_equal(a, b), "compat generator drifted from reference"
        p = "s%d_" % seed
        out.update({p + "H": H, p + "g": g, p + "A": A, p + "l": l, p + "u": u})
        qp = ref_setup(H, g, A, l, u)
        res, trace = ref_solve_traced(qp)
        pack_result(p, qp, res, trace, out)
        print("G2 seed", seed, int(res.info.iter), res.info.status, int(qp.rho_ind))
    np.savez_compressed(os.path.join(HERE, "g2_randqp_compat.npz"), **out)


def g3():
    out = {}
    for seed in range(3):
        H, g, A, l, u, xs = amd_utils.rand_qp(nx=10, n_eq=5, n_ineq=15, seed=seed, feasible=True)
        p = "s%d_" % seed
        out.update({p + "H": H, p + "g": g, p + "A": A, p + "l": l, p + "u": u, p + "x_planted": xs})
        qp = ref_setup(H, g, A, l, u)
        res, trace = ref_solve_traced(qp)
        pack_result(p, qp, res, trace, out)
        # tight run = "truth" for the planted optimum
        qpt = ref_setup(H, g, A, l, u, eps_abs=1e-9, max_iter=20000)
        rest, tracet = ref_solve_traced(qpt)
        pack_result(p + "tight_", qpt, rest, tracet, out)
        print("G3 seed", seed, int(res.info.iter), res.info.status, int(qp.rho_ind),
              "tight", int(rest.info.iter), rest.info.status,
              "err vs planted", float(np.abs(rest.x.numpy() - xs).max()))
    np.savez_compressed(os.path.join(HERE, "g3_c1_feasible.npz"), **out)


def g4():
    out = {}
    for seed in range(3):
        H, g, A, l, u, xs = amd_utils.rand_qp(nx=100, n_eq=25, n_ineq=275, seed=seed, feasible=True)
        p = "s%d_" % seed
        out[p + "input_sha256"] = sha(H, g, A, l, u)
        qp = ref_setup(H, g, A, l, u)
        res, trace = ref_solve_traced(qp)
        pack_result(p, qp, res, trace, out)
        print("G4 seed", seed, int(res.info.iter), res.info.status, int(qp.rho_ind))
        if seed < 2:
            qpt = ref_setup(H, g, A, l, u, eps_abs=1e-6)
            rest, tracet = ref_solve_traced(qpt)
            pack_result(p + "e6_", qpt, rest, tracet, out)
            print("   eps 1e-6:", int(rest.info.iter), rest.info.status, int(qpt.rho_ind))
    # C4-shaped instances (n=32, m=64)
    for seed in range(3):
        H, g, A, l, u, xs = amd_utils.rand_qp(nx=32, n_eq=8, n_ineq=56, seed=seed, feasible=True)
        p = "c4s%d_" % seed
        out[p + "input_sha256"] = sha(H, g, A, l, u)
        qp = ref_setup(H, g, A, l, u)
        res, trace = ref_solve_traced(qp)
        pack_result(p, qp, res, trace, out)
        print("G4/C4 seed", seed, int(res.info.iter), res.info.status, int(qp.rho_ind))
    np.savez_compressed(os.path.join(HERE, "g4_c2_feasible.npz"), **out)


def g5():
    H, g, A, l, u = builtin_qp()
    out = {}
    qp = ref_setup(H, g, A, l, u)
    qp.solve()
    g_new = np.array([-4.0, -6.0, 1.0])
    qp.update(g=g_new)
    res, trace = ref_solve_traced(qp)
    out["g_new"] = g_new
    pack_result("upd_g_", qp, res, trace, out)
    l_new = np.array([1.0, 0.0, -10.0, -10.0, -10.0])
    u_new = np.array([1.0, 0.0, np.inf, 0.5, np.inf])
    qp.update(l=l_new, u=u_new)
    res, trace = ref_solve_traced(qp)
    out["l_new"], out["u_new"] = l_new, u_new
    pack_result("upd_lu_", qp, res, trace, out)
    np.savez_compressed(os.path.join(HERE, "g5_update.npz"), **out)
    print("G5", out["upd_g_x"], int(out["upd_g_iter"]), out["upd_lu_x"], int(out["upd_lu_iter"]))


def g6():
    rs = np.random.RandomState(123)
    out = {}
    cases = []
    for c, (n, m) in enumerate([(3, 5), (10, 20), (32, 64), (7, 3)]):
        M = rs.randn(n, n)
        H = M.T @ M + np.eye(n)
        A = rs.randn(m, n)
        g = rs.randn(n)
        x, z, lam = rs.randn(n), rs.randn(m), rs.randn(m)
        cases.append((H, A, g, x, z, lam, 0.1 * (c + 1)))
    # 0/0 -> NaN case (Q17): everything zero
    n, m = 4, 6
    cases.append((np.eye(n), rs.randn(m, n), np.zeros(n), np.zeros(n), np.zeros(m), np.zeros(m), 0.1))
    # clamp-at-rho_max and clamp-at-rho_min cases
    H, A, g, x, z, lam, _ = cases[1]
    cases.append((H, A, -H @ x - A.T @ lam + 1e-9, x, np.zeros_like(z), lam, 1e5))   # num = 1, den ~ 1e-10 -> rho_max
    cases.append((H, A, g * 1e6, x, A @ x + 1e-12, lam, 1e-5))                       # num ~ 1e-12 -> rho_min
    for i, (H, A, g, x, z, lam, rho) in enumerate(cases):
        t = [torch.from_numpy(np.ascontiguousarray(a)) for a in (H, A, g, x, z, lam)]
        pri, dua, rho_new = R.ReLU_QP.compute_residuals(*t, torch.tensor(rho, dtype=torch.float64), 1e-6, 1e6)
        J = R.ReLU_QP.compute_J(t[0], t[2], t[3])
        p = "c%d_" % i
        out.update({p + "H": H, p + "A": A, p + "g": g, p + "x": x, p + "z": z, p + "lam": lam,
                    p + "rho_in": np.float64(rho), p + "pri": np.float64(pri), p + "dua": np.float64(dua),
                    p + "rho_out": np.float64(rho_new), p + "J": np.float64(J)})
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "g6_residuals.npz"), **out)
    print("G6", len(cases), "cases; NaN case rho_out =", float(out["c4_rho_out"]))


def g7():
    H, g, A, l, u, _ = RU.rand_qp(nx=10, n_eq=5, n_ineq=5, seed=0, compute_sol=False)
    qp = ref_setup(H, g, A, l, u)
    out = dict(H=H, g=g, A=A, l=l, u=u, rhos=qp.layers.rhos.numpy().copy())
    sigma = qp.settings.sigma
    n = 10
    for j in (3, 7, 12):
        W = qp.layers.W_ks[j].numpy().copy()
        B = qp.layers.B_ks[j].numpy().copy()
        out["W%d" % j] = W
        out["b%d" % j] = qp.layers.b_ks[j].numpy().copy()
        out["K%d" % j] = -B[:n].copy()  # B = [-K; -AK; 0]  (reluqpth.py:76)
    np.savez_compressed(os.path.join(HERE, "g7_matrices.npz"), **out)
    print("G7 ok; eq rows:", int(((u - l) <= qp.settings.eq_tol).sum()))


if __name__ == "__main__":
    torch.set_num_threads(1)  # deterministic reductions
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7"]
    for name in which:
        globals()[name]()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("total fixture bytes:", tot)
