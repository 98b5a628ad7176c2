#!/usr/bin/env python3
"""Random-QP benchmark harness: the measurement of the reference's
benchmarks/random_qps.py (random_initial_solve :47-81) on MI355X.

For nx in geomspace(nx_min, nx_max): n_seeds problems rand_qp(nx, nx/4, nx/4) -> setup + solve, status must be
"solved" (random_qps.py:23); mean/std of solve_time per size.  OSQP / ProxQP columns and the cross-solver
assertion (:68) are added only when those packages import (they are not installed in the build image).
Extension: --batch B solves B seeds of each size in ONE batched call (the MI355X-native way to use the solver).
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import reluqp.reluqpth as reluqpth  # noqa: E402
import reluqp.utils as utils  # noqa: E402

try:
    import osqp  # noqa: F401
    from scipy import sparse
    HAVE_OSQP = True
except ImportError:
    HAVE_OSQP = False


class Random_QP_benchmark():
    def __init__(self, precision=torch.float64, feasible=False):
        self.precision = precision
        self.feasible = feasible

    def reluqpth_solve(self, nx=10, n_eq=5, n_ineq=5, seed=1, tol=1e-4):
        H, g, A, l, u, x_sol = utils.rand_qp(nx=nx, n_eq=n_eq, n_ineq=n_ineq, seed=seed, compute_sol=False,
                                             feasible=self.feasible)
        model = reluqpth.ReLU_QP()
        model.setup(H=H, g=g, A=A, l=l, u=u, eps_abs=tol, precision=self.precision)
        results = model.solve()
        assert results.info.status == 'solved'
        return results.info.solve_time, results.x

    def osqp_solve(self, nx=10, n_eq=5, n_ineq=5, seed=1, tol=1e-4):
        H, g, A, l, u, x_sol = utils.rand_qp(nx=nx, n_eq=n_eq, n_ineq=n_ineq, seed=seed, compute_sol=False,
                                             feasible=self.feasible)
        model = osqp.OSQP()
        model.setup(P=sparse.csc_matrix(H), q=g, A=sparse.csc_matrix(A), l=l, u=u, eps_abs=tol, eps_rel=0, verbose=False)
        results = model.solve()
        return results.info.solve_time, results.x

    def random_initial_solve(self, nx_min=10, nx_max=100, n_sample=6, n_seeds=5, tol=1e-4, check=None):
        """reference random_qps.py:47-81.  ``check(nx, seed, tol, x)``: optional cross-solver hook called per solve (the
        reference compares with OSQP at :68; OSQP is used when importable, tests pass an independent checker)."""
        nx_list = np.geomspace(nx_min, nx_max, num=n_sample)
        rows = []
        for _ in range(3):                                   # warm the runtime (reference: "make sure reluqp is compiled")
            self.reluqpth_solve()
        for nx in nx_list:
            nx = int(nx)
            times, otimes = [], []
            for seed in range(n_seeds):
                t, x = self.reluqpth_solve(nx=nx, n_eq=nx // 4, n_ineq=nx // 4, seed=seed, tol=tol)
                times.append(t)
                if check is not None:
                    check(nx, seed, tol, x)
                if HAVE_OSQP:
                    to, xo = self.osqp_solve(nx=nx, n_eq=nx // 4, n_ineq=nx // 4, seed=seed, tol=tol)
                    otimes.append(to)
                    assert np.linalg.norm(x.cpu().numpy() - xo, ord=np.inf) < 10 * tol     # random_qps.py:68
            row = {"nx": nx, "reluqpth_mean_s": float(np.mean(times)), "reluqpth_std_s": float(np.std(times))}
            if otimes:
                row.update({"osqp_mean_s": float(np.mean(otimes)), "osqp_std_s": float(np.std(otimes))})
            rows.append(row)
            print(json.dumps(row))
        return rows

    def batched_solve(self, nx, batch, tol=1e-4):
        H, g, A, l, u, _ = utils.rand_qp_batch(batch, nx, nx // 4, nx // 4, seed0=0, feasible=self.feasible)
        model = reluqpth.ReLU_QP()
        model.setup(H, g, A, l, u, eps_abs=tol, precision=self.precision)
        res = model.solve()
        return {"nx": nx, "batch": batch, "solve_time_s": res.info.solve_time, "setup_time_s": res.info.setup_time,
                "solved_frac": float(np.mean([s == "solved" for s in res.info.status])),
                "qp_per_s": batch / res.info.solve_time, "kernel": model.kernel}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx-min", type=int, default=10)
    ap.add_argument("--nx-max", type=int, default=100)
    ap.add_argument("--n-sample", type=int, default=6)
    ap.add_argument("--n-seeds", type=int, default=5)
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--precision", choices=["f32", "f64"], default="f64")
    a = ap.parse_args()
    bench = Random_QP_benchmark(precision=torch.float32 if a.precision == "f32" else torch.float64)
    if a.batch:
        for nx in np.geomspace(a.nx_min, a.nx_max, num=a.n_sample):
            print(json.dumps(bench.batched_solve(int(nx), a.batch, a.tol)))
    else:
        bench.random_initial_solve(a.nx_min, a.nx_max, a.n_sample, a.n_seeds, a.tol)
