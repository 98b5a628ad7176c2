#!/usr/bin/env python3
"""Smallest use of the solver, the call sequence of the reference's examples/reluqpth-simple.py (random QP, setup, solve,
print) -- plus the two things the MI355X build adds on the same API: a batch in one call, and a warm-started update."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import reluqp.reluqpth as reluqpth  # noqa: E402
import reluqp.utils as utils  # noqa: E402

# one QP, float64, reference defaults -------------------------------------------------------------
H, g, A, l, u, x_sol = utils.rand_qp(nx=10, n_eq=5, n_ineq=5, seed=1, compute_sol=False)
model = reluqpth.ReLU_QP()
model.setup(H=H, g=g, A=A, l=l, u=u)
results = model.solve()
print("single:", results.info.status, "iter", results.info.iter, "solve_time %.3f ms" % (1e3 * results.info.solve_time))
print("  x =", np.round(results.x.cpu().numpy(), 4))

# new linear term, warm-started re-solve (reluqpth.py:159-183) --------------------------------------
model.update(g=0.9 * g)
results = model.solve()
print("update(g): iter", results.info.iter, results.info.status)

# a batch of 1024 independent QPs in ONE call, float32 ----------------------------------------------
Hb, gb, Ab, lb, ub, _ = utils.rand_qp_batch(1024, 100, 25, 275, seed0=0, feasible=True, dtype=np.float32)
batch = reluqpth.ReLU_QP()
batch.setup(Hb, gb, Ab, lb, ub, precision=torch.float32)
res = batch.solve()
print("batch of 1024 (n=100, m=300): kernel=%s, %.2f ms, %d solved, mean iter %.1f"
      % (batch.kernel, 1e3 * res.info.run_time, int((res.info.status_code == 0).sum()), float(res.info.iter.float().mean())))
