"""Diagnostic: one case of tests/test_fuzz_gpu.py::test_shared_matrix_batches_all_kernels_vs_oracle on every float32 kernel."""
import sys, os
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd")); sys.path.insert(0, R0)
import numpy as np, torch
from oracle import reluqp_oracle as O
from reluqp import utils
import reluqp.reluqpth as reluqpth
sys.path.insert(0, os.path.join(R0, "tests"))
import test_fuzz_gpu as T
case = [c for c in T._shared_cases() if c.id == sys.argv[1]][0]
n, n_eq, n_ineq, B, seed0, st = case.values
print(case.id, st)
H, g0, A, l0, u0, _ = utils.rand_qp(n, n_eq, n_ineq, seed=seed0, compute_sol=False, feasible=True)
upd = [utils.update_qp(H, A, n_eq, n_ineq, seed=seed0 + 1 + b, compute_sol=False, feasible=True) for b in range(B)]
g = np.stack([x[1] for x in upd]); l = np.stack([x[3] for x in upd]); u = np.stack([x[4] for x in upd])
ref = O.solve_batch(H, g, A, l, u, form="factored", **st)
print("oracle f64 ", ref["iter"])
i = int(sys.argv[2]) if len(sys.argv) > 2 else 1
qp = O.OracleQP(form="factored"); qp.setup(H, g[i], A, l[i], u[i], **st); r = qp.solve()
print("oracle trace inst", i, [(round(t[0], 6), round(t[1], 6), t[3]) for t in qp.trace][:25])
thr_p = st["eps_abs"] * np.sqrt(n_eq + n_ineq); thr_d = st["eps_abs"] * np.sqrt(n)
print("thresholds", thr_p, thr_d, "eps_rel", st["eps_rel"])
for kernel in ("mfma", "resident", "wave", "generic"):
    try:
        m = reluqpth.ReLU_QP(); m.collect_trace = True
        m.setup(H, g, A, l, u, device=torch.device("cuda:0"), precision=torch.float32, kernel=kernel, **st)
        rr = m.solve()
        print("%-9s" % m.kernel, rr.info.iter.cpu().numpy())
        if m.last_trace is not None:
            tr = m.last_trace[i].cpu().numpy()
            print("   trace", [(round(float(t[0]), 6), round(float(t[1]), 6), int(t[3])) for t in tr[:25] if not np.isnan(t[3])])
    except Exception as e:
        print(kernel, "->", type(e).__name__, str(e)[:100])
