import sys, os
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
from oracle import reluqp_oracle as O
from reluqp import utils
import reluqp.reluqpth as reluqpth
B, n, n_eq, n_ineq = 6, 10, 3, 12
H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=40, feasible=True)
rs = np.random.RandomState(1)
M = 0.1 * rs.randn(B, n, n)
H2 = H + np.einsum("bij,bkj->bik", M, M)
for kern in ("auto", "generic"):
    m = reluqpth.ReLU_QP(); m.collect_trace = True
    m.setup(H, g, A, l, u, device=torch.device("cuda:0"), precision=torch.float64, kernel=kern)
    r0 = m.solve()
    print(kern, "r0 it", r0.info.iter.tolist(), "ri", r0.info.rho_ind.tolist())
    m.update(Hx=H2)
    r1 = m.solve()
    print(kern, "r1 it", r1.info.iter.tolist(), "ri", r1.info.rho_ind.tolist())
    print(" trace b=1", m.last_trace[1][:6].cpu().numpy())
    K = m.layers.K(7, 1).cpu().numpy()
    Kref = np.linalg.inv(H2[1] + 1e-6 * np.eye(n) + A[1].T @ (O.rho_vector(0.1, l[1], u[1], 1e-6)[:, None] * A[1]))
    print(" K err", np.abs(K - Kref).max(), np.abs(Kref).max())
for b in range(B):
    qp = O.OracleQP(form="factored"); qp.setup(H[b], g[b], A[b], l[b], u[b]); a0 = qp.solve(); i0, r0_ = a0.info.iter, qp.rho_ind
    qp.update(Hx=H2[b]); a1 = qp.solve()
    print("oracle b", b, "a0", i0, r0_, "a1", a1.info.iter, qp.rho_ind, qp.trace[:3] if b == 1 else "")
