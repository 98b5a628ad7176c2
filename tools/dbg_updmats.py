import sys, os
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
from oracle import reluqp_oracle as O
from reluqp import utils
import reluqp.reluqpth as reluqpth
B, n, n_eq, n_ineq = 4, 100, 25, 275
H, g, A, l, u, _ = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=40, feasible=True)
rs = np.random.RandomState(1)
M = 0.1 * rs.randn(B, n, n)
H2 = H + np.einsum("bij,bkj->bik", M, M)
A2 = A + 0.05 * rs.randn(*A.shape)
def tight(Hb, Ab, b):
    qp = O.OracleQP(form="factored"); qp.setup(Hb, g[b], Ab, l[b], u[b], eps_abs=1e-8, max_iter=20000); return qp.solve().x
xs = {"HA": tight(H[0], A[0], 0), "H2A": tight(H2[0], A[0], 0), "H2A2": tight(H2[0], A2[0], 0), "HA2": tight(H[0], A2[0], 0)}
for kern in ("resident", "generic"):
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=torch.device("cuda:0"), precision=torch.float32, kernel=kern, eps_abs=1e-5, max_iter=20000)
    def show(tag):
        r = m.solve(); x = r.x[0].cpu().double().numpy()
        print(kern, tag, "it", r.info.iter.tolist(), {k: float(np.abs(x - v).max()) for k, v in xs.items()})
    show("setup      ")
    m.update(Hx=H2); show("Hx=H2      ")
    m.update(Ax=A2); show("Ax=A2      ")
    m.update(Hx=H, Ax=A); show("Hx=H,Ax=A  ")
