"""Host-side cost of one synchronous solve() call (not product): cProfile of 2000 warm-started solves of a tiny batch."""
import cProfile, pstats, sys, os, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
H, g, A, l, u, _ = utils.rand_qp_batch(64, 32, 8, 56, seed0=0, feasible=True, dtype=np.float32)
m = reluqpth.ReLU_QP()
m.setup(H, g, A, l, u, device=torch.device("cuda:0"), precision=torch.float32, warm_starting=True)
for _ in range(50):
    m.solve()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 2000
for _ in range(N):
    m.solve()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("per solve: %.1f us wall, kernel %.1f us" % (dt * 1e6, m.last_kernel_time * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    m.solve()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
