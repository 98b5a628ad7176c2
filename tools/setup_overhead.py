"""Host-side cost of setup() on device-resident inputs (not product): cProfile of 5 setups of the headline batch."""
import cProfile, pstats, sys, os, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, g, A, l, u, _ = utils.rand_qp_batch(B, 100, 25, 275, seed0=0, feasible=True, dtype=np.float32)
dev = torch.device("cuda:0")
Hd, gd, Ad, ld, ud = (torch.as_tensor(t, device=dev) for t in (H, g, A, l, u))
m = reluqpth.ReLU_QP()
for _ in range(2):
    m.setup(Hd, gd, Ad, ld, ud, device=dev, precision=torch.float32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m.setup(Hd, gd, Ad, ld, ud, device=dev, precision=torch.float32)
torch.cuda.synchronize()
print("setup wall: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), "info.setup_time %.2f ms" % (m.results.info.setup_time * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    m.setup(Hd, gd, Ad, ld, ud, device=dev, precision=torch.float32)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
