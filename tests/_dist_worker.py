"""Worker of tests/test_distributed_cpu.py: launched by torch.distributed.run with the gloo
backend, world_size 2, on the CPU.  It exercises the N>1 host path of bench.py -- shard the
batch by rank, solve the shard, reduce the reported scalars -- with the ORACLE standing in as
the per-rank solver (tests may use the oracle; the product path needs a GPU)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "reluqp-py_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import reluqp_oracle as O  # noqa: E402
from reluqp import distributed as D  # noqa: E402
from reluqp import utils  # noqa: E402


def main():
    total, n, n_eq, n_ineq = int(sys.argv[1]), 10, 5, 15
    out_path = sys.argv[2]
    rank, world, local_rank, dist = D.init(backend="gloo")
    start, size = D.shard_range(total, rank, world)
    H, g, A, l, u, xs = utils.rand_qp_batch(size, n, n_eq, n_ineq, seed0=start, feasible=True)
    if dist is not None:
        dist.barrier()
    ref = O.solve_batch(H, g, A, l, u, form="factored")
    elapsed = 0.25 * (rank + 1)                       # synthetic per-rank time: the report takes the MAX
    cpu = torch.device("cpu")
    el, extra, tot_it, tot_solved, tot_q = D.reduce_report(
        dist, cpu, elapsed, ref["iter"].sum(), sum(s == "solved" for s in ref["status"]), size, extra_max=[float(rank), -elapsed])      # (a minimum rides along as the maximum of the negated value: bench.py's per_gpu)
    xs_all = D.gather_shards(dist, torch.from_numpy(ref["x"]), total, rank, world)
    it_all = D.gather_shards(dist, torch.from_numpy(ref["iter"]), total, rank, world)
    if rank == 0:
        json.dump({"world": world, "elapsed_max": el, "extra_max": extra, "total_iters": tot_it,
                   "total_solved": tot_solved, "total_qps": tot_q, "x": xs_all.numpy().tolist(),
                   "iter": it_all.numpy().tolist()}, open(out_path, "w"))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
