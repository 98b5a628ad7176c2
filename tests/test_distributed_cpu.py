"""N>1 host path on the CPU: world_size-2 gloo run of the batch-split logic that bench.py and
reluqp.distributed use on the 8-GPU node (contiguous batch split, no data-path
collective; only the reported scalars are reduced)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import reluqp_oracle as O
from reluqp import distributed as D
from reluqp import utils

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 4096, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            if world > total:
                continue
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (s0, n0), (s1, n1) in zip(spans, spans[1:]):
                assert s0 + n0 == s1
            assert spans[-1][0] + spans[-1][1] == total
            sizes = [n for _, n in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(8, 8, 8)
    assert D.shard_range(65536, 3, 8) == (3 * 8192, 8192)          # BASELINE config 4: 8192 per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
def test_two_rank_gloo_batch_split(tmp_path):
    total = 12
    out = tmp_path / "dist.json"
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "tests", "_dist_worker.py"), str(total), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out))
    # unsharded reference: the union of the shards must reproduce it exactly (independent instances)
    H, g, A, l, u, xs = utils.rand_qp_batch(total, 10, 5, 15, seed0=0, feasible=True)
    ref = O.solve_batch(H, g, A, l, u, form="factored")
    assert got["world"] == 2
    assert got["elapsed_max"] == 0.5 and got["extra_max"] == [1.0, -0.25]    # MAX over ranks; the second entry is -min(elapsed) (bench.py per_gpu)
    assert got["total_qps"] == total and got["total_solved"] == total        # SUM over ranks
    assert got["total_iters"] == float(ref["iter"].sum())
    assert np.array_equal(np.array(got["iter"]), ref["iter"])
    np.testing.assert_array_equal(np.array(got["x"]), ref["x"])              # same seeds -> bit-identical
