"""GPU tests of the dispatch order of the per-instance kernels (`SolveArgs.order`, k_order_lpt): the path bench.py's
`value_with_history` times, and the grid-order path its headline `value` times.

Instances are independent (reference reluqpth.py:201-249: one QP per object), so the order in which workgroups are issued
must not change a single bit of any result.  Every solve here writes into outputs pre-filled with NaN / -7
(`prefill_outputs`): an instance a bad permutation skipped cannot pass with the values a previous solve left in the
caching allocator's block.

  * headline batch (B=4096, n=100, m=300, float32, k_admm_res2): solve #1 runs in grid order, solve #2 longest-first,
    solve #3 with the history switched off; all three bit-identical in x, z, lam, iter, status, rho index, residuals,
    rho estimate, objective; instances 0-2 equal the reference's own outputs (golden G4 s0-s2); a 64-instance subset
    equals the oracle; KKT residuals of every instance recomputed in float64 on the device.
  * the recorded order is a permutation of the batch, sorted by the recorded iteration counts, descending; the counts
    are the ones the solve returned.
  * same bit-identity for k_admm_wave (B=8192, n=32, m=64) and k_admm_res64 (B=1024, float64).
"""
import numpy as np
import pytest
import torch

from oracle import reluqp_oracle as O
from reluqp import utils

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _setup(B, n, n_eq, n_ineq, prec, **kw):
    import reluqp.reluqpth as reluqpth
    dt = np.float32 if prec == torch.float32 else np.float64
    H, g, A, l, u, xs = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=0, feasible=True, dtype=dt)
    m = reluqpth.ReLU_QP()
    m.prefill_outputs = True
    m.setup(H, g, A, l, u, device=DEV, precision=prec, warm_starting=False, **kw)
    return m, (H, g, A, l, u, xs)


def _snapshot(res):
    i = res.info
    return dict(x=res.x.clone(), z=res.z.clone(), y=res.y.clone(), iter=i.iter.clone(), status=i.status_code.clone(),
                rho_ind=i.rho_ind.clone(), pri=i.pri_res.clone(), dua=i.dua_res.clone(), rho=i.rho_estimate.clone(),
                obj=i.obj_val.clone())


def _assert_written(s, B):
    for k in ("x", "z", "y", "pri", "dua", "rho", "obj"):
        assert not bool(torch.isnan(s[k]).any()), "%s: an instance was not written" % k
    for k in ("iter", "status", "rho_ind"):
        assert bool((s[k] != -7).all()), "%s: an instance was not written" % k
    assert s["iter"].shape[0] == B


def _assert_identical(a, b, what):
    for k in a:
        assert torch.equal(a[k], b[k]), "%s: %s differs between the two dispatch orders" % (what, k)


def _check_order(m, snap, B):
    d = m.get_dispatch()
    assert d is not None, "no dispatch order recorded after a solve of a batch of >= 4 workgroups per CU"
    order, last = d[0].cpu().numpy(), d[1].cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(B)), "order is not a permutation of the batch"
    assert np.array_equal(last, snap["iter"].cpu().numpy()), "ranked by other counts than the solve returned"
    ranked = last[order]
    assert np.all(ranked[:-1] >= ranked[1:]), "not sorted longest-first"


def test_headline_batch_ordered_launch_bit_identical(golden):
    B, n, m_ = 4096, 100, 300
    m, (H, g, A, l, u, xs) = _setup(B, n, 25, 275, torch.float32)
    assert m.kernel == "resident2"
    assert m.get_dispatch() is None                                   # fresh handle: nothing to rank by
    s1 = _snapshot(m.solve())                                         # grid order
    _assert_written(s1, B)
    _check_order(m, s1, B)
    s2 = _snapshot(m.solve())                                         # longest-first (the order just checked)
    _assert_written(s2, B)
    _assert_identical(s1, s2, "grid order vs longest-first")
    _check_order(m, s2, B)
    m.dispatch_history(False)                                         # what bench.py's headline `value` runs
    assert m.get_dispatch() is None
    s3 = _snapshot(m.solve())
    _assert_written(s3, B)
    _assert_identical(s1, s3, "history off")
    assert m.get_dispatch() is None
    m.dispatch_history(True)
    s4 = _snapshot(m.solve())                                         # back on: first launch after the switch is grid order again
    _assert_identical(s1, s4, "history back on")
    _check_order(m, s4, B)

    it = s1["iter"].cpu().numpy()
    assert bool((s1["status"] == 0).all())
    assert it.min() >= 50 and it.max() <= 600 and np.all(it % 25 == 0)
    # the reference's own outputs for instances 0..2 (seeds 0..2 of the same generator)
    gold = golden("g4_c2_feasible.npz")
    for s in range(3):
        assert int(it[s]) == int(gold["s%d_iter" % s])
        assert int(s1["rho_ind"][s]) == int(gold["s%d_rho_ind_final" % s])
        np.testing.assert_allclose(s1["x"][s].cpu().double().numpy(), gold["s%d_x" % s], rtol=0,
                                   atol=2e-5 * np.abs(gold["s%d_x" % s]).max())
    # oracle (float64, factored form) on 64 instances spread over the batch; float32 tolerance of test_hip_parity
    idx = np.unique(np.concatenate([np.arange(0, B, B // 56)[:56], np.arange(B - 8, B)]))
    ref = O.solve_batch(H[idx].astype(np.float64), g[idx].astype(np.float64), A[idx].astype(np.float64),
                        l[idx].astype(np.float64), u[idx].astype(np.float64), form="factored")
    same = it[idx] == ref["iter"]
    assert same.mean() >= 0.9 and np.all(np.abs(it[idx] - ref["iter"]) <= 25)
    xg = s1["x"].cpu().double().numpy()[idx]
    np.testing.assert_allclose(xg[same], ref["x"][same], rtol=0, atol=5e-5 * np.abs(ref["x"]).max())
    # KKT residuals of EVERY instance, float64 on the device, against the thresholds the kernel tested (reluqpth.py:233)
    Hd, Ad, gd = (torch.from_numpy(t).to(DEV).double() for t in (H, A, g))
    x, z, y = s1["x"].double(), s1["z"].double(), s1["y"].double()
    pri = (torch.einsum("bmn,bn->bm", Ad, x) - z).abs().amax(1)
    dua = (torch.einsum("bij,bj->bi", Hd, x) + torch.einsum("bmn,bm->bn", Ad, y) + gd).abs().amax(1)
    assert float(pri.max()) < 1e-3 * np.sqrt(m_) * 1.01 and float(dua.max()) < 1e-3 * np.sqrt(n) * 1.05
    err = (x - torch.from_numpy(xs).to(DEV).double()).abs().amax(1)
    assert float(err.max()) < 2e-2                                    # planted optimum at eps_abs = 1e-3


@pytest.mark.parametrize("B,n,n_eq,n_ineq,prec,kernel", [(8192, 32, 8, 56, torch.float32, "wave"),
                                                         (1024, 100, 25, 275, torch.float64, "resident64")])
def test_wave_and_res64_ordered_launch_bit_identical(B, n, n_eq, n_ineq, prec, kernel):
    m, _ = _setup(B, n, n_eq, n_ineq, prec)
    assert m.kernel == kernel
    s1 = _snapshot(m.solve())
    _assert_written(s1, B)
    assert bool((s1["status"] == 0).all())
    _check_order(m, s1, B)
    s2 = _snapshot(m.solve())
    _assert_written(s2, B)
    _assert_identical(s1, s2, "%s: grid order vs longest-first" % kernel)
    m.dispatch_history(False)
    s3 = _snapshot(m.solve())
    _assert_written(s3, B)
    _assert_identical(s1, s3, "%s: history off" % kernel)


def test_small_batches_record_no_order():
    """Below 4 workgroups per CU (2 for the float64 tile) a launch is one wave of workgroups: nothing to order."""
    m, _ = _setup(64, 20, 5, 15, torch.float32)
    m.solve()
    assert m.get_dispatch() is None
