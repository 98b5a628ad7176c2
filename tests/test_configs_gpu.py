"""BASELINE.json configurations at their full sizes on the GPU (round-1 VERDICT "configs_untested"), through the C-ABI.

  C3  batch=4096 linear-MPC QPs (N=20, nx=12, nu=4 -> n=80, m=320), shared (H, A): the MFMA kernel against the oracle
      on a subset and, on the whole batch, the KKT residuals (primal AND dual) re-derived independently on the device.
  C4  batch=8192 (the per-GPU share) and 65536, n=32, m=64: one-wavefront kernel; residuals re-derived on the device;
      instances 0..2 of the batch equal the reference's own outputs (tests/golden/g4_c2_feasible.npz, c4s*).
  C5  1000-step warm-started closed loop at the C3 shape against an oracle closed loop on the first instances.

Stated tolerances: float32 path, x/z within 1e-4 * max|x| of the float64 oracle where both stop at the same check,
iteration counts equal on >= 85 % of a subset (a check whose residual sits on the threshold may end one solve a check
earlier or later: both exits are valid).
"""
import numpy as np
import pytest
import torch

from oracle import reluqp_oracle as O
from reluqp import mpc, utils

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _kkt(H, A, g, x, z, y):
    """(primal, dual) inf-norm residuals per instance, float64 on the device; H, A shared [n,n]/[m,n] or batched."""
    x, z, y = x.double(), z.double(), y.double()
    if H.dim() == 2:
        Ax = x @ A.T
        dua = x @ H.T + y @ A + g
    else:
        Ax = torch.einsum("bmn,bn->bm", A, x)
        dua = torch.einsum("bij,bj->bi", H, x) + torch.einsum("bmn,bm->bn", A, y) + g
    return (Ax - z).abs().amax(1), dua.abs().amax(1)


def _c3(B, seed=5):
    Ad, Bd = mpc.random_plant(12, 4, seed=0)
    ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="condensed",
                        device=DEV, precision=torch.float32, eps_abs=1e-3)
    x0 = np.random.RandomState(seed).randn(B, 12)
    return ctl, x0


# ------------------------------------------------------------------------------------------ C3
def test_c3_batch4096_mfma_vs_oracle_and_kkt():
    import reluqp.reluqpth as reluqpth
    B, NS = 4096, 64
    ctl, x0 = _c3(B)
    g, l, u = ctl.qp_vectors(x0)
    assert ctl.H.shape == (80, 80) and ctl.A.shape == (320, 80)
    m = reluqpth.ReLU_QP()
    m.setup(ctl.H, g, ctl.A, l, u, device=DEV, precision=torch.float32, eps_abs=1e-3)
    assert m.kernel == "mfma"                                        # the default dispatch at this shape and batch
    res = m.solve()
    status = np.array(res.info.status)
    assert np.mean(status == "solved") == 1.0
    # (1) oracle on a 64-instance subset spread over the batch (tiles 0, 64, 128, ... and the last one)
    idx = np.unique(np.concatenate([np.arange(0, B, B // (NS - 8))[:NS - 8], np.arange(B - 8, B)]))
    ref = O.solve_batch(ctl.H, g[idx], ctl.A, l[idx], u[idx], form="factored", eps_abs=1e-3)
    it = res.info.iter.cpu().numpy()[idx]
    assert list(status[idx]) == ref["status"]
    assert np.mean(it == ref["iter"]) >= 0.85 and np.all(np.abs(it - ref["iter"]) <= 50)
    same = it == ref["iter"]
    scale = max(1.0, np.abs(ref["x"]).max())
    x = res.x.cpu().double().numpy()[idx]
    z = res.z.cpu().double().numpy()[idx]
    y = res.y.cpu().double().numpy()[idx]
    np.testing.assert_allclose(x[same], ref["x"][same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(z[same], ref["z"][same], rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(y[same], ref["lam"][same], rtol=0, atol=2e-3 * max(1.0, np.abs(ref["lam"]).max()))
    assert np.array_equal(res.info.rho_ind.cpu().numpy()[idx][same], ref["rho_ind"][same])
    # (2) whole batch: termination test on residuals recomputed outside the kernel, in float64
    Hd = torch.as_tensor(ctl.H, device=DEV)
    Ad_ = torch.as_tensor(ctl.A, device=DEV)
    gd = torch.as_tensor(g, device=DEV)
    pri, dua = _kkt(Hd, Ad_, gd, res.x, res.z, res.y)
    assert float(pri.max()) < 1e-3 * np.sqrt(320) * 1.02
    assert float(dua.max()) < 1e-3 * np.sqrt(80) * 1.05
    np.testing.assert_allclose(res.info.pri_res.cpu().double().numpy(), pri.cpu().numpy(), rtol=2e-2, atol=2e-5)
    np.testing.assert_allclose(res.info.dua_res.cpu().double().numpy(), dua.cpu().numpy(), rtol=5e-2, atol=2e-4)
    ld, ud = torch.as_tensor(l, device=DEV).float().double(), torch.as_tensor(u, device=DEV).float().double()
    zz = res.z.double()
    assert bool(((zz >= ld) & (zz <= ud)).all())


# ------------------------------------------------------------------------------------------ C4
@pytest.mark.parametrize("B", [8192, 65536])
def test_c4_full_batches_wave(golden, B):
    import reluqp.reluqpth as reluqpth
    n, n_eq, n_ineq = 32, 8, 56
    H, g, A, l, u, xs = utils.rand_qp_batch(B, n, n_eq, n_ineq, seed0=0, feasible=True, dtype=np.float32)
    m = reluqpth.ReLU_QP()
    Hd, gd, Ad, ld, ud = (torch.from_numpy(t).to(DEV) for t in (H, g, A, l, u))
    m.setup(Hd, gd, Ad, ld, ud, device=DEV, precision=torch.float32)
    assert m.kernel == "wave"
    res = m.solve()
    sc = res.info.status_code
    assert int((sc != 0).sum()) == 0                                  # every instance "solved"
    it = res.info.iter.cpu().numpy()
    assert it.min() >= 25 and it.max() <= 1500 and np.all(it % 25 == 0)
    pri, dua = _kkt(Hd.double(), Ad.double(), gd.double(), res.x, res.z, res.y)
    assert float(pri.max()) < 1e-3 * np.sqrt(64) * 1.02
    assert float(dua.max()) < 1e-3 * np.sqrt(32) * 1.05
    np.testing.assert_allclose(res.info.pri_res.cpu().double().numpy(), pri.cpu().numpy(), rtol=2e-2, atol=2e-5)
    zz = res.z.double()
    assert bool(((zz >= ld.double()) & (zz <= ud.double())).all())
    err = (res.x.double().cpu() - torch.from_numpy(xs).double()).abs().amax(1)
    assert float(err.max()) < 5e-2                                    # eps_abs = 1e-3 accuracy around the planted optimum
    # instances 0..2 are the reference's own C4-shape runs (golden c4s*): same exits inside the big batch
    gold = golden("g4_c2_feasible.npz")
    for s in range(3):
        p = "c4s%d_" % s
        assert int(it[s]) == int(gold[p + "iter"])
        np.testing.assert_allclose(res.x[s].cpu().double().numpy(), gold[p + "x"], rtol=0,
                                   atol=2e-5 * max(1.0, np.abs(gold[p + "state"]).max()))
        assert int(res.info.rho_ind[s]) == int(gold[p + "rho_ind_final"])


# ------------------------------------------------------------------------------------------ C5
def _oracle_closed_loop(ctl, x0, steps):
    """Closed loop of the oracle (factored form, float64): update(g, l, u) + warm-started solve() per step on each
    instance, plant step x+ = Ad x + Bd u0.  Returns (states [steps+1, NB, nx], iteration counts [steps, NB])."""
    NB = x0.shape[0]
    qps = []
    x = x0.copy()
    xs, its = [x.copy()], np.zeros((steps, NB), np.int64)
    for k in range(steps):
        g, l, u = ctl.qp_vectors(x)
        sol = np.zeros((NB, ctl.H.shape[0]))
        for b in range(NB):
            if k == 0:
                qp = O.OracleQP(form="factored")
                qp.setup(ctl.H, g[b], ctl.A, l[b], u[b], eps_abs=1e-3)
                qps.append(qp)
            else:
                qps[b].update(g=g[b], l=l[b], u=u[b])
            r = qps[b].solve()
            sol[b] = r.x
            its[k, b] = r.info.iter
        u0 = ctl.first_input(sol, x)
        x = x @ ctl.Ad.T + u0 @ ctl.Bd.T
        xs.append(x.copy())
    return np.stack(xs), its


@pytest.mark.parametrize("B,tile,kernel", [(4096, None, "mfma"), (64, None, "resident2"), (64, torch.float16, "resident2"),
                                           (4096, torch.float16, "mfma"), (4096, torch.bfloat16, "mfma16")])
def test_c5_closed_loop_1000_steps_vs_oracle(B, tile, kernel):
    """1000 control steps, update(g,l,u) from the current state + warm-started solve() per step (the path of
    reluqpth.py:159-183 + :201-249) at the C3 shape; instances 0..7 against the oracle closed loop.  Batch 4096 runs on
    the MFMA kernel, the small batch ("1 instance stream or small batch") on the register-resident tile.
    tile=float16: BASELINE config 5's "fp16 iterate / fp32 residual" mode -- the K(rho) tile in fp16 (it only
    preconditions dx = -K d), H, A, the state and every residual in float32/float64; at batch 4096 the MFMA
    kernel takes the same fp16-rounded K into its operand image, so the mode keeps the large-batch kernel.
    tile=bfloat16: the batch on the 16-bit matrix pipe (k_admm_mfma16: two bf16 planes per operand, tests/test_mfma16_gpu.py)."""
    import reluqp.reluqpth as reluqpth
    NB, STEPS = 8, 1000
    ctl, x0 = _c3(B, seed=11)
    x0 = 1.5 * x0
    g, l, u = ctl.qp_vectors(x0)
    m = reluqpth.ReLU_QP()
    m.setup(ctl.H, g, ctl.A, l, u, device=DEV, precision=torch.float32, eps_abs=1e-3, iterate_dtype=tile)
    assert m.kernel == kernel
    mp = ctl._device_maps(DEV, torch.float32)
    x = torch.as_tensor(x0, device=DEV, dtype=torch.float32)
    m.synchronous = False                                             # enqueue only; the steps chain on the stream
    its, traj, solved = [], [x[:NB].clone()], []
    for k in range(STEPS):
        if k > 0:
            m.update_affine(x, mp["gmap"], mp["lumap"], mp["ladd"], mp["uadd"])
        res = m.solve()
        x = torch.addmm(x @ mp["Aclt"], res.x[:, :ctl.nu], mp["Bdt"])
        its.append(res.info.iter[:NB].clone())
        solved.append((res.info.status_code == 0).sum())
        traj.append(x[:NB].clone())
    torch.cuda.synchronize()
    its = torch.stack(its).cpu().numpy()
    traj = torch.stack(traj).cpu().double().numpy()
    solved = torch.stack(solved).cpu().numpy()
    assert np.all(solved == B)                                        # every instance solved at every step
    xs_ref, its_ref = _oracle_closed_loop(ctl, x0[:NB], STEPS)
    # same per-step iteration counts over the first 50 steps (transient: constraints active, rho moves)
    agree = np.mean(its[:50] == its_ref[:50])
    assert agree >= (0.9 if tile is None else 0.8), agree
    assert np.mean(its == its_ref) >= (0.9 if tile is None else 0.8)
    # trajectories: the loop is contractive, eps_abs-level input differences do not accumulate
    scale = np.abs(xs_ref).max()
    assert np.abs(traj - xs_ref).max() < 2e-2 * scale
    assert np.abs(traj[-1] - xs_ref[-1]).max() < 5e-3 * scale
    assert np.linalg.norm(traj[-1], axis=1).max() < 0.05 * np.linalg.norm(traj[0], axis=1).max()   # regulated
    # steady state: warm starts stop at the first check
    assert np.median(its[-100:]) == 25
