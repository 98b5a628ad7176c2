#!/usr/bin/env python3
"""Random linear-MPC benchmark driver (BASELINE configs 3 and 5).

The reference names this driver (loose_code/RandomLinMPC.py) but ships no runnable version; this one
follows Appendix C.  Plant: nx=12, nu=4, horizon N=20, Q=I, R=0.1 I, Qf = Riccati P, box |u|<=u_max,
|x|<=x_max.  H and A are shared by the batch; (g, l, u) are per instance.

    python benchmarks/random_lin_mpc.py --batch 4096 --form condensed        # config 3: cold batched solve
    python benchmarks/random_lin_mpc.py --batch 256 --closed-loop 1000        # config 5: warm-started MPC sequence
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import reluqp.reluqpth as reluqpth  # noqa: E402
from reluqp import mpc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--form", choices=["condensed", "sparse"], default="condensed")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--nx", type=int, default=12)
    ap.add_argument("--nu", type=int, default=4)
    ap.add_argument("--u-max", type=float, default=0.5)
    ap.add_argument("--x-max", type=float, default=10.0)
    ap.add_argument("--x0-scale", type=float, default=1.0)
    ap.add_argument("--eps-abs", type=float, default=1e-3)
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--closed-loop", type=int, default=0, help="number of closed-loop control steps (0: cold solves)")
    ap.add_argument("--graph", action="store_true", help="closed loop: capture the control step in a HIP graph and replay it")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X"
    dev = torch.device("cuda", 0)
    prec = torch.float32 if args.precision == "f32" else torch.float64

    Ad, Bd = mpc.random_plant(args.nx, args.nu, seed=args.seed)
    Q, R = np.eye(args.nx), 0.1 * np.eye(args.nu)
    ctl = mpc.LinearMPC(Ad, Bd, Q, R, args.horizon, args.u_max, args.x_max, form=args.form,
                        device=dev, precision=prec, eps_abs=args.eps_abs)
    rs = np.random.RandomState(args.seed + 1)
    x0 = args.x0_scale * rs.randn(args.batch, args.nx)
    n, m = ctl.H.shape[0], ctl.A.shape[0]

    if args.closed_loop > 0:
        xw, _ = ctl.simulate_device(x0, 1, dev, prec)   # setup + first (cold) solve, untimed
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if args.graph:
            xf, mean_it = ctl.simulate_graph(xw.cpu().numpy(), args.closed_loop, dev, prec)
        else:
            xf, mean_it = ctl.simulate_device(xw.cpu().numpy(), args.closed_loop, dev, prec)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        out = {"bench": "closed-loop linear MPC (update + warm-started solve per step), maps on device" +
               (", control step replayed from a HIP graph" if args.graph else ""), "form": args.form,
               "batch": args.batch, "n": n, "m": m, "control_steps": args.closed_loop,
               "qp_solves_per_sec": args.batch * args.closed_loop / el, "ms_per_control_step": el / args.closed_loop * 1e3,
               "mean_iters_per_solve": mean_it, "final_state_norm_over_initial":
                   float(xf.double().norm(dim=1).mean().cpu() / np.linalg.norm(x0, axis=1).mean()),
               "kernel": ctl.solver.kernel, "dtype": args.precision}
    else:
        g, l, u = ctl.qp_vectors(x0)
        model = reluqpth.ReLU_QP()
        t0 = time.perf_counter()
        model.setup(ctl.H, g, ctl.A, l, u, device=dev, precision=prec, eps_abs=args.eps_abs, warm_starting=False)
        torch.cuda.synchronize(dev)
        setup_s = time.perf_counter() - t0
        model.solve()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = model.solve()
        torch.cuda.synchronize(dev)
        el = (time.perf_counter() - t0) / args.steps
        it = res.info.iter.double()
        out = {"bench": "batched cold linear-MPC solve, shared (H, A)", "form": args.form, "batch": args.batch,
               "n": n, "m": m, "qp_solves_per_sec": args.batch / el, "ms_per_batch": el * 1e3,
               "admm_iters_per_sec": float(it.sum()) / el, "mean_iters": float(it.mean()),
               "solved_frac": float((res.info.status_code == 0).double().mean()), "setup_s": setup_s,
               "kernel": model.kernel, "dtype": args.precision}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
