#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the MI355X ReLU-QP hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batched cold-start solve() of the per-GPU batch (default: 4096 random
dense QPs, n=100, m=300, float32, feasible rand_qp generator of SURVEY.md 8(d), eps_abs
1e-3, all reference defaults), inputs and the K(rho) table already resident in HBM.
Instances are independent: each rank owns its own shard (weak scaling: 4096 per GPU),
there is NO data-path collective; torch.distributed is used only for the barrier around
the timed region and for the max-over-ranks / sum reductions of the reported numbers.

Rank 0 prints ONE JSON line.  `value` = all QPs solved by all ranks / max-over-ranks time.
`roofline` prices the ADMM kernel against the HBM roof using the ALGORITHMIC bytes of
SURVEY.md 8(d): 4*(n^2 + m*n) bytes per instance-iteration (K and A streamed once); the
resident design may exceed 1.0 of that roof.  `cpu_baseline` times the oracle
(reference-faithful W-form, float64) on a bounded sample on the host cores.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "reluqp-py_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector fp32 peak
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_*_f32 (f32 in / f32 acc) dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--n-eq", type=int, default=25)
    ap.add_argument("--n-ineq", type=int, default=275)
    ap.add_argument("--eps-abs", type=float, default=1e-3)
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--workload", choices=["random_qp", "mpc"], default="random_qp",
                    help="random_qp: the headline metric's workload (default); mpc: BASELINE config 3, batch of condensed "
                         "linear-MPC QPs (horizon 20, nx=12, nu=4 -> n=80, m=320) sharing H and A")
    return ap.parse_args()


def cpu_baseline(H, g, A, l, u, eps_abs, budget_s):
    """Oracle (reference-faithful W form, float64, one instance at a time: how the reference runs on CPU) on the first
    instances of the batch until the time budget is spent.  BLAS threads are capped at 8: the (n+2m)^2 = 700^2 matvec
    does not scale further (tools/cpu_baseline_threads.py on the GPU box: 171 QP/s on 1 thread, 365 on 4-16, 60 on 64,
    23 on 256), so this is the host's best configuration, not its worst.  Returns solve-only QP/s (setup amortised,
    like `value`)."""
    from oracle import reluqp_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(8, avail)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=cores)
    except Exception:                      # threadpoolctl missing: BLAS keeps its default (all cores)
        limiter, cores = None, avail
    t_setup = t_solve = 0.0
    iters = 0
    done = 0
    shared = H.ndim == 2                      # linear MPC: one (H, A) for the whole batch
    t0 = time.perf_counter()
    try:
        while done < g.shape[0] and (time.perf_counter() - t0) < budget_s:
            qp = O.OracleQP(form="W", quirks=False)
            qp.setup(H if shared else H[done], g[done], A if shared else A[done], l[done], u[done], eps_abs=eps_abs)
            r = qp.solve()
            t_setup += qp.info.setup_time
            t_solve += r.info.run_time
            iters += r.info.iter
            done += 1
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    return {
        "value": done / t_solve,
        "unit": "QP/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d instances of the same batch, oracle W-form fp64 (dense (n+2m)^2 matvec per iteration, "
                  "numpy/BLAS on %d threads of %d available), solve-only; incl. per-QP setup: %.2f QP/s; %.0f ADMM it/s"
                  % (done, cores, avail, done / (t_solve + t_setup), iters / t_solve),
        "setup_plus_solve_value": done / (t_solve + t_setup),
        "admm_iters_per_sec": iters / t_solve,
    }


def pmc_traffic(kernel, args, kern_s):
    """HBM traffic of the dominant kernel in GB/s from the PMC passes committed under profiles/ (counters cannot be
    read from inside this process; the passes are separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of
    this same command).  KB units, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md).  None when no profile of this
    kernel on this workload is committed."""
    sig = (args.workload, args.batch, args.n, args.n_eq, args.n_ineq, args.precision)
    table = {
        ("resident2", ("random_qp", 4096, 100, 25, 275, "f32")): "r1_resident2/pmc_admm_res2.json",
        ("resident", ("random_qp", 4096, 100, 25, 275, "f32")): "r1_resident/pmc_admm_resident.json",
        ("generic", ("random_qp", 4096, 100, 25, 275, "f32")): "r1_generic/pmc_admm_generic.json",
        ("wave", ("random_qp", 65536, 32, 8, 56, "f32")): "r1_wave/pmc_hbm.json",
    }
    name = table.get((kernel, sig))
    if name is None and kernel == "mfma" and args.workload == "mpc" and args.batch == 4096 and args.precision == "f32":
        name = "r1_mfma/pmc_hbm_b4096.json"
    if name is None:
        return None, None
    path = os.path.join(REPO, "profiles", name)
    try:
        with open(path) as f:
            pm = json.load(f)
        if "hbm_bytes_per_launch" in pm:
            nbytes = float(pm["hbm_bytes_per_launch"])
        else:
            nbytes = (2.0 * float(pm["FETCH_SIZE"]) + float(pm["WRITE_SIZE"])) * 1024.0
    except (OSError, KeyError, ValueError):
        return None, None
    return nbytes / kern_s / 1e9, "profiles/" + name + " (bytes per launch / this run's kernel time)"


def main():
    args = parse()
    from reluqp import distributed as D
    rank, world, local_rank, dist = D.init()          # nccl (= RCCL) when WORLD_SIZE > 1; barrier/reductions only
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    dev = D.local_device(local_rank)
    torch.cuda.set_device(dev)

    import reluqp.reluqpth as reluqpth
    from reluqp import utils

    B, n, m = args.batch, args.n, args.n_eq + args.n_ineq
    prec = torch.float32 if args.precision == "f32" else torch.float64
    esz = 4 if args.precision == "f32" else 8
    # weak scaling: the job is world*B instances; rank r owns the contiguous shard [r*B, (r+1)*B)
    start, size = D.shard_range(world * B, rank, world)
    assert size == B
    if args.workload == "mpc":
        from reluqp import mpc
        Adyn, Bdyn = mpc.random_plant(12, 4, seed=0)                      # one plant, the batch is the initial states
        ctl = mpc.LinearMPC(Adyn, Bdyn, np.eye(12), 0.1 * np.eye(4), 20, 0.5, 10.0, form="condensed")
        x0 = np.random.RandomState(args.seed0 + 1 + rank).randn(B, 12)
        H, A = ctl.H, ctl.A                                               # shared by the batch (un-batched)
        g, l, u = ctl.qp_vectors(x0)
        n, m = H.shape[0], A.shape[0]
    else:
        H, g, A, l, u, xs = utils.rand_qp_batch(B, n, args.n_eq, args.n_ineq, seed0=args.seed0 + start, feasible=True)
    # inputs resident in HBM in the working dtype before anything is timed (host->device of 655 MB is data loading)
    Hd, gd, Ad, ld, ud = (torch.from_numpy(np.ascontiguousarray(t)).to(device=dev, dtype=prec) for t in (H, g, A, l, u))
    torch.cuda.synchronize(dev)
    model = reluqpth.ReLU_QP()
    t0 = time.perf_counter()
    model.setup(Hd, gd, Ad, ld, ud, device=dev, precision=prec, eps_abs=args.eps_abs, warm_starting=False)
    torch.cuda.synchronize(dev)
    setup_s = time.perf_counter() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        res = model.solve()
    barrier()
    t0 = time.perf_counter()
    kern_s = 0.0
    for _ in range(args.steps):
        res = model.solve()           # cold start every step (warm_starting=False clears the state)
        kern_s += model.last_kernel_time
    barrier()
    elapsed = time.perf_counter() - t0

    iters = res.info.iter.to(torch.float64)
    sum_iters = float(iters.sum())
    solved = float((res.info.status_code == 0).sum())
    elapsed, (kern_avg_s, setup_max), tot_iters, tot_solved, tot_qps = D.reduce_report(
        dist, dev, elapsed, sum_iters, solved, B, extra_max=[kern_s / max(1, args.steps), setup_s])

    if rank == 0:
        step_s = elapsed / args.steps
        value = tot_qps / step_s
        # algorithmic work per launch (one rank's launch): SURVEY.md 8(d)
        b_iter = esz * (n * n + m * n)
        f_iter = 2 * n * n + 4 * m * n
        alg_bytes = sum_iters * b_iter
        achieved = alg_bytes / kern_avg_s / 1e9
        traffic, traffic_src = pmc_traffic(model.kernel, args, kern_avg_s)
        if args.workload == "mpc":
            wl = ("batch=%d/GPU condensed linear-MPC QPs (horizon 20, nx=12, nu=4: n=%d, m=%d), H and A shared by the "
                  "batch, random initial states seed %d.., eps_abs=%g, cold start, reference defaults"
                  % (B, n, m, args.seed0 + 1, args.eps_abs))
            metric = "QP solves/sec (batch=%d linear-MPC QPs n=%d m=%d per GPU)" % (B, n, m)
        else:
            wl = ("batch=%d/GPU random dense QPs n=%d m=%d (n_eq=%d), feasible rand_qp seeds %d.., eps_abs=%g, cold "
                  "start, reference defaults" % (B, n, m, args.n_eq, args.seed0, args.eps_abs))
            metric = "QP solves/sec (batch=%d random dense QPs n=%d m=%d per GPU)" % (B, n, m)
        out = {
            "metric": metric,
            "value": value,
            "unit": "QP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": wl,
                       "global_batch": int(tot_qps), "parallelism": "batch-split x%d, no collectives" % world,
                       "kernel": model.kernel},
            "admm_iters_per_sec": tot_iters / step_s,
            "mean_iters": tot_iters / tot_qps,
            "solved_frac": tot_solved / tot_qps,
            "setup_s": setup_max,
            "setup_plus_solve_qps": tot_qps / (setup_max + step_s),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_admm_%s" % model.kernel, "kernel_ms": kern_avg_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "fp32_valu_tflops": sum_iters * f_iter / kern_avg_s / 1e12,
                         "fp32_valu_frac": sum_iters * f_iter / kern_avg_s / 1e12 / FP32_VALU_PEAK_TFLOPS},
        }
        if model.kernel == "mfma":
            # the batch is the N axis of fp32 MFMA GEMMs: price against the fp32 matrix peak with the dense
            # algorithmic flops 2*(2mn + 2n^2) per instance-iteration ("k_admm_mfma")
            tf = sum_iters * f_iter / kern_avg_s / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "GB/s (HBM)",
                               "traffic_source": traffic_src,
                               "kernel": "k_admm_mfma", "kernel_ms": kern_avg_s * 1e3,
                               "algorithmic_flops_per_launch": sum_iters * f_iter,
                               "note": "dense algorithmic flops 2*(2mn+2n^2) per instance-iteration; the kernel skips all-zero "
                                       "operand groups of the block-triangular MPC matrices, so it executes fewer "
                                       "(profiles/r1_mfma/pmc_admm_mfma.json: MOPS counter)"}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(H, g, A, l, u, args.eps_abs, args.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
