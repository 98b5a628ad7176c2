"""Diagnostic: the oracle's W-form float64 solve rate on the host for several BLAS thread counts (which thread count
should bench.py's cpu_baseline leg use?).  python tools/cpu_baseline_threads.py"""
import os, subprocess, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import reluqp_oracle as O
from reluqp import utils
H, g, A, l, u, _ = utils.rand_qp_batch(6, 100, 25, 275, seed0=0, feasible=True)
t_solve = 0.0; its = 0
for i in range(6):
    qp = O.OracleQP(form="W"); qp.setup(H[i], g[i], A[i], l[i], u[i], eps_abs=1e-3)
    r = qp.solve(); t_solve += r.info.run_time; its += r.info.iter
print("%%.1f QP/s  %%.0f it/s" %% (6 / t_solve, its / t_solve))
''' % (R0, os.path.join(R0, "reluqp-py_amd"))
for nt in (1, 4, 16, 64, 256):
    env = dict(os.environ, OMP_NUM_THREADS=str(nt), OPENBLAS_NUM_THREADS=str(nt), MKL_NUM_THREADS=str(nt))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("threads", nt, out.stdout.strip() or out.stderr.strip()[-200:])
