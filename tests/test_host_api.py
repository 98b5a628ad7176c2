"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol
include/rqp_abi.h declares, host-only entry points behave, the Python mirror has the
reference's API surface, and the product path FAILS LOUDLY without a GPU (no CPU
fallback).  No compute calls are made here."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

from reluqp import _cabi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "rqp_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rqp_[a-z_A-Z0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _cabi.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    assert sorted(_cabi.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), "librqp_hip.so does not export %s" % name


def test_host_only_entry_points():
    lib = _cabi.load()
    assert b"gfx950" in lib.rqp_version()
    assert lib.rqp_strerror(0) == b"ok"
    assert lib.rqp_strerror(-1) == b"invalid argument"
    s = _cabi.CSettings()
    assert lib.rqp_default_settings(ctypes.byref(s)) == 0
    # defaults of reference classes.py:36-46
    assert (s.rho, s.rho_min, s.rho_max, s.sigma) == (0.1, 1e-6, 1e6, 1e-6)
    assert (s.adaptive_rho_tolerance, s.eps_abs, s.eq_tol) == (5.0, 1e-3, 1e-6)
    assert (s.adaptive_rho, s.max_iter, s.check_interval, s.warm_starting) == (1, 4000, 25, 1)
    assert lib.rqp_default_settings(None) == -1
    # argument validation happens before any device call
    h = ctypes.c_void_p()
    bad = _cabi.Dims(n=0, m=5, batch=1, shared_mats=0, dtype=0, kernel=0, tile_dtype=0, flags=0)
    assert lib.rqp_create(ctypes.byref(h), ctypes.byref(bad), ctypes.byref(s), 0) == -1
    bad = _cabi.Dims(n=3, m=5, batch=1, shared_mats=0, dtype=7, kernel=0, tile_dtype=0, flags=0)
    assert lib.rqp_create(ctypes.byref(h), ctypes.byref(bad), ctypes.byref(s), 0) == -1
    ok = _cabi.Dims(n=3, m=5, batch=1, shared_mats=0, dtype=0, kernel=0, tile_dtype=0, flags=0)
    s.adaptive_rho_tolerance = 1.0
    assert lib.rqp_create(ctypes.byref(h), ctypes.byref(ok), ctypes.byref(s), 0) == -1
    assert lib.rqp_solve(None, None, None, None, None, None) == -1
    assert lib.rqp_destroy(None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    """Without a HIP device the product raises; it never computes on the CPU."""
    import reluqp.reluqpth as reluqpth
    lib = _cabi.load()
    s = _cabi.CSettings()
    lib.rqp_default_settings(ctypes.byref(s))
    h = ctypes.c_void_p()
    ok = _cabi.Dims(n=3, m=5, batch=1, shared_mats=0, dtype=0, kernel=0, tile_dtype=0, flags=0)
    assert lib.rqp_create(ctypes.byref(h), ctypes.byref(ok), ctypes.byref(s), 0) == -3   # RQP_ERR_HIP
    m = reluqpth.ReLU_QP()
    with pytest.raises(_cabi.RqpUnavailable):
        m.setup(np.eye(3), np.ones(3), np.eye(3), -np.ones(3), np.ones(3))
    with pytest.raises(_cabi.RqpUnavailable):
        m.setup(np.eye(3), np.ones(3), np.eye(3), -np.ones(3), np.ones(3), device=torch.device("cpu"))
    with pytest.raises(RuntimeError):
        m.solve()


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "reluqp-py_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "reluqp_oracle" not in txt or f.endswith(".hip"), f   # .hip files only cite it in comments


def test_api_surface_matches_reference():
    """Method names, keyword names and defaults of reference reluqpth.py:102-117,159-160,185,201,251-254."""
    import reluqp.reluqpth as reluqpth
    from reluqp.classes import Info, Results, Settings, QP  # noqa: F401
    sig = inspect.signature(reluqpth.ReLU_QP.setup)
    want = dict(verbose=False, warm_starting=True, scaling=False, rho=0.1, rho_min=1e-6, rho_max=1e6,
                sigma=1e-6, adaptive_rho=True, adaptive_rho_interval=1, adaptive_rho_tolerance=5,
                max_iter=4000, eps_abs=1e-3, check_interval=25, precision=torch.float64)
    for k, v in want.items():
        assert sig.parameters[k].default == v, k
    assert list(sig.parameters)[:6] == ["self", "H", "g", "A", "l", "u"]
    assert list(inspect.signature(reluqpth.ReLU_QP.update).parameters) == ["self", "g", "l", "u", "Hx", "Ax"]
    assert list(inspect.signature(reluqpth.ReLU_QP.warm_start).parameters) == ["self", "x", "z", "lam", "rho"]
    for name in ("solve", "update_settings", "clear_primal_dual"):
        assert callable(getattr(reluqpth.ReLU_QP, name))
    info = Info()
    for f in ("iter", "status", "obj_val", "pri_res", "dua_res", "setup_time", "solve_time", "update_time",
              "run_time", "rho_estimate"):
        assert hasattr(info, f)                                    # classes.py:67-88
    r = Results(info=info)
    assert hasattr(r, "x") and hasattr(r, "z") and r.info is info
    s = Settings()
    assert (s.rho, s.eq_tol, s.check_interval, s.max_iter) == (0.1, 1e-6, 25, 4000)


def test_qp_shapes_and_batch_detection():
    from reluqp.classes import QP
    cpu = torch.device("cpu")
    q = QP(np.eye(3), np.ones(3), np.ones((5, 3)), -np.ones(5), np.ones(5), device=cpu)
    assert (q.nx, q.nc, q.batch, q.batched, q.shared_mats) == (3, 5, 1, False, False)
    q = QP(np.eye(3), np.ones((4, 3)), np.ones((5, 3)), -np.ones((4, 5)), np.ones((4, 5)), device=cpu,
           precision=torch.float32)
    assert (q.batch, q.batched, q.shared_mats) == (4, True, True) and q.H.dtype == torch.float32
    q = QP(np.zeros((4, 3, 3)), np.ones((4, 3)), np.ones((4, 5, 3)), -np.ones((4, 5)), np.ones((4, 5)), device=cpu)
    assert (q.batch, q.shared_mats) == (4, False)
    with pytest.raises(ValueError):
        QP(np.eye(3), np.ones(3), np.ones((5, 2)), -np.ones(5), np.ones(5), device=cpu)
    with pytest.raises(ValueError):
        QP(np.zeros((2, 3, 3)), np.ones((4, 3)), np.ones((4, 5, 3)), -np.ones((4, 5)), np.ones((4, 5)), device=cpu)


def test_generators():
    from reluqp import utils
    H, g, A, l, u, xs = utils.rand_qp(8, 3, 9, seed=4, feasible=True)
    # the planted point satisfies the KKT conditions exactly (feasible variant)
    assert np.allclose(A[:3] @ xs, l[:3]) and np.all(A[3:] @ xs >= l[3:] - 1e-12)
    assert np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > 0)
    Hb, gb, Ab, lb, ub, xb = utils.rand_qp_batch(3, 8, 3, 9, seed0=4, feasible=True)
    assert np.array_equal(Hb[0], H) and np.array_equal(gb[0], g) and np.array_equal(xb[0], xs)
    H2, g2, A2, l2, u2, _ = utils.update_qp(H, A, 3, 9, seed=9, compute_sol=False)
    assert np.array_equal(H2, H) and np.array_equal(A2, A) and not np.array_equal(g2, g)


def test_bench_cpu_baseline_leg_runs_without_a_gpu():
    """bench.py's cpu_baseline leg = child processes that never import torch: one worker on a tiny sample, JSON out."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--n", "10", "--n-eq", "3", "--n-ineq", "12", "--batch", "8",
                          "--cpu-worker", "refine:f32:0:2:5"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["done"] == 4 and d["t_solve"] > 0 and d["iters"] >= 4 * 25      # instances 0, 2, 4, 6 of the batch
    sys.path.insert(0, REPO)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--workload", "c4"]
        a = bench.parse()
        assert (a.n, a.n_eq, a.n_ineq, a.batch, a.scaling) == (32, 8, 56, 8192, "weak")
        sys.argv = ["bench.py", "--workload", "c4", "--scaling", "strong"]
        assert bench.parse().batch == 65536
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert (a.batch, a.n, a.n_eq + a.n_ineq, a.precision, a.gpus) == (4096, 100, 300, "f32", 1)
    finally:
        sys.argv = old
