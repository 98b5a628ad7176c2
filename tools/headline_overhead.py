"""Diagnostic (not product): where the wall time of one synchronous solve() of the headline batch goes beyond the kernel."""
import os, sys, time, ctypes
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R0, "reluqp-py_amd"))
import numpy as np, torch
import reluqp.reluqpth as reluqpth
from reluqp import utils, _cabi
dev = torch.device("cuda:0")
B = 4096
H, g, A, l, u, _ = utils.rand_qp_batch(B, 100, 25, 275, seed0=0, feasible=True, dtype=np.float32, workers=8)
for full in (False, True):
    m = reluqpth.ReLU_QP()
    m.setup(H, g, A, l, u, device=dev, precision=torch.float32, warm_starting=False, full_ladder=full)
    m.dispatch_history(False)
    for _ in range(3):
        m.solve()
    torch.cuda.synchronize()
    N = 20
    t0 = time.perf_counter()
    ks = 0.0
    for _ in range(N):
        m.solve(); ks += m.last_kernel_time
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N
    # the C call alone (launch + the library's own synchronisation on a windowed handle), outputs preallocated
    lib = _cabi.load()
    x = torch.empty(B, 100, device=dev); z = torch.empty(B, 300, device=dev); y = torch.empty(B, 300, device=dev)
    ints = torch.empty(3, B, device=dev, dtype=torch.int32); dbls = torch.empty(4, B, device=dev, dtype=torch.float64)
    ci = _cabi.CInfo(iter=ints[0].data_ptr(), status=ints[1].data_ptr(), rho_ind=ints[2].data_ptr(), pri_res=dbls[0].data_ptr(),
                     dua_res=dbls[1].data_ptr(), rho_estimate=dbls[2].data_ptr(), obj_val=dbls[3].data_ptr(), trace=None, trace_cap=0, reserved=0)
    st = torch.cuda.current_stream(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tc = 0.0
    for _ in range(N):
        t1 = time.perf_counter()
        lib.rqp_solve(m._h, _cabi.ptr(x), _cabi.ptr(z), _cabi.ptr(y), ctypes.byref(ci), ctypes.c_void_p(st.cuda_stream))
        tc += time.perf_counter() - t1
        torch.cuda.synchronize()
    wall_c = (time.perf_counter() - t0) / N
    print("full_ladder=%s: solve() wall %.3f ms, event kernel %.3f ms; bare rqp_solve + sync wall %.3f ms (call returns after %.3f ms)"
          % (full, wall * 1e3, ks / N * 1e3, wall_c * 1e3, tc / N * 1e3), flush=True)
    # asynchronous enqueue of N solves, one synchronisation at the end
    m.synchronous = False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        m.solve()
    torch.cuda.synchronize()
    print("   synchronous=False: %.3f ms per solve" % ((time.perf_counter() - t0) / N * 1e3), flush=True)
    del m
