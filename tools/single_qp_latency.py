"""Diagnostic (not product): end-to-end latency of setup() / solve() for ONE QP through the Python API."""
import sys, time, numpy as np, torch
import os; R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R0, 'reluqp-py_amd')); sys.path.insert(0, R0)
import reluqp.reluqpth as R
from reluqp import utils
for (n,ne,ni,prec) in [(10,5,15,torch.float32),(10,5,15,torch.float64),(100,25,275,torch.float32),(100,25,275,torch.float64)]:
    H,g,A,l,u,_=utils.rand_qp(n,ne,ni,seed=1,compute_sol=False,feasible=True)
    m=R.ReLU_QP()
    t0=time.perf_counter(); m.setup(H,g,A,l,u,precision=prec); torch.cuda.synchronize(); ts=time.perf_counter()-t0
    r=m.solve()
    m.clear_primal_dual()
    t=[]
    for _ in range(20):
        m.clear_primal_dual()
        t0=time.perf_counter(); r=m.solve(); t.append(time.perf_counter()-t0)
    print(n, ne+ni, str(prec)[6:], m.kernel, 'setup %.2f ms'%(ts*1e3), 'solve wall %.3f ms (min %.3f)'%(np.median(t)*1e3, min(t)*1e3), 'kernel %.3f ms'%(m.last_kernel_time*1e3), 'iters', r.info.iter, r.info.status, 'run_time %.3f ms'%(r.info.run_time*1e3))
