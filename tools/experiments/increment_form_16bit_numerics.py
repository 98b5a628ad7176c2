import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/reluqp-py_amd')
import numpy as np
from oracle import reluqp_oracle as O
from reluqp import mpc

def bf16(a):
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def f16(a): return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)

def run(H, A, g, l, u, mode, rnd=bf16, eps_abs=1e-3, max_iter=4000, ci=25, split=False):
    B, n = g.shape; m = l.shape[1]
    st = O.Settings()
    rhos = O.setup_rhos(0.1, 1e-6, 1e6, 5.0)
    ri = np.full(B, int(np.argmin(np.abs(rhos - 0.1))))
    c = np.where((u - l) <= 1e-6, 1e3, 1.0)          # [B,m]; shared pattern
    c0 = c[0]
    Ks = [np.linalg.inv(H + 1e-6*np.eye(n) + A.T @ ((r*c0)[:,None]*A)).astype(np.float32) for r in rhos]
    Hf, Af = H.astype(np.float32), A.astype(np.float32)
    if mode == "incr":
        Hq, Aq = rnd(Hf), rnd(Af); Kq = [rnd(K) for K in Ks]
    x = np.zeros((B,n)); z = np.zeros((B,m)); lam = np.zeros((B,m)); zt = np.zeros((B,m))
    gf = g.astype(np.float32)
    rho_est = rhos[ri].astype(np.float32)
    done = np.zeros(B,bool); iters = np.zeros(B,int)
    d = None; nu_prev=None; dx_prev=None
    thr_p, thr_d = eps_abs*np.sqrt(m), eps_abs*np.sqrt(n)
    def q(v):
        if not split: return rnd(v)
        hi = rnd(v); return hi + rnd(v - hi)
    for k in range(1, max_iter+1):
        rv = rhos[ri][:,None]*c
        p = zt - z
        lam_hat = lam + rv*p
        nu = (lam_hat + rv*p).astype(np.float32)
        xf = x.astype(np.float32)
        if mode == "refine" or d is None:
            d = xf @ Hf.T + gf + nu @ Af          # exact f32
        else:
            dnu = (nu - nu_prev).astype(np.float32)
            d = d + q(dx_prev) @ Hq.T + q(dnu) @ Aq
        Kb = np.stack([ (Kq if mode=="incr" else Ks)[j] for j in ri])   # [B,n,n]
        dvec = q(d) if mode=="incr" else d
        dx = -np.einsum('bij,bj->bi', Kb, dvec).astype(np.float32)
        x = x + dx
        adx = (q(dx) @ Aq.T) if mode=="incr" else (dx @ Af.T)
        zt = zt + adx
        z = np.clip(zt + lam_hat/rv, l, u)
        lam = lam_hat
        nu_prev, dx_prev = nu, dx
        if k % ci == 0:
            xf = x.astype(np.float32)
            if mode == "incr": zt = (xf @ Af.T).astype(np.float64)     # exact refresh of A x
            t1 = zt.astype(np.float32); t2 = xf @ Hf.T; t3 = lam.astype(np.float32) @ Af
            pri = np.abs(t1 - z).max(1); dua = np.abs(t2+t3+gf).max(1)
            num = pri/np.maximum(np.abs(t1).max(1), np.abs(z).max(1)); den = dua/np.maximum(np.maximum(np.abs(t2).max(1),np.abs(t3).max(1)),np.abs(gf).max(1))
            rho_est = np.clip(rho_est*np.sqrt(num/den), 1e-6, 1e6)
            up = (rho_est > rhos[ri]*5) & (ri < len(rhos)-1); dn = (~up) & (rho_est < rhos[ri]/5) & (ri>0)
            ri = np.where(done, ri, ri + up.astype(int) - dn.astype(int))
            conv = (pri < thr_p) & (dua < thr_d) & ~done
            iters[conv] = k; done |= conv
            d = None                                   # exact d at the next iteration
            if done.all(): break
    iters[~done] = max_iter
    return iters, x

Ad, Bd = mpc.random_plant(12, 4, seed=0)
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1*np.eye(4), 20, 0.5, 10.0, form="condensed")
B = int(sys.argv[1]) if len(sys.argv)>1 else 128
x0 = np.random.RandomState(1).randn(B, 12)
g,l,u = ctl.qp_vectors(x0)
H,A = ctl.H, ctl.A
print("n,m", H.shape, A.shape, "max|A|", np.abs(A).max(), "max|H|", np.abs(H).max())
t=time.time(); it0,x_0 = run(H,A,g,l,u,"refine"); print("refine f32", np.bincount(it0//25), time.time()-t)
for name,rnd,split in (("bf16",bf16,False),("f16",f16,False),("bf16x2 vectors",bf16,True)):
    it1,x_1 = run(H,A,g,l,u,"incr",rnd,split=split)
    print(name, np.bincount(it1//25), "same iters %.3f" % np.mean(it1==it0), "mean", it1.mean(), it0.mean(), "max|dx| %.2e" % np.abs(x_1-x_0)[it1==it0].max())

print("---- variants with split matrices")
def run2(H,A,g,l,u,rnd,msplit,vsplit):
    # monkeypatch: emulate split matrices by using higher-precision rounding for matrices
    global bf16
    return None
# random shared-matrix QP with equality rows (rho x 1e3 rows)
from reluqp import utils
n, ne, ni = 80, 20, 220
H2, g0, A2, l0, u0, _ = utils.rand_qp(n, ne, ni, seed=5, compute_sol=False, feasible=True)
qs = [utils.update_qp(H2, A2, ne, ni, seed=50+b, compute_sol=False, feasible=True) for b in range(64)]
g2, l2, u2 = (np.stack([q[i] for q in qs]) for i in (1,3,4))
u2 = np.where(np.isinf(u2), 1e30, u2)
it0,x_0 = run(H2,A2,g2,l2,u2,"refine"); print("eq-rows problem refine f32", np.bincount(it0//25))
for name,rnd,split in (("bf16",bf16,False),("f16",f16,False),("bf16x2 vectors",bf16,True)):
    with np.errstate(all="ignore"):
        it1,x_1 = run(H2,A2,g2,l2,u2,"incr",rnd,split=split)
    same = it1==it0
    print(name, np.bincount(it1//25), "same iters %.3f" % np.mean(same), "max|dx| %.2e" % (np.abs(x_1-x_0)[same].max() if same.any() else np.nan), "nan", np.isnan(x_1).any())
