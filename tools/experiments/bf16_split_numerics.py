import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/reluqp-py_amd'); sys.path.insert(0,'/tmp')
import numpy as np
from oracle import reluqp_oracle as O
from reluqp import mpc, utils
def bf16(a):
    a = np.ascontiguousarray(a, np.float32); u = a.view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split(a):
    hi = bf16(a); mid = bf16(a - hi); return hi, mid
def mm3(v, Mt):   # v [B,k] @ Mt [k,n] with 3-term split products, f32 accumulate
    vh, vm = split(v); Mh, Mm = split(Mt)
    return (vh @ Mh + vh @ Mm + vm @ Mh).astype(np.float32)
def mm1(v, Mt): return (bf16(v) @ bf16(Mt)).astype(np.float32)

def run(H, A, g, l, u, mode, eps_abs=1e-3, max_iter=4000, ci=25, kmode="bf16", chk_exact=False):
    B, n = g.shape; m = l.shape[1]
    rhos = O.setup_rhos(0.1, 1e-6, 1e6, 5.0)
    ri = np.full(B, int(np.argmin(np.abs(rhos - 0.1))))
    c = np.where((u - l) <= 1e-6, 1e3, 1.0); c0 = c[0]
    Ks = [np.linalg.inv(H + 1e-6*np.eye(n) + A.T @ ((r*c0)[:,None]*A)).astype(np.float32) for r in rhos]
    Hf, Af = H.astype(np.float32), A.astype(np.float32)
    x = np.zeros((B,n)); z = np.zeros((B,m)); lam = np.zeros((B,m)); zt = np.zeros((B,m))
    gf = g.astype(np.float32); rho_est = rhos[ri].astype(np.float32)
    done = np.zeros(B,bool); iters = np.zeros(B,int)
    thr_p, thr_d = eps_abs*np.sqrt(m), eps_abs*np.sqrt(n)
    mm = (lambda v, Mt: (v @ Mt).astype(np.float32)) if mode=="f32" else mm3
    for k in range(1, max_iter+1):
        rv = rhos[ri][:,None]*c
        p = zt - z; lam_hat = lam + rv*p
        nu = (lam_hat + rv*p).astype(np.float32); xf = x.astype(np.float32)
        d = mm(xf, Hf.T) + gf + mm(nu, Af)
        Kb = np.stack([Ks[j] for j in ri])
        if mode=="f32" or kmode=="f32": dx = -np.einsum('bij,bj->bi', Kb, d).astype(np.float32)
        elif kmode=="bf16": dx = -np.einsum('bij,bj->bi', bf16(Kb), bf16(d)).astype(np.float32)
        x = x + dx
        zt = zt + mm(dx, Af.T)
        z = np.clip(zt + lam_hat/rv, l, u); lam = lam_hat
        if k % ci == 0:
            xf = x.astype(np.float32); me = (lambda v, Mt: (v @ Mt).astype(np.float32)) if chk_exact else mm
            t1 = zt.astype(np.float32); t2 = me(xf, Hf.T); t3 = me(lam.astype(np.float32), Af)
            pri = np.abs(t1 - z).max(1); dua = np.abs(t2+t3+gf).max(1)
            num = pri/np.maximum(np.abs(t1).max(1), np.abs(z).max(1)); den = dua/np.maximum(np.maximum(np.abs(t2).max(1),np.abs(t3).max(1)),np.abs(gf).max(1))
            rho_est = np.clip(rho_est*np.sqrt(num/den), 1e-6, 1e6)
            up = (rho_est > rhos[ri]*5) & (ri < len(rhos)-1); dn = (~up) & (rho_est < rhos[ri]/5) & (ri>0)
            ri = np.where(done, ri, ri + up.astype(int) - dn.astype(int))
            conv = (pri < thr_p) & (dua < thr_d) & ~done
            iters[conv] = k; done |= conv
            if done.all(): break
    iters[~done] = max_iter
    # exact KKT check
    xf = x.astype(np.float64)
    return iters, x

Ad, Bd = mpc.random_plant(12, 4, seed=0)
ctl = mpc.LinearMPC(Ad, Bd, np.eye(12), 0.1*np.eye(4), 20, 0.5, 10.0, form="condensed")
B = 64
x0 = np.random.RandomState(1).randn(B, 12); g,l,u = ctl.qp_vectors(x0); H,A = ctl.H, ctl.A
for eps in (1e-3, 1e-5):
    it0,x_0 = run(H,A,g,l,u,"f32",eps_abs=eps); print("MPC eps",eps,"f32", np.bincount(it0//25))
    it1,x_1 = run(H,A,g,l,u,"split",eps_abs=eps); s=it1==it0
    print("   split3 + bf16 K", np.bincount(it1//25), "same %.3f" % s.mean(), "max|dx| %.2e" % np.abs(x_1-x_0)[s].max())
n, ne, ni = 80, 20, 220
H2, g0, A2, l0, u0, _ = utils.rand_qp(n, ne, ni, seed=5, compute_sol=False, feasible=True)
qs = [utils.update_qp(H2, A2, ne, ni, seed=50+b, compute_sol=False, feasible=True) for b in range(64)]
g2, l2, u2 = (np.stack([q[i] for q in qs]) for i in (1,3,4)); u2 = np.where(np.isinf(u2), 1e30, u2)
it0,x_0 = run(H2,A2,g2,l2,u2,"f32"); print("eq-rows f32", np.bincount(it0//25))
for km in ("bf16","f32"):
    it1,x_1 = run(H2,A2,g2,l2,u2,"split",kmode=km); s=it1==it0
    print("   split3, K", km, np.bincount(it1//25), "same %.3f" % s.mean(), "max|dx| %.2e" % (np.abs(x_1-x_0)[s].max() if s.any() else -1))
