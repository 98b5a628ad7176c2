import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/reluqp-py_amd')
import numpy as np
from oracle import reluqp_oracle as O
from reluqp import utils
def run(H, A, g, l, u, mode, eps_abs=1e-3, max_iter=4000, ci=25):
    n, m = H.shape[0], A.shape[0]
    rhos = O.setup_rhos(0.1, 1e-6, 1e6, 5.0)
    ri = int(np.argmin(np.abs(rhos - 0.1)))
    c = np.where((u - l) <= 1e-6, 1e3, 1.0)
    K64 = [np.linalg.inv(H + 1e-6*np.eye(n) + A.T @ ((r*c)[:,None]*A)) for r in rhos]
    Ks = [k.astype(np.float32) for k in K64]
    Hf, Af, gf = H.astype(np.float32), A.astype(np.float32), g.astype(np.float32)
    AK = [ (A @ k).astype(np.float32) for k in K64]     # computed in f64 at setup, rounded
    HK = [ (H @ k).astype(np.float32) for k in K64]
    x = np.zeros(n); z = np.zeros(m); lam = np.zeros(m); zt = np.zeros(m)
    rho_est = np.float32(rhos[ri])
    thr_p, thr_d = eps_abs*np.sqrt(m), eps_abs*np.sqrt(n)
    for k in range(1, max_iter+1):
        rv = rhos[ri]*c
        p = zt - z; lam_hat = lam + rv*p
        nu = (lam_hat + rv*p).astype(np.float32); xf = x.astype(np.float32)
        if mode == "refine":
            d = Hf @ xf + gf + Af.T @ nu
            dx = -(Ks[ri] @ d)
        else:
            Kg = Ks[ri] @ gf
            dx = -(HK[ri].T @ xf + AK[ri].T @ nu + Kg)
        dx = dx.astype(np.float32)
        x = x + dx
        zt = zt + (Af @ dx)
        z = np.clip(zt + lam_hat/rv, l, u); lam = lam_hat
        if k % ci == 0:
            xf = x.astype(np.float32)
            t1 = zt.astype(np.float32); t2 = Hf @ xf; t3 = Af.T @ lam.astype(np.float32)
            pri = np.abs(t1 - z).max(); dua = np.abs(t2+t3+gf).max()
            num = pri/max(np.abs(t1).max(), np.abs(z).max()); den = dua/max(np.abs(t2).max(),np.abs(t3).max(),np.abs(gf).max())
            rho_est = np.clip(rho_est*np.sqrt(num/den), 1e-6, 1e6)
            if rho_est > rhos[ri]*5 and ri < len(rhos)-1: ri += 1
            elif rho_est < rhos[ri]/5 and ri > 0: ri -= 1
            if pri < thr_p and dua < thr_d: return k, x
    return max_iter, x
res = []
for seed in range(int(sys.argv[1]) if len(sys.argv)>1 else 40):
    H,g,A,l,u,xs = utils.rand_qp(100,25,275,seed=seed,compute_sol=False,feasible=True)
    u = np.where(np.isinf(u), 1e30, u)
    k0,x0 = run(H,A,g,l,u,"refine"); k1,x1 = run(H,A,g,l,u,"fused")
    res.append((k0,k1,np.abs(x0-x1).max()))
res=np.array(res)
print("same", np.mean(res[:,0]==res[:,1]), "iters refine", res[:,0].mean(), "fused", res[:,1].mean()); print(res[:12])
print(res[res[:,0]!=res[:,1]][:10])
