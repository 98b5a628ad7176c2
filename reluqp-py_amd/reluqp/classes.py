"""Data classes of the solver API -- attribute bags whose FIELD NAMES are the
contract (reference ReLU-QP-py/reluqp/classes.py:4-95).  Batch extension: every
problem array may carry a leading batch dimension; H and A may stay un-batched
(= shared by all instances, the linear-MPC case).
"""
import numpy as np
import torch


def _default_device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _as_tensor(a):
    if isinstance(a, np.ndarray):
        return torch.from_numpy(a)
    if not torch.is_tensor(a):
        return torch.as_tensor(np.asarray(a))
    return a


class QP(object):
    """Problem data on the device (classes.py:4-30); honours device / precision (Q2 fixed)."""

    def __init__(self, H, g, A, l, u, device=None, precision=torch.double):
        device = _default_device() if device is None else device
        H, g, A, l, u = (_as_tensor(t) for t in (H, g, A, l, u))
        self.H = H.to(device=device, dtype=precision).contiguous()
        self.g = g.to(device=device, dtype=precision).contiguous()
        self.A = A.to(device=device, dtype=precision).contiguous()
        self.l = l.to(device=device, dtype=precision).contiguous()
        self.u = u.to(device=device, dtype=precision).contiguous()

        self.nx = self.H.shape[-1]  # number of decision variables
        self.nc = self.A.shape[-2]  # number of constraints
        self.batched = self.g.dim() == 2
        self.batch = self.g.shape[0] if self.batched else 1
        self.shared_mats = self.batched and self.H.dim() == 2
        self._validate()

    def _validate(self):
        n, m, B = self.nx, self.nc, self.batch
        lead = (B,) if self.batched else ()
        matlead = () if (self.shared_mats or not self.batched) else (B,)
        want = {"H": matlead + (n, n), "A": matlead + (m, n), "g": lead + (n,), "l": lead + (m,), "u": lead + (m,)}
        for name, shape in want.items():
            got = tuple(getattr(self, name).shape)
            if got != shape:
                raise ValueError("QP.%s has shape %s, expected %s" % (name, got, shape))


class Settings(object):
    """classes.py:32-65, same names and defaults."""

    def __init__(self, verbose=False,
                 warm_starting=True,
                 scaling=False,
                 rho=0.1,
                 rho_min=1e-6,
                 rho_max=1e6,
                 sigma=1e-6,
                 adaptive_rho=True,
                 adaptive_rho_interval=1,
                 adaptive_rho_tolerance=5,
                 max_iter=4000,
                 eps_abs=1e-3,
                 eq_tol=1e-6,
                 check_interval=25,
                 device=None,
                 precision=torch.float64,
                 eps_rel=0.0,
                 eps_prim_inf=1e-4,
                 eps_dual_inf=1e-4,
                 check_infeasibility=False):
        self.verbose = verbose
        self.warm_starting = warm_starting
        self.scaling = scaling
        self.rho = rho
        self.rho_min = rho_min
        self.rho_max = rho_max
        self.sigma = sigma
        self.adaptive_rho = adaptive_rho
        self.adaptive_rho_interval = adaptive_rho_interval
        self.adaptive_rho_tolerance = adaptive_rho_tolerance
        self.max_iter = max_iter
        self.eps_abs = eps_abs
        self.eq_tol = eq_tol
        self.check_interval = check_interval
        self.device = _default_device() if device is None else device
        self.precision = precision
        # extensions; the defaults reproduce the reference
        self.eps_rel = eps_rel
        self.eps_prim_inf = eps_prim_inf
        self.eps_dual_inf = eps_dual_inf
        self.check_infeasibility = check_infeasibility


class Info(object):
    """classes.py:67-88.  Un-batched solves hold scalars (iter int, status str, the
    rest 0-dim tensors as in the reference); batched solves hold [batch] tensors and
    a list of status strings (+ ``status_code`` int32 tensor)."""

    def __init__(self, iter=None,
                 status=None,
                 obj_val=None,
                 pri_res=None,
                 dua_res=None,
                 setup_time=0,
                 solve_time=0,
                 update_time=0,
                 run_time=0,
                 rho_estimate=None,
                 ):
        self.iter = iter
        self._status = status
        self.obj_val = obj_val
        self.pri_res = pri_res
        self.dua_res = dua_res
        self.setup_time = setup_time
        self.solve_time = solve_time
        self.update_time = update_time
        self.run_time = run_time
        self.rho_estimate = rho_estimate
        self.status_code = None
        self.rho_ind = None

    # batched solves keep the exit codes on the device; the list of strings ("solved" /
    # "max_iters_reached", reluqpth.py:236,245) is built on first access (one device->host copy)
    @property
    def status(self):
        if self._status is None and self.status_code is not None and hasattr(self.status_code, "cpu"):
            from reluqp._cabi import STATUS_STR
            self._status = [STATUS_STR[int(c)] for c in self.status_code.cpu().tolist()]
        return self._status

    @status.setter
    def status(self, value):
        self._status = value


class Results(object):
    """classes.py:91-95 (+ ``y``/``lam``: the dual of the returned state, Q7)."""

    def __init__(self, x=None, z=None, info: Info = None, y=None):
        self.x = x
        self.z = z
        self.y = y
        self.lam = y
        self.info = info
