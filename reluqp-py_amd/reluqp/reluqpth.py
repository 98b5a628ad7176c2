"""OSQP-style solver API: drop-in for the reference's ``reluqp.reluqpth``
(ReLU-QP-py/reluqp/reluqpth.py:92-333) on MI355X.

    import reluqp.reluqpth as reluqpth
    model = reluqpth.ReLU_QP()
    model.setup(H, g, A, l, u, eps_abs=1e-4)
    results = model.solve()          # results.x, results.z, results.info.{iter,status,...}
    model.update(g=g2, l=l2, u=u2); results = model.solve()     # warm-started

Same method names, keyword arguments, defaults and Results/Info fields as the
reference.  Host code here is plumbing only: tensors in HBM, raw pointers and the
current HIP stream handed to librqp_hip.so (include/rqp_abi.h).  All arithmetic of
setup()/solve()/update() runs in hand-written gfx950 kernels; there is no torch or
CPU fallback and the calls raise if the library or a GPU is missing.

Extensions: a leading batch dimension on g, l, u (and
optionally on H, A -- un-batched H, A are shared by the batch), ``precision``
honoured (float32 / float64), ``eq_tol`` settable, ``Results.y`` (dual), torch or
numpy accepted everywhere, per-instance info tensors in batch mode.

Reference quirks are handled as listed in SURVEY.md Appendix B (DESIGN.md has the
table): Q1/Q2/Q3/Q6/Q8/Q9/Q11/Q13/Q14 fixed, Q4/Q5/Q10/Q15/Q16/Q17 replicated.
"""
import contextlib
import ctypes

import numpy as np
import torch

from reluqp import _cabi
from reluqp.classes import QP, Info, Results, Settings, _as_tensor, _default_device

_CHANGEABLE = ("max_iter", "eps_abs", "verbose", "check_interval", "eps_rel", "check_infeasibility")
_FROZEN = ("rho", "rho_min", "rho_max", "sigma", "adaptive_rho", "adaptive_rho_interval",
           "adaptive_rho_tolerance")


class _Layers(object):
    """Stand-in for the reference's ``ReLU_QP.layers`` (class ReLU_Layer, reluqpth.py:8-89):
    exposes the rho ladder and the per-rho KKT inverses; W is never materialised."""

    def __init__(self, owner):
        self._o = owner
        self.rhos = owner._rhos

    def K(self, rho_ind, instance=0):
        """K_j = (H + sigma I + A' diag(rho_j c) A)^-1 of one instance (reluqpth.py:56)."""
        o = self._o
        out = torch.empty(o.QP.nx, o.QP.nx, device=o.settings.device, dtype=o.settings.precision)
        lib = _cabi.load()
        _cabi.check(o._h, lib.rqp_get_K(o._h, int(instance), int(rho_ind), _cabi.ptr(out), o._stream()), "rqp_get_K")
        torch.cuda.current_stream(o.settings.device).synchronize()
        return out


class ReLU_QP(object):
    def __init__(self):
        super().__init__()
        self.info = Info()
        self.results = Results(info=self.info)
        self._h = None
        self.settings = None
        self.QP = None
        # True (reference behaviour, reluqpth.py:201-249): update()/solve() return after the device has finished and
        # fill the *_time fields.  False: they only enqueue on the current stream (batched results are device tensors
        # ordered on that stream; times read 0) -- closed-loop drivers keep the GPU busy instead of the host waiting.
        self.synchronous = True
        self.last_kernel_time = None
        self._shards = None           # setup(devices=[...]): one child solver per device (reluqp/multidevice.py)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.settings.device).cuda_stream)

    def _events(self):
        # one pair per solver object, reused by every timed call (each call synchronises on the second before returning)
        ev = getattr(self, "_ev_pair", None)
        if ev is None:
            ev = self._ev_pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        return ev

    def _csettings(self):
        s = self.settings
        return _cabi.CSettings(rho=s.rho, rho_min=s.rho_min, rho_max=s.rho_max, sigma=s.sigma,
                               adaptive_rho_tolerance=s.adaptive_rho_tolerance, eps_abs=s.eps_abs,
                               eq_tol=s.eq_tol, adaptive_rho=int(bool(s.adaptive_rho)),
                               max_iter=int(s.max_iter), check_interval=int(s.check_interval),
                               warm_starting=int(bool(s.warm_starting)), eps_rel=float(s.eps_rel),
                               eps_prim_inf=float(s.eps_prim_inf), eps_dual_inf=float(s.eps_dual_inf),
                               scaling=(10 if s.scaling is True else int(s.scaling or 0)),
                               check_infeasibility=int(bool(s.check_infeasibility)))

    def _to_dev(self, a, shape, name):
        t = _as_tensor(a).to(device=self.settings.device, dtype=self.settings.precision).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
        return t

    def __del__(self):
        self._destroy()

    def _destroy(self):
        self._ev_pair = None              # (events belong to the device of the handle)
        if getattr(self, "_shards", None):
            self._shards.destroy()
            self._shards = None
        if getattr(self, "_h", None):
            try:
                _cabi.load().rqp_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    # -------------------------------------------------------------------- setup
    def setup(self, H, g, A, l, u,
              verbose=False,
              warm_starting=True,
              scaling=False,  # reference: accepted, unused TODO (reluqpth.py:105, Q12).  Here: True / k = Ruiz passes
              rho=0.1,
              rho_min=1e-6,
              rho_max=1e6,
              sigma=1e-6,
              adaptive_rho=True,
              adaptive_rho_interval=1,
              adaptive_rho_tolerance=5,
              max_iter=4000,
              eps_abs=1e-3,
              check_interval=25,
              device=None,
              precision=torch.float64,
              eq_tol=1e-6,
              eps_rel=0.0,
              check_infeasibility=False,
              eps_prim_inf=1e-4,
              eps_dual_inf=1e-4,
              kernel="auto",
              iterate_dtype=None,
              devices=None,
              low_memory=False,
              full_ladder=False):
        """
        Setup ReLU-QP solver problem of the form

        minimize     1/2 x' * H * x + g' * x
        subject to   l <= A * x <= u

        solver settings can be specified as additional keyword arguments
        (reference reluqpth.py:102-157).  g, l, u may carry a leading batch
        dimension; H and A then either carry it too or are shared.

        Extensions beyond the reference (all default to its behaviour): ``eps_rel``, ``scaling`` (k Ruiz passes; results,
        residuals and the termination test in the caller's units), ``check_infeasibility`` (OSQP's
        certificates at every check on the streaming kernel, which ``kernel="auto"`` then picks; an explicitly requested
        resident / wave / mfma kernel labels an infeasible instance only after it has spent ``max_iter``); ``kernel`` = "auto" | "generic" | "resident" | "wave" |
        "mfma" (C-ABI rqp_dims.kernel; an explicit kernel that cannot hold the problem raises);
        ``iterate_dtype=torch.float16`` keeps the K(rho) tile of the register-resident kernels in fp16
        (BASELINE config 5; H, A, state and residuals stay float32); ``iterate_dtype=torch.bfloat16`` runs a shared-(H, A) batch
        on the bf16 matrix pipe (every matrix tile and vector operand as two bf16 planes, three 16-bit MFMAs per product,
        float32 accumulation, state and residuals: rqp_abi.h RQP_TILE_BF16).  ``devices=[0, 1, ...]`` splits a batch
        contiguously over several GPUs inside this process (one handle and stream per device, results gathered on
        devices[0]; no collective -- reluqp/multidevice.py).  ``low_memory=True`` (rqp_dims.flags RQP_FLAG_LOW_MEMORY): the
        resident float32 kernel keeps and reads K(rho) as the row-major table instead of the larger tile-padded register
        image -- ~12 % less workspace, 1 % more solve time, bit-identical results.  ``full_ladder=True`` (RQP_FLAG_FULL_LADDER): build
        K(rho) for every entry of the rho ladder of every matrix as the reference does (reluqpth.py:52-78); by default batches
        of >= 32 per-instance matrices keep a window of 5 entries around each instance's index and re-factor on demand
        (bit-identical results, ~3x less setup time and workspace; solve() then synchronises the stream, so use
        full_ladder=True under HIP-graph capture).
        """
        if devices is not None:
            from reluqp.multidevice import DeviceShards
            if not torch.cuda.is_available():
                raise _cabi.RqpUnavailable("ReLU_QP needs a HIP device; the MI355X build has no CPU path")
            self._destroy()
            kw = dict(verbose=verbose, warm_starting=warm_starting, scaling=scaling, rho=rho, rho_min=rho_min, rho_max=rho_max,
                      sigma=sigma, adaptive_rho=adaptive_rho, adaptive_rho_interval=adaptive_rho_interval,
                      adaptive_rho_tolerance=adaptive_rho_tolerance, max_iter=max_iter, eps_abs=eps_abs,
                      check_interval=check_interval, precision=precision, eq_tol=eq_tol, eps_rel=eps_rel,
                      check_infeasibility=check_infeasibility, eps_prim_inf=eps_prim_inf, eps_dual_inf=eps_dual_inf,
                      kernel=kernel, iterate_dtype=iterate_dtype, low_memory=low_memory, full_ladder=full_ladder)
            self._shards = DeviceShards(ReLU_QP, list(devices), H, g, A, l, u, kw)
            first = self._shards.children[0]
            self.settings, self.QP, self.layers, self._rhos = first.settings, first.QP, first.layers, first._rhos
            self.rho_ind = first.rho_ind
            self.kernel = self._shards.kernel
            self.results.info.setup_time = max(c.results.info.setup_time for c in self._shards.children)
            return None
        device = _default_device() if device is None else torch.device(device)
        if precision not in (torch.float32, torch.float64):
            raise ValueError("precision must be torch.float32 or torch.float64")
        if device.type != "cuda":
            raise _cabi.RqpUnavailable(
                "ReLU_QP needs a HIP device (got device=%s); the MI355X build has no CPU path" % device)
        lib = _cabi.load()
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._destroy()

        self.settings = Settings(verbose=verbose, warm_starting=warm_starting, scaling=scaling, rho=rho,
                                 rho_min=rho_min, rho_max=rho_max, sigma=sigma, adaptive_rho=adaptive_rho,
                                 adaptive_rho_interval=adaptive_rho_interval,
                                 adaptive_rho_tolerance=adaptive_rho_tolerance, max_iter=max_iter,
                                 eps_abs=eps_abs, eq_tol=eq_tol, check_interval=check_interval,
                                 device=device, precision=precision, eps_rel=eps_rel,
                                 eps_prim_inf=eps_prim_inf, eps_dual_inf=eps_dual_inf,
                                 check_infeasibility=check_infeasibility)
        if kernel not in _cabi.KERNELS:
            raise ValueError("kernel must be one of %s" % sorted(_cabi.KERNELS))
        if iterate_dtype not in (None, precision, torch.float16, torch.bfloat16):
            raise ValueError("iterate_dtype must be None, the working precision, torch.float16 or torch.bfloat16")
        if iterate_dtype in (torch.float16, torch.bfloat16) and precision != torch.float32:
            raise ValueError("iterate_dtype=torch.float16 / torch.bfloat16 needs precision=torch.float32")
        tile = {torch.float16: _cabi.RQP_TILE_F16, torch.bfloat16: _cabi.RQP_TILE_BF16}.get(iterate_dtype, _cabi.RQP_TILE_SAME)
        with torch.cuda.device(device):
            start, end = self._events()
            start.record()
            self.QP = QP(H, g, A, l, u, device=device, precision=precision)
            qp = self.QP
            dims = _cabi.Dims(n=qp.nx, m=qp.nc, batch=qp.batch, shared_mats=int(qp.shared_mats),
                              dtype=_cabi.RQP_F32 if precision == torch.float32 else _cabi.RQP_F64,
                              kernel=_cabi.KERNELS[kernel], tile_dtype=tile,
                              flags=(_cabi.FLAG_LOW_MEMORY if low_memory else 0) | (_cabi.FLAG_FULL_LADDER if full_ladder else 0))
            cs = self._csettings()
            h = ctypes.c_void_p()
            _cabi.check(None, lib.rqp_create(ctypes.byref(h), ctypes.byref(dims), ctypes.byref(cs), device.index),
                        "rqp_create")
            self._h = h
            _cabi.check(h, lib.rqp_setup(h, _cabi.ptr(qp.H), _cabi.ptr(qp.g), _cabi.ptr(qp.A), _cabi.ptr(qp.l),
                                         _cabi.ptr(qp.u), self._stream()), "rqp_setup")
            cnt = ctypes.c_int32()
            _cabi.check(h, lib.rqp_get_rhos(h, None, 0, ctypes.byref(cnt)), "rqp_get_rhos")
            buf = (ctypes.c_double * cnt.value)()
            _cabi.check(h, lib.rqp_get_rhos(h, buf, cnt.value, ctypes.byref(cnt)), "rqp_get_rhos")
            self._rhos = torch.tensor(list(buf), dtype=precision, device=device)
            self.layers = _Layers(self)
            self.rho_ind = int(np.argmin(np.abs(np.array(list(buf)) - self.settings.rho)))
            end.record()
            if self.synchronous:
                end.synchronize()
                self.results.info.setup_time = start.elapsed_time(end) / 1000.0
            else:                    # enqueue only (devices=[...]: every shard is set up before anything is waited for)
                self.results.info.setup_time = 0.0
        self.kernel = lib.rqp_kernel_name(h).decode()
        return None

    def _finish_setup_time(self):
        """setup_time of a setup() that ran with ``synchronous = False``, once its stream has been waited for."""
        start, end = self._events()
        self.results.info.setup_time = start.elapsed_time(end) / 1000.0

    # ------------------------------------------------------------------- update
    def update(self, g=None, l=None, u=None, Hx=None, Ax=None):
        """
        Update ReLU-QP problem arguments (reference reluqpth.py:159-183; numpy or torch, Q9)
        """
        if self._shards:
            return self._shards.update(g=g, l=l, u=u, Hx=Hx, Ax=Ax)
        self._need_setup()
        lib = _cabi.load()
        qp = self.QP
        lead = (qp.batch,) if qp.batched else ()
        with torch.cuda.device(self.settings.device):
            start, end = self._events()
            start.record()
            # The reference asserts here (`updating Hx and Ax is not supported yet`, reluqpth.py:176-177).  SURVEY.md
            # 8(f)-4: Hx / Ax are the new dense H / A (same shapes as at setup); the device re-runs the setup chain
            # (C-ABI rqp_update_mats) and keeps the ADMM state, so the next solve() is warm-started.
            if Hx is not None or Ax is not None:
                matlead = (qp.batch,) if (qp.batched and not qp.shared_mats) else ()
                if Hx is not None:
                    qp.H = self._to_dev(Hx, matlead + (qp.nx, qp.nx), "Hx")
                if Ax is not None:
                    qp.A = self._to_dev(Ax, matlead + (qp.nc, qp.nx), "Ax")
                both = bool(self.settings.scaling)       # with scaling the packed copies are scaled: the ABI wants both raw matrices
                _cabi.check(self._h, lib.rqp_update_mats(self._h, _cabi.ptr(qp.H if (Hx is not None or both) else None),
                                                         _cabi.ptr(qp.A if (Ax is not None or both) else None), self._stream()),
                            "rqp_update_mats")
            if g is not None:
                qp.g = self._to_dev(g, lead + (qp.nx,), "g")
            if l is not None:
                qp.l = self._to_dev(l, lead + (qp.nc,), "l")
            if u is not None:
                qp.u = self._to_dev(u, lead + (qp.nc,), "u")
            _cabi.check(self._h, lib.rqp_update(self._h, _cabi.ptr(qp.g if g is not None else None),
                                                _cabi.ptr(qp.l if l is not None else None),
                                                _cabi.ptr(qp.u if u is not None else None), self._stream()),
                        "rqp_update")
            end.record()
            if self.synchronous:
                end.synchronize()
                self.results.info.update_time = start.elapsed_time(end) / 1000.0
            else:
                self.results.info.update_time = 0.0
        return None

    def update_affine(self, p, g_map, lu_map, l0, u0):
        """Parametric update for linear MPC, evaluated on the device in one pass (C-ABI rqp_update_affine):
        ``g = p @ g_map.T``, ``l = l0 + p @ lu_map.T``, ``u = u0 + p @ lu_map.T`` with p [batch, np] (the current
        states x0), g_map [n, np], lu_map [m, np], l0/u0 [m].  Equivalent to ``update(g=..., l=..., u=...)`` with those
        vectors (the x0 update of the reference's RandomLinMPC driver); ``self.QP.g/l/u`` are not refreshed."""
        self._need_setup()
        qp, st = self.QP, self.settings
        if not qp.batched:
            raise ValueError("update_affine needs a batched problem")
        p = self._to_dev(p, (qp.batch, _as_tensor(p).shape[-1]), "p")
        npar = p.shape[-1]
        g_map = self._to_dev(g_map, (qp.nx, npar), "g_map")
        lu_map = self._to_dev(lu_map, (qp.nc, npar), "lu_map")
        l0 = self._to_dev(l0, (qp.nc,), "l0")
        u0 = self._to_dev(u0, (qp.nc,), "u0")
        with torch.cuda.device(st.device):
            if self.synchronous:
                start, end = self._events()
                start.record()
            _cabi.check(self._h, _cabi.load().rqp_update_affine(self._h, _cabi.ptr(p), int(npar), _cabi.ptr(g_map),
                                                                  _cabi.ptr(lu_map), _cabi.ptr(l0), _cabi.ptr(u0),
                                                                  self._stream()), "rqp_update_affine")
            if self.synchronous:
                end.record()
                end.synchronize()
                self.results.info.update_time = start.elapsed_time(end) / 1000.0
            else:
                self.results.info.update_time = 0.0
        return None

    def update_settings(self, **kwargs):
        """
        Update ReLU-QP solver settings

        It is possible to change: 'max_iter', 'eps_abs', 'verbose', 'check_interval'
        (reference reluqpth.py:185-199; the whitelist typo "eps_ab" is tolerated, Q8)
        """
        if self._shards:
            return self._shards.update_settings(**kwargs)
        self._need_setup()
        for key, value in kwargs.items():
            if key == "eps_ab":
                key = "eps_abs"
            if key in _CHANGEABLE:
                setattr(self.settings, key, value)
            elif key in _FROZEN:
                raise ValueError("Cannot change {} after setup".format(key))
            else:
                raise ValueError("Invalid setting: {}".format(key))
        cs = self._csettings()
        _cabi.check(self._h, _cabi.load().rqp_update_settings(self._h, ctypes.byref(cs)), "rqp_update_settings")

    # -------------------------------------------------------------------- solve
    def solve(self):
        """
        Solve QP Problem (reference reluqpth.py:201-249 + update_results :278-305):
        one kernel launch for the whole batch, synchronised before returning.
        """
        if self._shards:
            return self._solve_shards()
        self._need_setup()
        lib = _cabi.load()
        st, qp = self.settings, self.QP
        dev, B, n, m = st.device, qp.batch, qp.nx, qp.nc
        # (host cost of a call matters at small batches -- tools/solve_overhead.py: the device guard is entered only when
        #  another device is current, and the current stream is looked up once for the launch and both event records)
        guard = torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else contextlib.nullcontext()
        with guard:
            stream = torch.cuda.current_stream(dev)
            timed = self.synchronous
            xzl = torch.empty(B * (n + 2 * m), device=dev, dtype=st.precision)      # one allocation, three contiguous views
            x, z, lam = xzl[:B * n].view(B, n), xzl[B * n:B * (n + m)].view(B, m), xzl[B * (n + m):].view(B, m)
            ints = torch.empty(3, B, device=dev, dtype=torch.int32)
            dbls = torch.empty(4, B, device=dev, dtype=torch.float64)
            if getattr(self, "prefill_outputs", False):   # test hook: an instance the launch skipped cannot pass with the values a
                xzl.fill_(float("nan"))                   # previous solve left in the allocator's block
                ints.fill_(-7)
                dbls.fill_(float("nan"))
            trace, cap = None, 0
            if st.verbose or getattr(self, "collect_trace", False):
                cap = max(1, st.max_iter // st.check_interval)
                trace = torch.full((B, cap, 4), float("nan"), device=dev, dtype=torch.float64)
            ci = _cabi.CInfo(iter=ints[0].data_ptr(), status=ints[1].data_ptr(), rho_ind=ints[2].data_ptr(),
                             pri_res=dbls[0].data_ptr(), dua_res=dbls[1].data_ptr(),
                             rho_estimate=dbls[2].data_ptr(), obj_val=dbls[3].data_ptr(),
                             trace=trace.data_ptr() if trace is not None else None, trace_cap=cap, reserved=0)
            if timed:                    # one pair of HIP events around the launch: run_time = what the device spent
                k0, k1 = self._events()
                k0.record(stream)
            _cabi.check(self._h, lib.rqp_solve(self._h, _cabi.ptr(x), _cabi.ptr(z), _cabi.ptr(lam),
                                               ctypes.byref(ci), ctypes.c_void_p(stream.cuda_stream)), "rqp_solve")
            if timed:
                k1.record(stream)
                k1.synchronize()
                run_time = k0.elapsed_time(k1) / 1000.0
                self.last_kernel_time = run_time                       # the ADMM launch (+ its order / un-scaling passes)
            else:                        # enqueue only: results are device tensors ordered on the current stream
                run_time = 0.0
                self.last_kernel_time = None
        self.last_trace = trace      # [batch][checks][pri, dua, rho_estimate, rho_ind before the move]
        if st.verbose:
            self._print_trace(trace)
        self._update_results(x, z, lam, ints, dbls, run_time)
        return self.results

    def _print_trace(self, trace):
        tr = trace[0].cpu().numpy()      # instance 0, same line format as reluqpth.py:230
        for c in range(tr.shape[0]):
            if np.isnan(tr[c, 3]):
                break
            print('Iter: {}, rho: {:.2e}, res_p: {:.2e}, res_d: {:.2e}'.format(
                (c + 1) * self.settings.check_interval, tr[c, 2], tr[c, 0], tr[c, 1]))

    def _update_results(self, x, z, lam, ints, dbls, run_time):
        """Update results and info (reference reluqpth.py:278-305)."""
        qp, prec = self.QP, self.settings.precision
        info = self.results.info
        if qp.batched:
            self.results.x, self.results.z, self.results.y = x, z, lam
            info.iter = ints[0]
            info.status_code = ints[1]
            info.status = None            # materialised lazily from status_code (classes.Info.status)
            info.rho_ind = ints[2]
            # per-instance scalars of a batch stay float64 as the kernels wrote them (no cast launch per solve)
            info.pri_res, info.dua_res = dbls[0], dbls[1]
            info.rho_estimate, info.obj_val = dbls[2], dbls[3]
            self.rho_ind = ints[2]
        else:
            self.results.x, self.results.z, self.results.y = x[0], z[0], lam[0]
            ih = ints[:, 0].cpu().tolist()
            info.iter = int(ih[0])
            info.status_code = int(ih[1])
            info.status = _cabi.STATUS_STR[int(ih[1])]
            info.rho_ind = int(ih[2])
            info.pri_res, info.dua_res = dbls[0, 0].to(prec), dbls[1, 0].to(prec)
            info.rho_estimate, info.obj_val = dbls[2, 0].to(prec), dbls[3, 0].to(prec)
            self.rho_ind = int(ih[2]) if self.settings.warm_starting else self._rho_ind0()
        self.results.lam = self.results.y
        self.x, self.z, self.lam = self.results.x, self.results.z, self.results.y
        info.run_time = run_time
        info.solve_time = info.update_time + run_time

    def _solve_shards(self):
        """devices=[...]: every shard's launch is enqueued on its own device / stream, then gathered on devices[0]."""
        import time
        t0 = time.perf_counter()
        o = self._shards.solve()
        run_time = time.perf_counter() - t0
        info = self.results.info
        self.results.x, self.results.z, self.results.y = o["x"], o["z"], o["y"]
        self.results.lam = self.results.y
        info.iter, info.status_code, info.status, info.rho_ind = o["it"], o["sc"], None, o["ri"]
        # per-instance scalars stay float64 as the kernels wrote them -- the same dtypes as the single-device batch path
        info.pri_res, info.dua_res = o["pri"], o["dua"]
        info.rho_estimate, info.obj_val = o["rho"], o["obj"]
        self.rho_ind = o["ri"]
        self.x, self.z, self.lam = self.results.x, self.results.z, self.results.y
        info.run_time = run_time
        info.solve_time = info.update_time + run_time
        return self.results

    def _rho_ind0(self):
        return int(np.argmin(np.abs(self._rhos.cpu().numpy() - self.settings.rho)))

    # --------------------------------------------------------------- warm start
    def warm_start(self, x=None, z=None, lam=None, rho=None):
        """
        Warm start primal or dual variables, lagrange multipliers, and rho
        (reference reluqpth.py:251-276; values are written into the iterate, Q6 fixed)
        """
        if self._shards:
            return self._shards.warm_start(x=x, z=z, lam=lam, rho=rho)
        self._need_setup()
        qp = self.QP
        lead = (qp.batch,) if qp.batched else ()
        xt = None if x is None else self._to_dev(x, lead + (qp.nx,), "x")
        zt = None if z is None else self._to_dev(z, lead + (qp.nc,), "z")
        lt = None if lam is None else self._to_dev(lam, lead + (qp.nc,), "lam")
        with torch.cuda.device(self.settings.device):
            _cabi.check(self._h, _cabi.load().rqp_warm_start(
                self._h, _cabi.ptr(xt), _cabi.ptr(zt), _cabi.ptr(lt), int(rho is not None),
                float(rho) if rho is not None else 0.0, self._stream()), "rqp_warm_start")
            torch.cuda.current_stream(self.settings.device).synchronize()
        if rho is not None:
            self.rho_ind = int(np.argmin(np.abs(self._rhos.cpu().numpy() - rho)))
        return None

    def clear_primal_dual(self):
        """
        Clear primal and dual variables and reset rho index (reference reluqpth.py:324-333)
        """
        if self._shards:
            return self._shards.clear_primal_dual()
        self._need_setup()
        with torch.cuda.device(self.settings.device):
            _cabi.check(self._h, _cabi.load().rqp_clear_primal_dual(self._h, self._stream()),
                        "rqp_clear_primal_dual")
        self.rho_ind = self._rho_ind0()
        return None

    # ---------------------------------------------------- test / inspection hooks
    def dispatch_history(self, mode):
        """Longest-first dispatch from the previous solve's iteration counts (C-ABI rqp_dispatch_history): True / 1 = on
        (default), False / 0 = off (every launch in grid order, like the first solve of a fresh handle), 2 = forget now."""
        self._need_setup()
        _cabi.check(self._h, _cabi.load().rqp_dispatch_history(self._h, int(mode)), "rqp_dispatch_history")

    def get_window(self):
        """(slots, wbase): K(rho) slots per matrix and, on a windowed handle, the int32 tensor [batch] of the ladder index of
        slot 0 of every instance's window (None when the whole ladder is built)."""
        self._need_setup()
        st = self.settings
        with torch.cuda.device(st.device):
            wb = torch.full((self.QP.batch,), -1, device=st.device, dtype=torch.int32)
            slots = ctypes.c_int32(0)
            _cabi.check(self._h, _cabi.load().rqp_get_window(self._h, ctypes.byref(slots), wb.data_ptr(), self._stream()),
                        "rqp_get_window")
            torch.cuda.current_stream(st.device).synchronize()
        return slots.value, (wb if slots.value < len(self._rhos) else None)

    def get_dispatch(self):
        """(order, last_iter) int32 tensors [batch] the next launch would be issued by, or None when no order is recorded."""
        self._need_setup()
        st, B = self.settings, self.QP.batch
        with torch.cuda.device(st.device):
            out = torch.full((2, B), -1, device=st.device, dtype=torch.int32)
            valid = ctypes.c_int32(0)
            _cabi.check(self._h, _cabi.load().rqp_get_dispatch(self._h, out[0].data_ptr(), out[1].data_ptr(),
                                                               ctypes.byref(valid), self._stream()), "rqp_get_dispatch")
            torch.cuda.current_stream(st.device).synchronize()
        return (out[0], out[1]) if valid.value else None

    def iterate(self, k):
        """k plain ADMM iterations (ReLU_Layer.forward, reluqpth.py:80-89), no checks."""
        self._need_setup()
        with torch.cuda.device(self.settings.device):
            _cabi.check(self._h, _cabi.load().rqp_iterate(self._h, int(k), self._stream()), "rqp_iterate")
        return self.get_state()[0]

    def get_state(self):
        """(output, rho_ind): ``output`` = [x; z; lam] as the reference's ``ReLU_QP.output``."""
        if self._shards:
            return self._shards.get_state()
        self._need_setup()
        st, qp = self.settings, self.QP
        B, n, m = qp.batch, qp.nx, qp.nc
        with torch.cuda.device(st.device):
            x = torch.empty(B, n, device=st.device, dtype=st.precision)
            z = torch.empty(B, m, device=st.device, dtype=st.precision)
            lam = torch.empty(B, m, device=st.device, dtype=st.precision)
            ri = torch.empty(B, device=st.device, dtype=torch.int32)
            _cabi.check(self._h, _cabi.load().rqp_get_state(self._h, _cabi.ptr(x), _cabi.ptr(z), _cabi.ptr(lam),
                                                            _cabi.ptr(ri), self._stream()), "rqp_get_state")
            torch.cuda.current_stream(st.device).synchronize()
        out = torch.cat([x, z, lam], dim=1)
        return (out, ri) if qp.batched else (out[0], int(ri[0]))

    @property
    def output(self):
        return self.get_state()[0]

    def compute_residuals(self, rho):
        """(primal_res, dual_res, rho_estimate, obj) of the current state for a carried
        estimate ``rho`` (reference compute_residuals/compute_J, reluqpth.py:307-322)."""
        self._need_setup()
        st, B = self.settings, self.QP.batch
        with torch.cuda.device(st.device):
            out = torch.empty(4, B, device=st.device, dtype=torch.float64)
            _cabi.check(self._h, _cabi.load().rqp_compute_residuals(
                self._h, float(rho), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                self._stream()), "rqp_compute_residuals")
            torch.cuda.current_stream(st.device).synchronize()
        if self.QP.batched:
            return out[0], out[1], out[2], out[3]
        return out[0, 0], out[1, 0], out[2, 0], out[3, 0]

    def _need_setup(self):
        if self._shards:
            raise RuntimeError("not available with devices=[...]: call it on a shard (model._shards.children[i])")
        if self._h is None:
            raise RuntimeError("ReLU_QP.setup() must be called first")


if __name__ == "__main__":
    # the reference's inline known-answer test (reluqpth.py:338-367)
    H = torch.tensor([[6, 2, 1], [2, 5, 2], [1, 2, 4.0]], dtype=torch.double)
    g = torch.tensor([-8.0, -3, -3], dtype=torch.double)
    A = torch.tensor([[1, 0, 1], [0, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=torch.double)
    l = torch.tensor([3.0, 0, -10.0, -10, -10], dtype=torch.double)
    u = torch.tensor([3.0, 0, torch.inf, torch.inf, torch.inf], dtype=torch.double)
    qp = ReLU_QP()
    qp.setup(H=H, g=g, A=A, l=l, u=u)
    results = qp.solve()
    assert torch.allclose(results.x.cpu(), torch.tensor([2.0, -1, 1], dtype=torch.float64))
    print("Test passed!")
    print(results.x, results.info.solve_time, results.info.setup_time, results.info.iter, results.info.status)
