"""In-process batch split over several GPUs: ``ReLU_QP.setup(..., devices=[0, 1, ...])``.

QP instances of a batch are independent (no coupling term in the reference's solve loop, reluqpth.py:201-249), so the
batch shards contiguously (``distributed.shard_range``): device ``devices[i]`` owns one C-ABI handle, its slice of
(g, l, u[, H, A]) and its own HIP stream; ``solve()`` enqueues every shard's launch, then waits for all of them and
gathers the results on ``devices[0]``.  No collective, no peer access during the solve -- one device-to-device copy of
each shard's outputs at the end.  (``torch.distributed`` + one process per GPU is the other way to do the same thing:
``reluqp.distributed`` / ``bench.py --gpus N``.)  A device may appear several times (two shards on one GPU run on two
streams), which is also how the path is tested on a one-GPU box.
"""
from concurrent.futures import ThreadPoolExecutor

import torch

from reluqp import _cabi
from reluqp.classes import _as_tensor
from reluqp.distributed import shard_range


class DeviceShards(object):
    def __init__(self, owner_cls, devices, H, g, A, l, u, setup_kw):
        if not devices:
            raise ValueError("devices must list at least one device")
        H, g, A, l, u = (_as_tensor(t) for t in (H, g, A, l, u))
        if g.dim() != 2:
            raise ValueError("devices=[...] needs a batched problem (g, l, u with a leading batch dimension)")
        self.devices = [torch.device("cuda", int(d)) if not isinstance(d, torch.device) else d for d in devices]
        self.batch = g.shape[0]
        world = len(self.devices)
        if self.batch < world:
            raise ValueError("batch of %d cannot be split over %d devices" % (self.batch, world))
        self.ranges = [shard_range(self.batch, r, world) for r in range(world)]
        shared = H.dim() == 2
        self.children = [owner_cls() for _ in self.devices]
        self.streams = []
        for child, dev in zip(self.children, self.devices):
            child.synchronous = False                       # enqueue only; this class waits once for all shards
            with torch.cuda.device(dev):
                self.streams.append(torch.cuda.Stream(device=dev))
        # One host thread per shard ("one host thread (or process) per GPU to avoid serialising launches"):
        # every shard's setup chain is enqueued before anything is waited for, and a call that blocks inside the library
        # (the count read-back of a windowed rqp_solve) only blocks its own thread -- ctypes releases the GIL for the call.
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices)) if len(self.devices) > 1 else None
        self._each(lambda c, sl: c.setup(H if shared else H[sl], g[sl], A if shared else A[sl], l[sl], u[sl],
                                         device=c_dev(c, self), **setup_kw))
        self._wait()
        for c in self.children:
            c._finish_setup_time()
        self.kernel = ",".join(sorted(set(c.kernel for c in self.children)))

    # ---- helpers
    def _each(self, fn):
        def run(child, dev, stream, rng):
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                return fn(child, slice(rng[0], rng[0] + rng[1]))
        jobs = list(zip(self.children, self.devices, self.streams, self.ranges))
        if self._pool is None:
            return [run(*j) for j in jobs]
        return [f.result() for f in [self._pool.submit(run, *j) for j in jobs]]

    def _wait(self):
        for stream in self.streams:
            stream.synchronize()

    def _gather(self, parts, dim=0):
        d0 = self.devices[0]
        return torch.cat([p.to(d0, non_blocking=True) for p in parts], dim=dim)

    @staticmethod
    def _sl(a, sl):
        return None if a is None else _as_tensor(a)[sl]

    # ---- the solver surface
    def solve(self):
        res = self._each(lambda c, sl: c.solve())
        # every child returns its own Results object: read the tensors before anything else is enqueued
        parts = [dict(x=r.x, z=r.z, y=r.y, it=r.info.iter, sc=r.info.status_code, ri=r.info.rho_ind, pri=r.info.pri_res,
                      dua=r.info.dua_res, rho=r.info.rho_estimate, obj=r.info.obj_val) for r in res]
        self._wait()
        out = {k: self._gather([p[k] for p in parts]) for k in parts[0]}
        torch.cuda.current_stream(self.devices[0]).synchronize()
        return out

    def update(self, g=None, l=None, u=None, Hx=None, Ax=None):
        def fn(c, sl):
            shared = c.QP.shared_mats
            c.update(g=self._sl(g, sl), l=self._sl(l, sl), u=self._sl(u, sl),
                     Hx=Hx if (Hx is None or shared) else _as_tensor(Hx)[sl],
                     Ax=Ax if (Ax is None or shared) else _as_tensor(Ax)[sl])
        self._each(fn)
        self._wait()

    def warm_start(self, x=None, z=None, lam=None, rho=None):
        self._each(lambda c, sl: c.warm_start(x=self._sl(x, sl), z=self._sl(z, sl), lam=self._sl(lam, sl), rho=rho))
        self._wait()

    def clear_primal_dual(self):
        self._each(lambda c, sl: c.clear_primal_dual())
        self._wait()

    def update_settings(self, **kwargs):
        self._each(lambda c, sl: c.update_settings(**kwargs))

    def get_state(self):
        parts = self._each(lambda c, sl: c.get_state())
        self._wait()
        return self._gather([p[0] for p in parts]), self._gather([p[1] for p in parts])

    def destroy(self):
        for c in self.children:
            c._destroy()
        self.children = []
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=False)
            self._pool = None


def c_dev(child, shards):
    return shards.devices[shards.children.index(child)]


def status_strings(codes):
    return [_cabi.STATUS_STR[int(c)] for c in codes.cpu().tolist()]
