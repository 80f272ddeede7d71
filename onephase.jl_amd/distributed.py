"""Multi-GPU orchestration of ONE KKT factorisation + solve: elimination-tree subtrees sharded over ranks
(SURVEY.md section 8e).  The numeric work and the pack/unpack kernels live behind the C ABI
(include/okkt.h, okkt_dist_*); this module only sequences the phases and moves three flat buffers
(contribution blocks of the cut, contribution vectors of the cut, solution pieces) between ranks.

Two communicators:
  * TorchComm  -- one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo"
                  in the CPU tests): reduce(sum) to part 0, broadcast from part 0, all-reduce of 4 integers.
  * LocalComm  -- several parts driven from ONE process (virtual ranks); used by the single-GPU parity test
                  to run the exact sharded code path and compare it with the unsharded factorisation.
There is no reference counterpart: the reference is single-process.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .linear_system_solvers import OkktError, csc_arrays, finalize_b, initialize_b, linear_solver_HIP


class LocalComm:
    """nparts virtual ranks in one process; buffers live in HBM of the solver's device, collectives go
    through the host (numpy) -- a test vehicle, not a performance path."""

    def __init__(self, nparts):
        self.world = nparts
        self.ranks = list(range(nparts))

    def alloc(self, solver, ndoubles):
        return _DevBuf(solver, ndoubles)

    def reduce_sum(self, bufs, dst=0):
        tot = sum(b.download() for b in bufs)
        bufs[dst].upload(tot)

    def broadcast(self, bufs, src=0):
        v = bufs[src].download()
        for i, b in enumerate(bufs):
            if i != src:
                b.upload(v)

    def allreduce_counts(self, counts):
        tot = np.sum(np.stack(counts), axis=0)
        return [tot.copy() for _ in counts]


class _DevBuf:
    def __init__(self, solver, n):
        self.solver, self.n = solver, max(int(n), 1)
        self.ptr = solver.dev_alloc(8 * self.n)
        self.zero()

    def zero(self):
        self.upload(np.zeros(self.n))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self.solver._check(self.solver._lib.okkt_dev_upload(self.solver._h, C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes), "upload")

    def download(self):
        return self.solver.dev_download(self.ptr, (self.n,))

    def free(self):
        self.solver.dev_free(self.ptr)


class TorchComm:
    """One process per GPU.  Buffers are torch tensors (their data_ptr() is what the C ABI receives)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size()
        self.ranks = [dist.get_rank()]
        self.device = device

    def alloc(self, solver, ndoubles):
        return _TorchBuf(self.torch, max(int(ndoubles), 1), self.device)

    def _fence(self):
        # the library works on its own HIP stream and is synchronous at every C-ABI call; RCCL collectives
        # are only enqueued on torch's stream -- wait for them before the library touches the buffer
        if self.device is not None and str(self.device) != "cpu":
            self.torch.cuda.synchronize()

    def reduce_sum(self, bufs, dst=0):
        self.dist.reduce(bufs[0].t, dst=dst, op=self.dist.ReduceOp.SUM)
        self._fence()

    def broadcast(self, bufs, src=0):
        self.dist.broadcast(bufs[0].t, src=src)
        self._fence()

    def allreduce_counts(self, counts):
        t = self.torch.tensor(np.asarray(counts[0], dtype=np.int64), device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [t.cpu().numpy()]


class _TorchBuf:
    def __init__(self, torch, n, device):
        self.t = torch.zeros(n, dtype=torch.float64, device=device)
        self.ptr = self.t.data_ptr()
        self.n = n

    def zero(self):
        self.t.zero_()

    def download(self):
        return self.t.cpu().numpy()

    def free(self):
        pass


class ShardedLinearSolver:
    """ls_factor! / ls_solve of ONE matrix with its elimination tree sharded over `comm.world` parts."""

    def __init__(self, comm, sym="symmetric", **opts):
        self.comm = comm
        self.sym = sym
        self.solvers = []
        for r in comm.ranks:
            s = linear_solver_HIP(sym, **opts)
            initialize_b(s)
            self.solvers.append(s)
        self._bufs = None
        self.inertia = None

    def _call(self, s, fn, *args):
        return s._check(getattr(s._lib, fn)(s._h, *args), fn)

    def analyze(self, A):
        self.dim, colptr, rowval, self._nzval, base = csc_arrays(A)
        for s, r in zip(self.solvers, self.comm.ranks):
            s._check(s._lib.okkt_analyze(s._h, self.dim, L.p_i64(colptr), L.p_i64(rowval), base), "okkt_analyze")
            s._dim = self.dim
            self._call(s, "okkt_dist_set_partition", self.comm.world, r)
        s0 = self.solvers[0]
        cb, cv, nb = C.c_int64(), C.c_int64(), C.c_int64()
        pf = np.zeros(self.comm.world)
        tf = C.c_double()
        self._call(s0, "okkt_dist_info", C.byref(cb), C.byref(cv), C.byref(nb), L.p_f64(pf), C.byref(tf))
        self.info = dict(cb_doubles=cb.value, cv_doubles=cv.value, n_boundary=nb.value, part_flops=pf.tolist(), top_flops=tf.value)
        if self._bufs:
            for group in self._bufs.values():
                for b in group:
                    b.free()
        mk = lambda n: [self.comm.alloc(s, n) for s in self.solvers]
        self._bufs = dict(cb=mk(cb.value), cv=mk(cv.value), x=mk(self.dim), sol=mk(self.dim))
        return self.info

    def owners(self):
        s = self.solvers[0]
        st = s.stats()
        sn = np.zeros(st["nsuper"], dtype=np.int64)
        col = np.zeros(self.dim, dtype=np.int64)
        par = np.zeros(st["nsuper"], dtype=np.int64)
        self._call(s, "okkt_dist_get_owner", L.p_i64(sn), L.p_i64(col), L.p_i64(par))
        return sn, col, par

    def _timed(self, timings, key, rank, s, fn, *args):
        """One blocking C-ABI phase; with a timings dict its wall-clock (ms, the call synchronises the handle's stream) is
        appended to timings[key][rank] -- the measured components of the multi-GPU prediction (scripts/sharded_model.py)."""
        if timings is None:
            return self._call(s, fn, *args)
        import time
        t = time.perf_counter()
        rc = self._call(s, fn, *args)
        timings.setdefault(key, {}).setdefault(rank, []).append(1e3 * (time.perf_counter() - t))
        return rc

    def factor(self, d_vals_per_rank, n, m, timings=None):
        """d_vals_per_rank: device pointers of nzval, one per local rank.  Returns the 1/0 inertia flag."""
        kind = L.OKKT_SYM_DEFINITE if self.sym == "definite" else L.OKKT_SYM_SYMMETRIC
        cbs = self._bufs["cb"]
        for b in cbs:
            b.zero()
        for s, dv, r in zip(self.solvers, d_vals_per_rank, self.comm.ranks):
            self._timed(timings, "factor_local", r, s, "okkt_dist_factor_local", C.c_void_p(dv), n, m, kind)
        for s, b, r in zip(self.solvers, cbs, self.comm.ranks):
            self._timed(timings, "pack_cb", r, s, "okkt_dist_cb", C.c_void_p(b.ptr), 0)
        self.comm.reduce_sum(cbs, dst=0)                      # RCCL reduce of the parent-front contribution blocks
        for s, b, r in zip(self.solvers, cbs, self.comm.ranks):
            if r == 0:
                self._timed(timings, "unpack_cb", r, s, "okkt_dist_cb", C.c_void_p(b.ptr), 1)
                self._timed(timings, "factor_top", r, s, "okkt_dist_factor_top")
        counts = []
        for s in self.solvers:
            c = np.zeros(4, dtype=np.int64)
            self._call(s, "okkt_dist_counts", L.p_i64(c))
            counts.append(c)
        totals = self.comm.allreduce_counts(counts)
        flag = None
        for s, t in zip(self.solvers, totals):
            t = np.ascontiguousarray(t, dtype=np.int64)
            flag = self._call(s, "okkt_dist_finish", L.p_i64(t))
            self.inertia = tuple(int(v) for v in t)
        return int(flag)

    def solve(self, d_rhs_per_rank, timings=None):
        """Returns the solution (original ordering) as a numpy array on every rank that owns buffer 0."""
        cvs, xs, sols = self._bufs["cv"], self._bufs["x"], self._bufs["sol"]
        for b in cvs:
            b.zero()
        for s, dr, r in zip(self.solvers, d_rhs_per_rank, self.comm.ranks):
            self._timed(timings, "solve_fwd_local", r, s, "okkt_dist_solve_begin", C.c_void_p(dr))
        for s, b in zip(self.solvers, cvs):
            self._call(s, "okkt_dist_cv", C.c_void_p(b.ptr), 0)
        self.comm.reduce_sum(cvs, dst=0)
        for s, b, xb, r in zip(self.solvers, cvs, xs, self.comm.ranks):
            if r == 0:
                self._call(s, "okkt_dist_cv", C.c_void_p(b.ptr), 1)
                self._timed(timings, "solve_top", r, s, "okkt_dist_solve_top")
                self._call(s, "okkt_dist_x", C.c_void_p(xb.ptr), 0)
        self.comm.broadcast(xs, src=0)                        # separator solution to every part
        for s, xb, sb, r in zip(self.solvers, xs, sols, self.comm.ranks):
            self._call(s, "okkt_dist_x", C.c_void_p(xb.ptr), 1)
            self._timed(timings, "solve_bwd_local", r, s, "okkt_dist_solve_end")
            self._call(s, "okkt_dist_x", C.c_void_p(sb.ptr), 2)
        self.comm.reduce_sum(sols, dst=0)
        return sols[0].download()

    def finalize(self):
        for s in self.solvers:
            finalize_b(s)
        self.solvers = []


class RcclShardedLinearSolver:
    """ls_factor! / ls_solve of ONE matrix sharded over `world` processes (one per GPU) with the collectives INSIDE the
    library on RCCL (okkt_dist_factor / okkt_dist_solve: reduce of the contribution blocks, broadcast of the separator
    solution, all-reduce of the solution pieces on the handle's stream, one synchronisation per call).  This is the path a
    Julia host uses; Python only ships the 128-byte RCCL id from rank 0 to the others -- `exchange_id(bytes or None) ->
    bytes` (default: torch.distributed.broadcast_object_list when a process group exists; world == 1 needs none)."""

    def __init__(self, rank, world, sym="symmetric", exchange_id=None, **opts):
        self.rank, self.world, self.sym = int(rank), int(world), sym
        self._exchange = exchange_id
        self.solver = linear_solver_HIP(sym, **opts)
        initialize_b(self.solver)
        self.inertia = None
        self._ready = False

    def _call(self, fn, *args):
        s = self.solver
        return s._check(getattr(s._lib, fn)(s._h, *args), fn)

    def analyze(self, A):
        s = self.solver
        self.dim, colptr, rowval, _, base = csc_arrays(A)
        s._check(s._lib.okkt_analyze(s._h, self.dim, L.p_i64(colptr), L.p_i64(rowval), base), "okkt_analyze")
        s._dim = self.dim
        self._call("okkt_dist_set_partition", self.world, self.rank)
        uid = None
        if self.rank == 0:
            buf = (C.c_char * 128)()
            rc = s._lib.okkt_dist_unique_id(C.cast(buf, C.c_void_p))
            if rc != 0:
                raise OkktError(f"okkt_dist_unique_id failed ({rc}): librccl could not be opened")
            uid = bytes(buf)
        if self.world > 1:
            ex = self._exchange
            if ex is None:
                import torch.distributed as dist
                def ex(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            uid = ex(uid)
        idbuf = C.create_string_buffer(uid, 128)
        self._call("okkt_dist_comm_init", self.world, self.rank, C.cast(idbuf, C.c_void_p))
        cb, cv, nb = C.c_int64(), C.c_int64(), C.c_int64()
        pf = np.zeros(self.world)
        tf = C.c_double()
        self._call("okkt_dist_info", C.byref(cb), C.byref(cv), C.byref(nb), L.p_f64(pf), C.byref(tf))
        self.info = dict(cb_doubles=cb.value, cv_doubles=cv.value, n_boundary=nb.value, part_flops=pf.tolist(), top_flops=tf.value)
        self._ready = True
        return self.info

    def factor(self, d_vals, n, m):
        kind = L.OKKT_SYM_DEFINITE if self.sym == "definite" else L.OKKT_SYM_SYMMETRIC
        inert = L.OkktInertia()
        flag = self._call("okkt_dist_factor", C.c_void_p(d_vals), n, m, kind, C.byref(inert))
        self.inertia = inert.as_tuple()
        return int(flag)

    def solve(self, d_rhs, d_sol):
        self._call("okkt_dist_solve", C.c_void_p(d_rhs), C.c_void_p(d_sol))

    def finalize(self):
        if self.solver is not None:
            self.solver._lib.okkt_dist_comm_destroy(self.solver._h)
            finalize_b(self.solver)
            self.solver = None


# ---------------------------------------------------------------------------------------------------------------
# Speculative delta loop: replicas of ONE KKT system, every rank factors a different candidate of
# ipopt_strategy!'s delta sequence at the same time.  The candidates, their order and the returned
# (status, num_fac, delta) are exactly those of the serial loop (delta_strategy.jl:37-114): a round of W ranks costs
# the wall time of one factorisation instead of W.
# ---------------------------------------------------------------------------------------------------------------
def delta_candidates(tau, delta_prev, pars, max_it=500):
    """The deltas ipopt_strategy! would try, in order, generated lazily with the reference's own arithmetic
    (delta = delta * inc, not inc ** k).  The sequence ends after the first delta > delta.max (the serial loop
    returns :failure right after trying it) or after max_it entries of the for-loop."""
    d = pars.delta
    if tau > 0.0:
        tau = 0.0
        yield d.zero
    delta = None
    for i in range(1, max_it + 1):
        if i == 1:
            delta = max(d.min - tau, delta_prev * d.dec) if delta_prev != 0.0 else d.start - tau
        else:
            delta = delta * d.inc
        yield delta
        if delta > d.max:
            return


def _allgather_flags(comm, local_flags):
    """Inertia flags of all ranks in rank order.  LocalComm: the list is already global."""
    if isinstance(comm, LocalComm):
        return list(local_flags)
    t = comm.torch.tensor([int(local_flags[0])], dtype=comm.torch.int64, device=comm.device)
    out = [comm.torch.zeros_like(t) for _ in range(comm.world)]
    comm.dist.all_gather(out, t)
    return [int(v.item()) for v in out]


def speculative_ipopt_strategy(comm, kkt_solvers, it, sync_factor=True):
    """kkt_solvers: one formed KKT solver per LOCAL rank (comm.ranks), all holding the same system.
    Returns (status, num_fac, delta, owner_rank) identical on every rank; the owner's solver holds the successful
    factor, the others refactor with the chosen delta when sync_factor is set (in parallel: one more factorisation
    of wall time), so that every rank can go on to compute_direction!."""
    W = comm.world
    pars = kkt_solvers[0].pars
    tau = 1.5 * kkt_solvers[0].diag_min()
    gen = delta_candidates(tau, float(it.delta), pars)
    tried = 0
    while True:
        cands = []
        for _ in range(W):
            try:
                cands.append(next(gen))
            except StopIteration:
                break
        if not cands:
            raise OkktError("max it")                       # error("max it"), delta_strategy.jl:113
        local = []
        for k, r in zip(kkt_solvers, comm.ranks):
            local.append(k.factor_b(cands[r], trial=True) if r < len(cands) else -1)    # a candidate that fails is discarded
        flags = _allgather_flags(comm, local)
        for j, d in enumerate(cands):
            if flags[j] == 1:
                status, num_fac, delta, owner = "success", tried + j + 1, d, j
                break
            if d > pars.delta.max:
                status, num_fac, delta, owner = "failure", tried + j + 1, d, j
                break
        else:
            tried += len(cands)
            continue
        if sync_factor:
            for k, r in zip(kkt_solvers, comm.ranks):
                if r != owner:
                    k.factor_b(delta)
        return status, num_fac, delta, owner
