"""The task queue of the dataflow launch (csrc/dataflow_sched.cpp, run by csrc/dataflow.hip), checked on the CPU.

The queue is replayed IN ORDER on dense matrices with numpy: every task must find the tile states it waits for already
published (the persistent kernel pops the queue in order, so a dependency later in the queue would be a deadlock), every
tile must receive its panels in ascending order exactly once, and the result must be the partial LDL^T of the front
(the arithmetic the reference reaches through CHOLMOD, src/linear_system_solvers/julia.jl:34,52)."""
import ctypes as C

import numpy as np
import pytest

from onephase_jl_amd import _lib

D, T, U, TU, TA, TL = 0, 1, 2, 3, 4, 5


def build_queue(fronts, workers=256, group=2, rows=1, fuse_d=False, split=False, fuse_tl=False, lock=False):
    lib = _lib.load()
    n = len(fronts)
    f = (C.c_int32 * n)(*[a for a, _ in fronts])
    k = (C.c_int32 * n)(*[b for _, b in fronts])
    model = C.c_double(0)
    group = group | (rows << 8) | ((1 if fuse_d else 0) << 16) | ((1 if split else 0) << 17) | ((1 if fuse_tl else 0) << 18) | ((1 if lock else 0) << 19)
    cnt = lib.okkt_debug_dataflow_queue(n, f, k, workers, group, None, 0, C.byref(model))
    assert cnt >= 0
    buf = (C.c_int32 * (4 * max(cnt, 1)))()
    assert lib.okkt_debug_dataflow_queue(n, f, k, workers, group, buf, cnt, C.byref(model)) == cnt
    q = np.frombuffer(buf, dtype=np.int32).reshape(-1, 4)[:cnt]
    tasks = [(int(a), int(b) & 255, (int(b) >> 8) & 255, int(c) & 0xFFFF, int(c) >> 16, int(d), max(int(b) >> 16, 1)) for a, b, c, d in q]
    return tasks, model.value


def bounds(f, k):
    KB = (k + 127) // 128
    b = [min(128 * x, k) for x in range(KB)] + [k]
    while b[-1] < f:
        b.append(min(b[-1] + 128, f))
    if b[-1] != f or (len(b) >= 2 and b[-2] == b[-1]):
        b = sorted(set(b))
    return KB, b


def dense_partial_ldlt(A, k):
    A = A.copy()
    n = A.shape[0]
    for j in range(k):
        d = A[j, j]
        l = A[j + 1:, j] / d
        A[j + 1:, j + 1:] -= np.outer(l, A[j + 1:, j])
        A[j + 1:, j] = l
    return A


def replay(fronts, tasks, group, fused=False, split=False, fuse_tl=False):
    rng = np.random.default_rng(7)
    halves = set()
    mats, refs, states, Ws, grids = [], [], [], [], []
    for f, k in fronts:
        B = rng.normal(size=(f, f))
        A = B + B.T + np.diag(rng.choice([-1.0, 1.0], size=f) * (4.0 + 2 * np.sqrt(f)))
        mats.append(A.copy())
        refs.append(dense_partial_ldlt(A, k))
        KB, b = bounds(f, k)
        TB = len(b) - 1
        grids.append((KB, TB, b))
        states.append(np.zeros((TB, TB), dtype=int))
        Ws.append(np.zeros((f, k)))
    tl_tiles = {(t[0], t[3], t[4]) for t in tasks if t[1] == TL}

    def ready(task):
        """the tile states the kernels wait for before a popped task starts (dataflow.hip: the worker's poll, and D(q) inside TU / TA)"""
        (a, typ, nq, i0, j, q0, R) = task
        st = states[a]
        if typ == D:
            return st[i0, i0] >= i0
        if typ == T:
            return st[j, j] >= j + 1 and st[i0, j] >= j
        if typ == TL:      # popped on the update's operands; D(q) is awaited inside the task
            return st[i0, j] >= j - 1 and st[i0, j - 1] >= j and st[j, j - 1] >= j and st[j, j] >= j + 1
        if typ in (TU, TA):
            if typ == TU and (nq & 4) and (a, j) not in halves:
                return False
            return st[j, j] >= j + 1 and st[i0, j] >= j and st[i0, i0] >= j
        ql = q0 + nq - 1
        return all(st[i, ql] >= ql + 1 and st[i, j] >= q0 for i in range(i0, i0 + R)) and st[j, ql] >= ql + 1

    def run(task):
      (a, typ, nq, i0, j, q0, R) = task
      assert R == 1 or (typ == U and i0 > j), (typ, i0, j, R)
      for i in range(i0, i0 + R):
          f, k = fronts[a]
          KB, TB, b = grids[a]
          st, A, W = states[a], mats[a], Ws[a]
          ri = slice(b[i], b[i + 1])
          if typ == D:
              assert i == j and i < KB and st[i, i] == i, ("D out of order", a, i, st[i, i])
              assert not (fused and i > 0), "D(q >= 1) must ride in TU(q - 1)"
              blk = A[ri, ri]
              nb = blk.shape[0]
              for c in range(nb):
                  d = blk[c, c]
                  l = blk[c + 1:, c] / d
                  blk[c + 1:, c + 1:] -= np.outer(l, blk[c + 1:, c])
                  blk[c + 1:, c] = l
              st[i, i] = i + 1
          elif typ == TA:
              # the upper 64 rows of block row q + 1: solved, their part of the diagonal tile updated (sub-tiles that need no other row)
              q = j
              assert i == q + 1 and i < KB and st[q, q] >= q + 1 and st[i, q] == q and st[i, i] == q, ("TA out of order", a, i, q)
              assert b[i + 1] - b[i] > 64 and (a, q) not in halves
              ra = slice(b[i], b[i] + 64)
              cq = slice(b[q], b[q + 1])
              Lqq = np.tril(A[cq, cq], -1) + np.eye(b[q + 1] - b[q])
              d = np.diag(A[cq, cq])
              Wt = np.linalg.solve(Lqq, A[ra, cq].T).T
              W[ra, cq] = Wt
              A[ra, cq] = Wt / d
              A[ra, ra] -= np.tril(W[ra, cq] @ A[ra, cq].T)
              halves.add((a, q))
          elif typ in (T, TU, TL):
              q = j
              if typ == TL:      # the last update of the tile first: panel q - 1
                  assert q >= 1 and not (i == q + 1 and q + 1 < KB) and st[i, q] == q - 1 and st[i, q - 1] >= q and st[q, q - 1] >= q, ("TL out of order", a, i, q, st[i, q])
                  kq = slice(b[q - 1], b[q])
                  A[ri, b[q]:b[q + 1]] -= W[ri, kq] @ A[b[q]:b[q + 1], kq].T
                  st[i, q] = q
              assert i > q and q < KB and st[q, q] >= q + 1 and st[i, q] == q, ("T out of order", a, i, q)
              rows_t = ri
              if typ == TU:
                  assert i == q + 1 and i < KB and st[i, i] == q, ("TU out of order", a, i, q, st[i, i])
                  assert bool(nq & 4) == (split and b[i + 1] - b[i] > 64), ("TU split flag", a, i, q, nq)
                  if nq & 4:     # the lower half of a split block row: the upper one must have been popped before
                      assert (a, q) in halves, ("TU before its TA", a, q)
                      rows_t = slice(b[i] + 64, b[i + 1])
              else:
                  assert nq == 1
                  assert not (fuse_tl and typ == T and q >= 1), "T(i, q >= 1) must be a TL task"
              cq = slice(b[q], b[q + 1])
              Lqq = np.tril(A[cq, cq], -1) + np.eye(b[q + 1] - b[q])
              d = np.diag(A[cq, cq])
              Wt = np.linalg.solve(Lqq, A[rows_t, cq].T).T
              W[rows_t, cq] = Wt
              A[rows_t, cq] = Wt / d
              st[i, q] = q + 1
              if typ == TU:      # ... and the diagonal tile of block row i receives panel q
                  if nq & 4:
                      ra = slice(b[i], b[i] + 64)
                      A[rows_t, ra] -= W[rows_t, cq] @ A[ra, cq].T
                      A[rows_t, rows_t] -= np.tril(W[rows_t, cq] @ A[rows_t, cq].T)
                  else:
                      A[ri, ri] -= np.tril(W[ri, cq] @ A[ri, cq].T)
                  st[i, i] = q + 1
                  if nq & 2:     # ... and is factored by the same task (D(q + 1) is not a task of its own)
                      blk = A[ri, ri]
                      for c in range(blk.shape[0]):
                          dd = blk[c, c]
                          l = blk[c + 1:, c] / dd
                          blk[c + 1:, c + 1:] -= np.outer(l, blk[c + 1:, c])
                          blk[c + 1:, c] = l
                      st[i, i] = i + 1
          else:
              ql = q0 + nq - 1
              assert 1 <= nq <= max(group, 1) and i >= j > ql and ql < KB, ("U shape", a, i, j, q0, nq)
              assert not (ql == j - 1 and (a, i, j) in tl_tiles), ("this update belongs inside TL", a, i, j, q0, nq)
              assert not (fuse_tl and j < KB and ql == j - 1 and i > j and not (i == j + 1 and j + 1 < KB)), ("every block row below TU's is fused", a, i, j)
              assert st[i, ql] >= ql + 1 and st[j, ql] >= ql + 1 and st[i, j] == q0, ("U out of order", a, i, j, q0, nq, st[i, j])
              cj = slice(b[j], b[j + 1])
              kk = slice(b[q0], b[ql + 1])
              upd = W[ri, kk] @ A[cj, kk].T
              if i == j:
                  upd = np.tril(upd)
              A[ri, cj] -= upd
              st[i, j] = q0 + nq
    for task in tasks:
        assert ready(task), ("dependency later in the queue", task)
        run(task)
    for a, (f, k) in enumerate(fronts):
        KB, TB, b = grids[a]
        for i in range(TB):
            for j in range(i + 1):
                want = j + 1 if j < KB else KB
                assert states[a][i, j] == want, (a, i, j, states[a][i, j], want)
        got = np.tril(mats[a])
        ref = np.tril(refs[a])
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(got - ref)) <= 1e-9 * scale, (a, np.max(np.abs(got - ref)) / scale)


CASES = [
    [(129, 1)],
    [(300, 128)],
    [(300, 100)],
    [(700, 700)],
    [(700, 333)],
    [(1000, 256), (400, 130), (129, 129)],
    [(1500, 1030)],
]


@pytest.mark.parametrize("fronts", CASES)
@pytest.mark.parametrize("group,rows,fused,split,tl", [(1, 1, False, False, False), (2, 1, True, False, False), (3, 2, False, True, False), (2, 4, True, False, True), (4, 1, True, True, True), (2, 1, False, False, True)])
def test_queue_replays_to_the_partial_factorisation(fronts, group, rows, fused, split, tl):
    tasks, model = build_queue(fronts, workers=16, group=group, rows=rows, fuse_d=fused, split=split, fuse_tl=tl)
    assert model > 0
    replay(fronts, tasks, group, fused, split, fuse_tl=tl)


@pytest.mark.parametrize("fronts", CASES)
@pytest.mark.parametrize("group,rows,fused,tl", [(4, 1, True, True), (2, 2, False, False)])
def test_lockstep_queue_replays_to_the_partial_factorisation(fronts, group, rows, fused, tl):
    # round 6: TU(q) in lockstep with D(q) -- one task per block row (no TA whatever `split` says), flagged with bit 8 of its nq field
    tasks, model = build_queue(fronts, workers=16, group=group, rows=rows, fuse_d=fused, split=True, fuse_tl=tl, lock=True)
    assert model > 0 and not any(t[1] == TA for t in tasks)
    assert all(t[2] & 8 for t in tasks if t[1] == TU)
    replay(fronts, tasks, group, fused, False, fuse_tl=tl)


def test_queue_is_the_same_every_time_and_scales():
    a, _ = build_queue([(2000, 900), (600, 200)], workers=256, group=2)
    b, _ = build_queue([(2000, 900), (600, 200)], workers=256, group=2)
    assert a == b
    # task counts: D = KB, T = sum over panels of the blocks below, U = groups per tile
    f, k = 2000, 900
    KB, bb = bounds(f, k)
    TB = len(bb) - 1
    nD = sum(1 for t in a if t[0] == 0 and t[1] == D)
    nT = sum(1 for t in a if t[0] == 0 and t[1] in (T, TU))
    nTU = sum(1 for t in a if t[0] == 0 and t[1] == TU)
    assert nD == KB and nTU == KB - 1 and nT == sum(TB - 1 - q for q in range(KB))
    assert not any(t[1] == TA for t in a)
    c, _ = build_queue([(2000, 900), (600, 200)], workers=256, group=2, split=True)
    # 900 = 7 * 128 + 4: the last pivot block row has 4 rows and stays whole, the six before it are split; 200 = 128 + 72: split
    assert sum(1 for t in c if t[0] == 0 and t[1] == TA) == 6 and sum(1 for t in c if t[0] == 1 and t[1] == TA) == 1
    for p, t in enumerate(c):
        if t[1] == TA:
            assert c[p + 1][1] == TU and c[p + 1][0] == t[0] and c[p + 1][3:5] == t[3:5] and c[p + 1][2] & 4
    # the last panel of a pivot column comes alone (K = 128): it is what the next diagonal block / panel tile waits for
    for (fr, typ, nq, i, j, q0, R) in a:
        if fr == 0 and typ == U and j < KB and q0 + nq >= j - 1:
            assert nq == 1


def test_chain_is_woven_into_the_bulk():
    """On a front with many tiles the next diagonal block must not sit behind the whole trailing update of the previous
    panel: its position in the queue is early among that panel's update tasks."""
    tasks, _ = build_queue([(6000, 6000)], workers=256, group=2, rows=4)
    pos = {}
    for p, (a, typ, nq, i, j, q0, R) in enumerate(tasks):
        pos.setdefault((typ, i, j, q0), p)
    for q in (2, 10, 20):
        d_next = pos[(D, q + 1, q + 1, q + 1)]
        assert pos[(D, q, q, q)] < pos[(TU, q + 1, q, q)] < d_next      # TU(q) is popped while D(q) runs (its loads are in flight by the time D(q) is done)
        ups = [p for p, (a, typ, nq, i, j, q0, R) in enumerate(tasks) if typ == U and q0 <= q < q0 + nq and j > q + 1]
        assert ups and d_next < np.percentile(ups, 60), (q, d_next, np.percentile(ups, [10, 50, 90]))
