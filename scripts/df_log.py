"""Reads the per-task time stamps of a dataflow launch (OKKT_DEBUG_DATAFLOW=16, OKKT_DF_LOG=<file>) and prints where the time of
the launch goes: durations per task kind, the chain D(q) -> T(q+1, q) -> U(q+1, q+1) -> D(q+1) of the largest front step by step,
and how busy the workers were.
usage: python scripts/df_log.py <log> [launch index, default: the one with most tasks]"""
import sys

import os

import numpy as np

CHUNK = int(os.environ.get("OKKT_DF_KC", "32"))      # panel columns per operand chunk of the build that wrote the log

path = sys.argv[1]
launches = []
cur = None
for line in open(path):
    if line.startswith("#"):
        cur = []
        launches.append(cur)
        continue
    cur.append([int(x) for x in line.split()])
if not launches:
    sys.exit("no launches in the log")
for n_, l_ in enumerate(launches):
    if not l_:
        continue
    A = np.array(l_, dtype=np.int64)
    sp_ = (A[:, 10].max() - A[:, 8].min()) / 100.0
    nw_ = len(set(A[:, 7]))
    body_ = (A[:, 10] - A[:, 9]).sum() / 100.0
    print(f"launch {n_:3d}: {len(A):6d} tasks {len(set(A[:, 1])):4d} fronts {nw_:4d} workers span {sp_:8.1f} us  busy {body_ / (nw_ * sp_) * 100:5.1f} %  D {int((A[:, 2] == 0).sum()):4d}")
which = int(sys.argv[2]) if len(sys.argv) > 2 else int(np.argmax([len(l) for l in launches]))
L = np.array(launches[which], dtype=np.int64)
idx, front, typ, ti, tj, q0, nq, worker, pop, ready, end = L.T[:11]
place = worker >> 16              # two-kernel form: XCC id | HW_ID << 4 of the bulk workers
worker = worker & 0xffff
if place.any():
    xcc = place & 15
    hw = place >> 4
    cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)      # cu_id, sh_id, se_id, xcc
    m = worker >= 1000
    wk = np.unique(np.stack([worker[m], cu[m]], axis=1), axis=0)
    per_cu = np.bincount(np.unique(wk[:, 1], return_inverse=True)[1])
    print(f"  bulk workers seen: {len(wk)} on {len(per_cu)} CUs; CUs with 1 / 2 / more workers: {(per_cu == 1).sum()} / {(per_cu == 2).sum()} / {(per_cu > 2).sum()}; per XCD: {np.bincount(wk[:, 1] >> 8, minlength=8).tolist()}")
nrows = np.maximum(nq >> 8, 1)
nq = nq & 255
marks = L[:, 11:15] if L.shape[1] >= 15 else None
t0 = pop.min()
pop = (pop - t0) / 100.0
ready = (ready - t0) / 100.0
end = (end - t0) / 100.0
span = end.max()
print(f"launch {which}: {len(L)} tasks, {len(set(worker))} workers, span {span:.1f} us")
names = {0: "D", 1: "T", 2: "U", 3: "TU", 4: "TA", 5: "TL"}
for k in (0, 1, 2, 3, 4, 5):
    m = typ == k
    if not m.any():
        continue
    for sel, label in ((m, names[k]),) if k != 2 else ((m & (nq == 1), "U K=128"), (m & (nq == 2), "U K=256"), (m & (nq == 3), "U K=384"), (m & (nq >= 4), "U K=512")):
        if not sel.any():
            continue
        body = end[sel] - ready[sel]
        wait = ready[sel] - pop[sel]
        print(f"  {label:8s} n {sel.sum():6d}  body us: mean {body.mean():6.1f} med {np.median(body):6.1f} max {body.max():6.1f} | wait us: mean {wait.mean():6.1f} med {np.median(wait):6.1f} max {wait.max():7.1f} | sum body {body.sum() / 1e3:8.2f} ms")
if os.environ.get("OKKT_DF_LOG_SAVE"):
    np.savez_compressed(os.environ["OKKT_DF_LOG_SAVE"], L=L)
# time line: per twentieth of the span the share of the workers inside a task body / waiting in a popped task, the update work
# started and the block column the chain has reached
nbk = 20
edges = np.linspace(0.0, span, nbk + 1)
nwk = len(set(worker))
print("  time line (us from | body % | wait % | U tasks started | diagonal blocks done by then)")
dends = np.sort(end[(typ == 0) | ((typ == 3) & ((nq & 2) != 0))])
for b in range(nbk):
    lo, hi = edges[b], edges[b + 1]
    inb = np.clip(np.minimum(end, hi) - np.maximum(ready, lo), 0, None).sum()
    inw = np.clip(np.minimum(ready, hi) - np.maximum(pop, lo), 0, None).sum()
    nu = int(((typ == 2) & (ready >= lo) & (ready < hi)).sum())
    print(f"    {lo:8.0f} | {inb / (nwk * (hi - lo)) * 100:5.1f} | {inw / (nwk * (hi - lo)) * 100:5.1f} | {nu:5d} | {int((dends <= hi).sum()):3d}")
busy = (end - ready).sum()
waits = (ready - pop).sum()
nw = len(set(worker))
print(f"  workers: body {busy / (nw * span) * 100:.1f} % of {nw} x span, waiting in a popped task {waits / (nw * span) * 100:.1f} %")
# chain of the largest front
big = np.bincount(front).argmax()
sel = front == big
D = {int(i): (pop[a], ready[a], end[a]) for a, i in zip(np.where(sel & (typ == 0))[0], ti[sel & (typ == 0)])}
T = {(int(i), int(j)): (pop[a], ready[a], end[a]) for a, i, j in zip(np.where(sel & (typ == 1))[0], ti[sel & (typ == 1)], tj[sel & (typ == 1)])}
U = {}
for a in np.where(sel & (typ == 2))[0]:
    U[(int(ti[a]), int(tj[a]), int(q0[a]) + int(nq[a]))] = (pop[a], ready[a], end[a])
TUt = {int(j): (pop[a], ready[a], end[a], int(nq[a])) for a, j in zip(np.where(sel & (typ == 3))[0], tj[sel & (typ == 3)])}
fused = any(v[3] & 2 for v in TUt.values())
if fused:
    # D(q + 1) rides in TU(q): the end of TU(q) is the end of the diagonal block of block column q + 1
    print(f"chain of front {big} (TU(q) + D(q + 1) in one task): q | TU pop ready end | end - end of the previous diagonal block")
    prev = D[0][2] if 0 in D else None
    ends = [prev] if prev is not None else []
    for q in sorted(TUt):
        t = TUt[q]
        row = f"  {q:3d} | {t[0]:8.1f} {t[1]:8.1f} {t[2]:8.1f}"
        if prev is not None:
            row += f" | {t[2] - prev:6.1f}"
        prev = t[2]
        ends.append(prev)
        if q < 5 or q % 8 == 0 or q >= max(TUt) - 2:
            print(row)
    if len(ends) > 1:
        print(f"  mean distance between diagonal blocks {(ends[-1] - ends[0]) / (len(ends) - 1):.1f} us")
    qs = []
else:
    print(f"chain of front {big}: step | D pop ready end (body) | T(q+1,q) ready end | U(q+1,q+1,->q+1) ready end | D(q+1) ready - D(q) end")
    qs = sorted(D)
for q in qs:
    d = D[q]
    t = T.get((q + 1, q))
    u = U.get((q + 1, q + 1, q + 1))
    dn = D.get(q + 1)
    row = f"  {q:3d} | {d[0]:8.1f} {d[1]:8.1f} {d[2]:8.1f} ({d[2] - d[1]:5.1f})"
    if t:
        row += f" | {t[1]:8.1f} {t[2]:8.1f} ({t[2] - t[1]:5.1f})"
    if u:
        row += f" | {u[1]:8.1f} {u[2]:8.1f} ({u[2] - u[1]:5.1f})"
    if dn:
        row += f" | {dn[1] - d[2]:6.1f}"
    if q < 6 or q % 8 == 0 or q >= qs[-1] - 2:
        print(row)
if marks is not None and ((typ == 3) & ((nq & 4) != 0)).any():
    # split block rows: absolute times of the hand-over, relative to the arrival of D(q) at the lower half
    ta = {(int(f_), int(j_)): a for a, (f_, j_) in enumerate(zip(front, tj)) if typ[a] == 4}
    rows_ = []
    for a in np.where((typ == 3) & ((nq & 4) != 0))[0]:
        b_ = ta.get((int(front[a]), int(tj[a])))
        if b_ is None:
            continue
        mb, ma = (marks[a] - t0) / 100.0, (marks[b_] - t0) / 100.0
        rows_.append([ma[0] - mb[0], ma[1] - mb[0], ma[2] - mb[0], ma[3] - mb[0], end[b_] - mb[0], mb[1] - mb[0], mb[2] - mb[0], mb[3] - mb[0]])
    if rows_:
        r_ = np.mean(np.array(rows_), axis=0)
        print(f"  split block rows (us after D(q) reached the lower half, mean): upper half: D seen {r_[0]:.1f}, rows solved {r_[1]:.1f}, W / L out and published {r_[2]:.1f}, sub-tiles updated {r_[3]:.1f}, stored and drained {r_[4]:.1f} | lower half: rows solved {r_[5]:.1f}, upper W published seen {r_[6]:.1f}, tile updated {r_[7]:.1f}")
if marks is not None and ((typ == 3) & ((nq & 4) == 0)).any():
    m = (typ == 3) & ((nq & 4) == 0)
    mk = (marks[m] - t0) / 100.0
    print(f"  TU phases (us, mean): ready -> D arrived {np.mean(mk[:, 0] - ready[m]):.1f}, rows solved +{np.mean(mk[:, 1] - mk[:, 0]):.1f}, W in LDS +{np.mean(mk[:, 2] - mk[:, 1]):.1f}, tile updated, stored and published +{np.mean(mk[:, 3] - mk[:, 2]):.1f}, W and L stored + drained +{np.mean(end[m] - mk[:, 3]):.1f}")
if marks is not None and (typ == 0).any():
    m = (typ == 0) & (marks[:, 3] > 0)
    if m.any():
        mk = marks[m].astype(float) / 100.0
        tail = end[m] - (marks[m][:, 3] - t0) / 100.0
        print(f"  D phases (us, mean over {int(m.sum())} stand-alone diagonal blocks, seen by a row wave): before the loop {np.mean(end[m] - ready[m] - mk[:, 0] - mk[:, 1] - mk[:, 2] - tail):.1f}, waiting for the head of the update + panel copy {np.mean(mk[:, 0]):.1f}, row phase {np.mean(mk[:, 1]):.1f}, waiting for the rest of the update {np.mean(mk[:, 2]):.1f}, tail (pivot counts, 32 x 32 inverses, stores, drain) {np.mean(tail):.1f}")
if marks is not None and (typ == 4).any():
    m = typ == 4
    mk = (marks[m] - t0) / 100.0
    print(f"  TA phases (us, mean): ready -> D arrived {np.mean(mk[:, 0] - ready[m]):.1f}, rows solved +{np.mean(mk[:, 1] - mk[:, 0]):.1f}, W in LDS +{np.mean(mk[:, 2] - mk[:, 1]):.1f}, W and L out + published, sub-tiles updated +{np.mean(mk[:, 3] - mk[:, 2]):.1f}, stored + drained +{np.mean(end[m] - mk[:, 3]):.1f}")
if marks is not None and (typ == 2).any():
    for kq in sorted(set(nq[typ == 2])):
        m = (typ == 2) & (nq == kq) & (marks[:, 2] > 0)
        if not m.any():
            continue
        mk = (marks[m] - t0) / 100.0
        print(f"  U K={128 * kq} phases (us, median): ready -> C tile + first chunk landed {np.median(mk[:, 0] - ready[m]):.1f}, main loop +{np.median(mk[:, 1] - mk[:, 0]):.1f} ({np.median(mk[:, 1] - mk[:, 0]) / (128 * kq / CHUNK):.2f} per {CHUNK}-column chunk), stores issued +{np.median(mk[:, 2] - mk[:, 1]):.1f}, drained + barrier +{np.median(end[m] - mk[:, 2]):.1f}; pop -> ready {np.median(ready[m] - pop[m]):.1f}; shader clock in the main loop {np.median(marks[m][:, 3] / np.maximum(marks[m][:, 1] - marks[m][:, 0], 1)) / 10:.2f} GHz ({np.median(marks[m][:, 3]) / (128 * kq / CHUNK):.0f} cycles per chunk)")
if len(qs) > 1:
    per = (D[qs[-1]][2] - D[qs[0]][1]) / (len(qs) - 1)
    print(f"  mean distance between diagonal blocks {per:.1f} us")
