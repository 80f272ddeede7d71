"""Where does the chain of the largest front wait in the bulk-bound phase?  For every TU(q) of the launch with the most tasks:
when its two tiles had received their panels (the end of the tasks that brought them to state q), when it was popped, when D(q)
arrived, and when it ended.  usage: python scripts/df_chain_slack.py <log>"""
import sys
import numpy as np
launches, cur = [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cur = []; launches.append(cur); continue
    cur.append([int(x) for x in line.split()])
L = np.array(max(launches, key=len), dtype=np.int64)
idx, front, typ, ti, tj, q0, nq, worker, pop, ready, end = L.T[:11]
nqq = nq & 255
t0 = pop.min()
big = np.bincount(front).argmax()
sel = front == big
us = lambda x: (x - t0) / 100.0
# end time of the task that brought tile (i, j) to state s
fin = {}
for a in np.where(sel)[0]:
    t = typ[a]
    if t == 2:
        for r in range(max(nq[a] >> 8, 1)):
            fin[(int(ti[a]) + r, int(tj[a]), int(q0[a]) + int(nqq[a]))] = us(end[a])
    elif t == 3:      # TU(q): tile (q+1, q) -> q+1 ; (q+1,q+1) -> q+1 (or factored)
        fin[(int(ti[a]), int(tj[a]), int(tj[a]) + 1)] = us(end[a])
        fin[(int(ti[a]), int(ti[a]), int(tj[a]) + 1)] = us(end[a])
    elif t == 1:
        fin[(int(ti[a]), int(tj[a]), int(tj[a]) + 1)] = us(end[a])
rows = []
prev_end = None
for a in np.where(sel & (typ == 3))[0]:
    q = int(tj[a])
    t_row = fin.get((q + 1, q, q), 0.0)          # tile (q+1, q) has panels < q
    t_diag = fin.get((q + 1, q + 1, q), 0.0)     # tile (q+1, q+1) has panels < q
    rows.append((q, t_row, t_diag, us(pop[a]), us(ready[a]), us(L[a, 11]) if L.shape[1] > 11 else 0, us(end[a])))
rows.sort()
print("  q | tile (q+1,q) ready | tile (q+1,q+1) ready | TU popped | its tiles seen | D(q) seen | task end (= D(q+1) done) | popped after its tiles were ready by | D(q) done before it was seen by")
for k, r in enumerate(rows):
    q = r[0]
    dq_done = rows[k - 1][6] if k > 0 else float("nan")
    if q % 4 == 0 or q > rows[-1][0] - 3:
        print(f"{q:4d} | {r[1]:9.1f} | {r[2]:9.1f} | {r[3]:9.1f} | {r[4]:9.1f} | {r[5]:9.1f} | {r[6]:9.1f} | {r[3] - max(r[1], r[2]):8.1f} | {r[5] - dq_done:8.1f}")
# the feeder path of TU(q): D(q-1) -> T(q+1, q-1) -> U(q+1, q; panel q-1) and U(q+1, q+1; panel q-1)
print("feeder tasks for a few q (us): D(q-1) done | T(q+1,q-1) pop ready end | U(q+1,q;..q-1) pop ready end (K) | U(q+1,q+1;..q-1) pop ready end (K)")
Tt = {(int(ti[a]), int(tj[a])): a for a in np.where(sel & (typ == 1))[0]}
Ut = {}
for a in np.where(sel & (typ == 2))[0]:
    for r in range(max(nq[a] >> 8, 1)):
        Ut[(int(ti[a]) + r, int(tj[a]), int(q0[a]) + int(nqq[a]))] = a
ends = {r[0]: r[6] for r in rows}
for q in (8, 16, 24, 32, 40, 44):
    a = Tt.get((q + 1, q - 1)); b = Ut.get((q + 1, q, q)); c = Ut.get((q + 1, q + 1, q))
    f = lambda x: "   -" if x is None else f"{us(pop[x]):8.1f} {us(ready[x]):8.1f} {us(end[x]):8.1f}"
    k = lambda x: 0 if x is None else 128 * int(nqq[x])
    print(f"{q:4d} | {ends.get(q - 2, float('nan')):8.1f} | {f(a)} | {f(b)} ({k(b)}) | {f(c)} ({k(c)})")
