"""Chained update tasks in a task log (OKKT_DEBUG_DATAFLOW=16): how many tiles followed their predecessor without leaving the ring, what a tile
costs the worker with and without a successor.  usage: python scripts/r06_chain_log.py <log> [launch]"""
import sys
import numpy as np
launches, cur = [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cur = []; launches.append(cur); continue
    cur.append([int(x) for x in line.split()])
which = int(sys.argv[2]) if len(sys.argv) > 2 else int(np.argmax([len(l) for l in launches]))
L = np.array(launches[which], dtype=np.int64)
idx, front, typ, ti, tj, q0, nq, worker, pop, ready, end = L.T[:11]
K = (nq & 255) * 128
isU = typ == 2
print(f"launch {which}: {len(L)} tasks, span {(end.max() - pop.min()) / 100:.1f} us")
tot = 0; chained = 0; claimed_not_ready = 0
per = {}
gaps_c, gaps_n = [], []
for w in np.unique(worker):
    m = np.where(worker == w)[0]
    m = m[np.argsort(end[m])]
    for a, b in zip(m[:-1], m[1:]):
        if not (isU[a] and isU[b]):
            continue
        tot += 1
        # b was claimed while a was running
        if pop[b] < end[a] - 100:      # more than 1 us before a's end
            if ready[b] < end[a]:
                chained += 1
                gaps_c.append((end[b] - end[a]) / 100.0 - 0)      # time per tile in a chain: end to end
                per.setdefault(int(K[b]), []).append((end[b] - end[a]) / 100.0)
            else:
                claimed_not_ready += 1
        else:
            gaps_n.append((end[b] - end[a]) / 100.0)
print(f"consecutive update pairs on a worker: {tot}; chained {chained}; claimed early but not ready {claimed_not_ready}")
for k in sorted(per):
    v = np.array(per[k]); print(f"  chained tiles K = {k}: n {len(v)}, end-to-end median {np.median(v):.1f} us, mean {v.mean():.1f}")
for k in (256, 384, 512):
    sel = isU & (K == k)
    if sel.any():
        print(f"  all update tasks K = {k}: n {sel.sum()}, ready -> end median {np.median((end - ready)[sel]) / 100:.1f} us, pop -> end median {np.median((end - pop)[sel]) / 100:.1f}")
# worker-time per K = 512 tile: sum over workers of (last end - first pop) / tiles is dominated by waits; instead end-to-end of consecutive K = 512 tasks
sel512 = []
for w in np.unique(worker):
    m = np.where(worker == w)[0]
    m = m[np.argsort(end[m])]
    for a, b in zip(m[:-1], m[1:]):
        if isU[a] and isU[b] and K[b] == 512 and K[a] == 512:
            sel512.append((end[b] - end[a]) / 100.0)
if sel512:
    v = np.array(sel512); print(f"end-to-end of consecutive K = 512 tiles on one worker: n {len(v)}, median {np.median(v):.1f} us, 25 % {np.percentile(v, 25):.1f}, 75 % {np.percentile(v, 75):.1f}")
# early-popped tasks by kind: how long the popping worker kept them before it was free
names = {0: "D", 1: "T", 2: "U", 3: "TU", 4: "TA", 5: "TL"}
held = {}
for w in np.unique(worker):
    m = np.where(worker == w)[0]
    m = m[np.argsort(end[m])]
    for a, b in zip(m[:-1], m[1:]):
        if pop[b] < end[a] - 100:
            key = names[int(typ[b])] + (" K=%d" % K[b] if typ[b] == 2 else "")
            held.setdefault(key, []).append((end[a] - pop[b]) / 100.0)
for k in sorted(held):
    v = np.array(held[k]); print(f"  popped early: {k:10s} n {len(v):5d}  held for median {np.median(v):5.1f} us (max {v.max():5.1f})")
