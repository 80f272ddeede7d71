import sys, numpy as np
L = []
for line in open(sys.argv[1]):
    if line.startswith("#"): continue
    L.append([int(x) for x in line.split()])
L = np.array(L, dtype=np.int64)
m = (L[:, 2] == 0) & (L[:, 11] > 0)
end = L[m, 10]; mk = L[m, 11:15]
print("D tasks", m.sum(), "loop end -> counts flushed %.1f us, -> inverses in LDS %.1f, -> X stores issued %.1f, -> task end %.1f" % (
    np.mean(mk[:, 0] - mk[:, 3]) / 100, np.mean(mk[:, 1] - mk[:, 0]) / 100, np.mean(mk[:, 2] - mk[:, 1]) / 100, np.mean(end - mk[:, 2]) / 100))
