#!/bin/bash
# GPU box: A/B of a diagonal-block variant (this tree) against another build: D, L and x bit for bit, then the step times.
# usage: r05_diag_variant_ab.sh [other lib, default scripts/_bin/lib_rl0.so = the round-4 row phase]
OLD=${1:-scripts/_bin/lib_rl0.so}
for c in dense700 dense2600 S-C3; do
  OKKT_LIB_PATH=$OLD timeout 300 python scripts/df_check.py --run $c /tmp/rl0_$c.npz > /dev/null 2>&1
  timeout 300 python scripts/df_check.py --run $c /tmp/rl1_$c.npz > /dev/null 2>&1
  python3 - $c <<'PY'
import sys, numpy as np
c = sys.argv[1]
a, b = np.load(f"/tmp/rl0_{c}.npz"), np.load(f"/tmp/rl1_{c}.npz")
print(c, "D equal", np.array_equal(a["d"], b["d"]), "x equal", np.array_equal(a["x"], b["x"]), "L equal", ("Ldata" in a.files and np.array_equal(a["Ldata"], b["Ldata"])), "factor ms", a["tf"].min().round(3), "->", b["tf"].min().round(3), "resid", float(b["res"]))
PY
done
for c in S-metric S-C3 S-C5; do
  echo -n "other build: "; OKKT_LIB_PATH=$OLD timeout 300 python scripts/step_probe.py $c 2>&1 | tail -1
  echo -n "this tree  : "; timeout 300 python scripts/step_probe.py $c 2>&1 | tail -1
done
