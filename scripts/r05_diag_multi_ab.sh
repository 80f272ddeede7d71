#!/bin/bash
# GPU box: step times of several builds of the diagonal block.  usage: r05_diag_multi_ab.sh <lib name | cur> ...
for c in dense2600 S-C3; do
  for v in "$@"; do
    if [ $v = cur ]; then unset OKKT_LIB_PATH; else export OKKT_LIB_PATH=scripts/_bin/lib_$v.so; fi
    timeout 300 python scripts/df_check.py --run $c /tmp/v_${v}_$c.npz > /dev/null 2>&1
    python3 - $c $v <<'PY'
import sys, numpy as np
c, v = sys.argv[1:3]
b = np.load(f"/tmp/v_{v}_{c}.npz")
print(f"{c:10s} {v:8s} factor ms min {b['tf'].min():.3f}  D checksum {float(np.sum(b['d'])):.17g}")
PY
  done
done
