#!/bin/bash
# per-kernel register / LDS / occupancy table of one .hip file as hipcc compiles it (no GPU needed)
# usage: scripts/kernel_resources.sh onephase.jl_amd/csrc/numeric.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 \
 | awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
        /TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ {vg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
        /LDS Size/ {lds=$(NF-1); printf "%s vgpr %s sgpr %s scratch %s occ %s lds %s\n", name, vg, sg, sc, oc, lds}' | c++filt | sed 's/(okkt::DevPlan[^)]*)//' | sort -u
