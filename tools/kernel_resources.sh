#!/bin/bash
# Register / scratch / LDS use of every kernel in dualdiffusion_amd/csrc as the compiler reports it (-Rpass-analysis=kernel-resource-usage):
# one line per kernel, spilling kernels marked.  usage: tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt   (build container; no GPU)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd "$(dirname "$0")/.."
printf "%-14s %-120s %6s %6s %7s %8s %6s\n" file kernel VGPRs AGPRs spills scratch occ
for f in dualdiffusion_amd/csrc/*.hip; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/_kr.o 2>&1 |
    grep -E "Function Name|    VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - - |
    awk -v f="$(basename $f .hip)" '{ n=$3; v=""; a=""; s=""; sc=""; o="";
      for (i=1;i<=NF;i++) { if ($i=="VGPRs:" && $(i-1)!="Spill:") v=$(i+1); if ($i=="AGPRs:") a=$(i+1); if ($i=="Spill:") s=$(i+1); if ($i=="[bytes/lane]:") sc=$(i+1); if ($i=="[waves/SIMD]:") o=$(i+1) }
      cmd="echo " n " | c++filt"; cmd | getline d; close(cmd);
      printf "%-14s %-120s %6s %6s %7s %8s %6s%s\n", f, substr(d,1,120), v, a, s, sc, o, (s+0>0 || sc+0>0) ? "   <-- spills / scratch" : "" }'
done
