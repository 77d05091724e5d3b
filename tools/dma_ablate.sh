#!/bin/bash
# Timing ablations of the LDS-DMA conv kernel (GPU box; outputs are WRONG, only the times mean something): the full kernel against
# builds without DMA, without the matrix phase, without the epilogue stores, on tools/conv_big_ab.py cases.
#   bash tools/dma_ablate.sh "L0 64->64 x8 plain,ddec L0 32->32 act"
set -e
cd $GRAFT_REPO_ROOT
mkdir -p variants build/abl
L=dualdiffusion_amd/lib/libddx_hip.so
cp $L variants/lib_keep.so
for v in FULL NODMA NOMATRIX NOSTORE "NODMA -DDDX_ABL_NOSTORE" "NOMATRIX -DDDX_ABL_NOSTORE"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDDX_ABL_$v -c dualdiffusion_amd/csrc/conv_dma.hip -Idualdiffusion_amd/csrc -Iinclude -o build/abl/conv_dma.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L $(ls build/obj/*.o | grep -v conv_dma.o) build/abl/conv_dma.o
  echo "== $v"; python tools/conv_big_ab.py $2 --cases "$1" | cut -c1-52
done
cp variants/lib_keep.so $L
