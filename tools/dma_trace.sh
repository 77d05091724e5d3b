#!/bin/bash
# Phase timing of the LDS-DMA conv kernel (GPU box): builds libddx_hip.so with -DDDX_DMA_TRACE into variants/, swaps it in for
# this call only and prints cycles per wave and phase for the given tools/conv_big_ab.py cases.
#   bash tools/dma_trace.sh "L0 64->64 x8 plain,ddec L0 32->32 act" [--c16] ["-DDDX_ABL_NODMA -DDDX_ABL_NOSTORE"]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p variants build/trace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDDX_DMA_TRACE $3 -c dualdiffusion_amd/csrc/conv_dma.hip -Idualdiffusion_amd/csrc -Iinclude -o build/trace/conv_dma.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_trace.so $(ls build/obj/*.o | grep -v conv_dma.o) build/trace/conv_dma.o
cp dualdiffusion_amd/lib/libddx_hip.so variants/lib_keep.so
cp variants/lib_trace.so dualdiffusion_amd/lib/libddx_hip.so
DDX_DMA_TRACE=1 python -u tools/conv_big_ab.py --iters 3 $2 --cases "$1" 2>&1 | grep -v amdgpu.ids | awk '/dma trace/{l=$0} /TFLOP/{print l; print $0}'
cp variants/lib_keep.so dualdiffusion_amd/lib/libddx_hip.so
