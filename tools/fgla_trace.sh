#!/bin/bash
# Per-phase timing of the FGLA synth / analysis kernels (GPU box): builds libddx_hip.so with -DDDX_FGLA_TRACE into variants/, swaps it in for
# this call only and prints, per launch, the cycles a workgroup spends bringing its operands into LDS, in the FFT-6400 and writing results,
# its lifetime, the kernel span and the average number of workgroups in flight per CU.   bash tools/fgla_trace.sh [B] [iterations]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p variants build/trace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DDDX_FGLA_TRACE ${FGLA_DEFS} -c dualdiffusion_amd/csrc/fgla.hip -Idualdiffusion_amd/csrc -Iinclude -o build/trace/fgla.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_trace.so $(ls build/obj/*.o | grep -v /fgla.o) build/trace/fgla.o
cp dualdiffusion_amd/lib/libddx_hip.so variants/lib_keep.so
cp variants/lib_trace.so dualdiffusion_amd/lib/libddx_hip.so
FGLA_FUSED=0 python -u tools/fgla_bench.py ${1:-4} ${2:-3} 2>&1 | grep -v amdgpu.ids | tail -12
cp variants/lib_keep.so dualdiffusion_amd/lib/libddx_hip.so
