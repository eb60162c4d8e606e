#!/bin/bash
# diagnostic build of the library with per-phase s_memtime stamps in tri_attn8_kernel -> tools/probes/bin/libabx_stamp.so
set -e
cd "$(dirname "$0")/../../abx_amd/csrc"
mkdir -p ../../tools/probes/bin build_stamp
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I. -I../../include"
/opt/rocm/bin/hipcc $F -DTRI8_STAMP -c attention.hip -o build_stamp/attention.o
OBJS=""
for f in capi gemm gemm3 gemm_as ipa embed opm assemble_bias geometry diffuser guidance blocks; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build_stamp/attention.o -o ../../tools/probes/bin/libabx_stamp.so
