#!/bin/bash
# probe build of the library whose tri_attn8 producer wave stages nothing (-DTRI8_NO_PRODUCER: timing only, results are wrong)
# -> tools/probes/bin/libabx_noprod.so;  python tools/ab_lib.py tools/probes/bin/libabx_noprod.so tools/probes/kb_tri.py 100
set -e
cd "$(dirname "$0")/../../abx_amd/csrc"
mkdir -p ../../tools/probes/bin build_stamp
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I. -I../../include"
/opt/rocm/bin/hipcc $F -DTRI8_NO_PRODUCER -c attention.hip -o build_stamp/attention_noprod.o
OBJS=""
for f in capi gemm gemm3 gemm_as ipa embed opm assemble_bias geometry diffuser guidance blocks; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build_stamp/attention_noprod.o -o ../../tools/probes/bin/libabx_noprod.so
# the same with the loads kept (-DTRI8_LOADS_ONLY): the HBM stream of K / V without the split and the LDS writes
/opt/rocm/bin/hipcc $F -DTRI8_LOADS_ONLY -c attention.hip -o build_stamp/attention_loadsonly.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build_stamp/attention_loadsonly.o -o ../../tools/probes/bin/libabx_loadsonly.so
