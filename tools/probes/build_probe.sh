#!/bin/bash
# Probe build of libabx_hip.so: one source recompiled with extra defines, linked against the objects of the regular build.
#   tools/probes/build_probe.sh <out name> <source.hip> [-DDEFINE ...]      -> tools/probes/bin/libabx_hip_<out name>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/abx_amd/csrc
name=$1; src=$2; shift 2
mkdir -p $ROOT/tools/probes/bin /tmp/abx_probe_build
make -C $CS -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$CS -I$ROOT/include "$@" -c $CS/$src -o /tmp/abx_probe_build/${name}.o
objs=""
for f in capi gemm gemm3 gemm_as attention ipa embed geometry diffuser guidance blocks; do
  if [ "$f.hip" == "$src" ]; then objs="$objs /tmp/abx_probe_build/${name}.o"; else objs="$objs $CS/build/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o $ROOT/tools/probes/bin/libabx_hip_${name}.so
echo built $ROOT/tools/probes/bin/libabx_hip_${name}.so
