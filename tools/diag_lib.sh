#!/bin/bash
# A diagnostic build of libmm2d3d_hip.so with extra -D switches, for upper-bound experiments (results of such a build are NOT
# valid - parts of a kernel are switched off):   tools/diag_lib.sh <name> -DMM_DIAG_FAKESPLIT ...
# -> tools/_bin/libmm2d3d_hip_<name>.so ; use with MM_LIB_PATH=tools/_bin/libmm2d3d_hip_<name>.so python bench.py ...
set -e
name=$1; shift
cd "$(dirname "$0")/../mm2d3d_amd/csrc"
B=../../tools/_bin/diag_$name; mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $*"
pids=()
for f in *.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $B/${f%.hip}.o & pids+=($!); done
for f in conv2d bn2d misc2d; do /opt/rocm/bin/hipcc $FLAGS -DMM_ACT_FP16 -c $f.hip -o $B/${f}_f16.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_bin/libmm2d3d_hip_$name.so $B/*.o
echo built tools/_bin/libmm2d3d_hip_$name.so
