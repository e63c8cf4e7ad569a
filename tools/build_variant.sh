#!/bin/bash
# Experiment builds of ONE translation unit against the objects of the current tree:
#   tools/build_variant.sh render.hip ab/g1.so -DS2L_RENDER_G=1 [-DS2L_EXP_TRACE ...]
# (run `python -m speech2lip_amd.build` first; use with S2L_LIB=ab/g1.so.  ab/ is not tracked but travels with gpurun.)
set -e
src=$1; out=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
obj=$(mktemp /tmp/variant_XXXX.o)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off "$@" -I$R/speech2lip_amd/build -c $R/speech2lip_amd/csrc/$src -o $obj
others=$(ls $R/speech2lip_amd/build/*.o | grep -v "/${src%.hip}.o" | grep -v "/ref_")
mkdir -p "$(dirname $out)"
hipcc --offload-arch=gfx950 -shared -fPIC -o $out $obj $others
rm -f $obj
