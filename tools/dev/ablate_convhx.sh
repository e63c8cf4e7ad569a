#!/bin/bash
# Ablation builds of the alternating-roles convolution (results WRONG by construction; timing only):
#   bash tools/dev/ablate_convhx.sh build      (here: writes ab/chx_<exp>.so)
#   bash tools/dev/ablate_convhx.sh run        (on the GPU box: per-variant total of tools/bench_convh.py --nogate)
# EXP bits (csrc/gen_convh8_body.py): 1 no stores, 2 no halo DMA, 4 no weight DMA, 32 no B reads, 64 no A reads, 512 no epilogue, 2048 no MFMAs
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
VARIANTS="0 1 2 4 6 7 96 103 512 2048"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    d=$(mktemp -d /tmp/chx_XXXX)
    S2L_CH_EXP=$v python $R/speech2lip_amd/csrc/gen_convhx_body.py $d > /dev/null
    obj=$d/convh.o
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -DS2L_WITH_REFERENCE_KERNELS -I$d -I$R/speech2lip_amd/build -c $R/speech2lip_amd/csrc/convh.hip -o $obj 2>/dev/null
    others=$(ls $R/speech2lip_amd/build/*.o | grep -v "/convh.o" | grep -v "/ref_")
    mkdir -p $R/ab
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/chx_$v.so $obj $others
    echo built ab/chx_$v.so
  done
else
  for v in $VARIANTS; do
    printf "EXP %5s: " $v
    S2L_LIB=$R/ab/chx_$v.so python $R/tools/bench_convh.py 20 500 2 --nogate 2>/dev/null | tail -1
  done
fi
