#!/bin/bash
# Ablation / pricing builds of the default eight-wave half-width convolution (results WRONG by construction; timing only):
#   bash tools/dev/ablate_convh8.sh build "0 4096 8192"   (here)      bash tools/dev/ablate_convh8.sh run "0 4096 8192"   (GPU box)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
VARIANTS=${2:-"0 4096 8192"}
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    d=$(mktemp -d /tmp/ch8_XXXX)
    S2L_CH_EXP=$v python $R/speech2lip_amd/csrc/gen_convh8_body.py $d > /dev/null
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -I$d -I$R/speech2lip_amd/build -c $R/speech2lip_amd/csrc/convh.hip -o $d/convh.o 2>/dev/null
    others=$(ls $R/speech2lip_amd/build/*.o | grep -v "/convh.o" | grep -v "/ref_")
    mkdir -p $R/ab
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/ch8_$v.so $d/convh.o $others
    echo built ab/ch8_$v.so
  done
else
  for v in $VARIANTS; do
    printf "EXP %5s: " $v
    S2L_LIB=$R/ab/ch8_$v.so python $R/tools/bench_convh.py 20 500 0 --nogate 2>/dev/null | awk '{ if ($1=="total") print; else printf "%s%s ", $1, ($2=="fwd"?"f":"d") ":" $(NF-3) }'
  done
fi
