#!/bin/bash
# Pricing builds of the feature-split tile (results wrong by construction): bash tools/dev/ablate_render_fs.sh build|run "0 1 2 3 4"
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
V=${2:-"0 1 2 4"}
if [ "$1" = build ]; then
  for v in $V; do
    d=$(mktemp -d /tmp/fs_XXXX)
    S2L_FS_EXP=$v python $R/speech2lip_amd/csrc/gen_render_fs_body.py $d > /dev/null
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -I$d -I$R/speech2lip_amd/build -c $R/speech2lip_amd/csrc/render.hip -o $d/render.o 2>/dev/null
    others=$(ls $R/speech2lip_amd/build/*.o | grep -v "/render.o" | grep -v "/ref_")
    mkdir -p $R/ab
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/fs_$v.so $d/render.o $others
    echo built ab/fs_$v.so
  done
else
  for v in $V; do
    for cfg in "64 1" "64 4" "96 1"; do
      printf "EXP %s %s: " $v "$cfg"; S2L_LIB=$R/ab/fs_$v.so python $R/tools/dev/dbg_render_fs.py $cfg 2>/dev/null | grep "per call" | cut -c1-75
    done
  done
fi
