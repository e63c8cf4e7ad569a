#!/bin/bash
# Collect the rocprofv3 evidence for a round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r01'
# Kernel-trace stats and each PMC group are separate runs (PMC is never combined with sys/hip traces); every run under `timeout`
# (a profiler run that hangs would otherwise hold the box until gpurun's own limit).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
# never profile a stale library: rebuild if any source is newer than the .so (hipcc is on the box; ~10 s when it has to)
(cd $R && python -c "from speech2lip_amd.build import build_library; build_library()" > $O/build.log 2>&1)
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
C="python $R/tools/bench_composite.py 256"
timeout 300 $B > $O/bench_line.json 2> $O/bench.err
timeout 300 $C > $O/composite_line.json 2> $O/composite.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_mfma -o s -- $B > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o s -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o s -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -o s -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cstats -o s -- $C > $O/cstats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cpmc_fetch -o s -- $C > $O/cpmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/cpmc_write -o s -- $C > $O/cpmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TA_BUSY_avr SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/cpmc_sq -o s -- $C > $O/cpmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tpmc_fetch -o s -- python $R/tools/bench_train.py 64 bf16 > $O/tpmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tpmc_write -o s -- python $R/tools/bench_train.py 64 bf16 > $O/tpmc_write.log 2>&1
# the split-bf16 U-Net convolution: matrix-pipe busy and HBM bytes per launch
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/upmc_mfma -o s -- python $R/tools/bench_unet.py 16 > $O/upmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/upmc_fetch -o s -- python $R/tools/bench_unet.py 16 > $O/upmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/upmc_write -o s -- python $R/tools/bench_unet.py 16 > $O/upmc_write.log 2>&1
# the split-half speed mode of the renderer (render16_tiles_kernel): kernel stats, matrix-pipe busy, HBM bytes per launch
S="python $R/tools/bench_render_split.py 1000"
timeout 300 $S > $O/render_split_line.json 2> $O/render_split.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sstats -o s -- $S > $O/sstats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/spmc_mfma -o s -- $S > $O/spmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/spmc_fetch -o s -- $S > $O/spmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/spmc_write -o s -- $S > $O/spmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/spmc_sq -o s -- $S > $O/spmc_sq.log 2>&1
# the half-width convolution kernel of the train-mode chain (convh_asm_kernel)
Hc="python $R/tools/bench_convh.py 20"
timeout 300 $Hc > $O/convh_line.txt 2> $O/convh.err
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/hpmc_mfma -o s -- $Hc > $O/hpmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/hpmc_fetch -o s -- $Hc > $O/hpmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/hpmc_write -o s -- $Hc > $O/hpmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/hpmc_sq -o s -- $Hc > $O/hpmc_sq.log 2>&1
ls $O
# ---- the other rows: kernel-trace stats per tool (one rocprofv3 run each, no counters)
for spec in "train_bf16:tools/bench_train.py 64 bf16" "train_fp32:tools/bench_train.py 64 fp32" "unet:tools/bench_unet.py 16 --bf16" \
            "small_clips:tools/bench_small_clips.py" "config3_split:tools/bench_config3.py 1000 100 --split" \
            "syncnet:tools/bench_syncnet.py 16" "syncnet_split:tools/bench_syncnet.py 16 split" "warp:tools/bench_warp.py 256" "config3:tools/bench_config3.py 1000 100 --unet" \
            "config3_nounet:tools/bench_config3.py 5000 500" "stage1_sync:tools/bench_train.py 64 bf16 --sync=8" \
            "stage1_full:tools/bench_train.py 8 bf16 --full" "stage1_early:tools/bench_train.py 8 bf16 --full --early" "stage1_sync_trainbn:tools/bench_train.py 64 bf16 --sync=8 --trainbn"; do
  name=${spec%%:*}; cmd=${spec#*:}
  timeout 300 python $R/$cmd > $O/${name}_line.txt 2> $O/${name}.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_$name -o s -- python $R/$cmd > $O/x_$name.log 2>&1
done
ls $O
