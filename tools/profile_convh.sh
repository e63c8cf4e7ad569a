#!/bin/bash
# PMC view of the half-width convolution kernels (run through gpurun): per kernel name, averages per dispatch of
#   GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY | FETCH_SIZE | WRITE_SIZE, TCC hits / misses
#   gpurun --timeout 900 -- 'bash tools/profile_convh.sh TAG'
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_convh_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for K in 0 2; do
  B="python $R/tools/bench_convh.py 20 500 $K --nogate"
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/k${K}_sq -o s -- $B > $O/k${K}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/k${K}_fetch -o s -- $B > $O/k${K}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/k${K}_write -o s -- $B > $O/k${K}_write.log 2>&1
done
python - <<PY
import csv, glob, collections
for K in (0, 2):
    print("== s2l_set_unet_half_kernel(%d)" % K)
    for leg in ("sq", "fetch", "write"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for f in glob.glob("$O/k%d_%s/**/s_counter_collection.csv" % (K, leg), recursive=True):
            for r in csv.DictReader(open(f)):
                if "convh" not in r["Kernel_Name"]:
                    continue
                k = r["Kernel_Name"].split("(")[0][-28:]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print("  %-28s " % k + "  ".join("%s %.4g" % (c, v / n[(k, c)]) for c, v in sorted(acc[k].items())))
PY
