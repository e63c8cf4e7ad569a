#!/bin/bash
# PMC averages per dispatch of the kernels whose name contains PATTERN, for any command (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/dev/pmc_kernel.sh conv_wgrad_h python tools/bench_train.py 8 bf16 --full --early'
set -u
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_$PAT
mkdir -p $O
CMD=""
for a in "$@"; do case "$a" in tools/*|bench.py) CMD="$CMD $R/$a";; *) CMD="$CMD $a";; esac; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/a -o s -- $CMD > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/b -o s -- $CMD > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c -o s -- $CMD > $O/c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/d -o s -- $CMD > $O/d.log 2>&1
python - <<PY
import csv, glob, collections
for leg in "abcd":
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$O/%s/**/s_counter_collection.csv" % leg, recursive=True):
        for r in csv.DictReader(open(f)):
            if "$PAT" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print("%-40s " % k + "  ".join("%s %.4g" % (c, v / n[(k, c)]) for c, v in sorted(acc[k].items())))
PY
