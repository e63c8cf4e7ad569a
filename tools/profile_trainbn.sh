#!/bin/bash
# Kernel-level breakdown of the config-5 step with train-mode BatchNorm (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_trainbn.sh r04'
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_trainbn_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/tools/bench_train.py 64 bf16 --sync=8 --trainbn > $O/stats.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$O/stats/**/s_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%8.2f ms %6d calls %9.1f us avg  %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:110]))
PY
