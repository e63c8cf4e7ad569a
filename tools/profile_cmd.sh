#!/bin/bash
# Kernel-level breakdown of any command (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_cmd.sh TAG python tools/bench_train.py 8 bf16 --full --early'
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_cmd_$TAG
mkdir -p $O
CMD=""
for a in "$@"; do case "$a" in tools/*|bench.py) CMD="$CMD $R/$a";; *) CMD="$CMD $a";; esac; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/stats/**/s_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%8.2f ms %6d calls %9.1f us avg  %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:100]))
PY
