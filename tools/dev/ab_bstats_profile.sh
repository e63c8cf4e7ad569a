#!/bin/bash
# kernel-level A/B of the input-gradient convolution's backward statistics in the config-5 step (through gpurun from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for e in 0 1; do
  O=$R/gpurun_out/prof_bstats_$e
  rm -rf $O; mkdir -p $O
  if [ $e = 1 ]; then export S2L_NO_CONV_BSTATS=1; else unset S2L_NO_CONV_BSTATS; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/tools/bench_train.py 64 bf16 --sync=8 --trainbn > $O/stats.log 2>&1
  python - <<PY > $R/gpurun_out/prof_bstats_$e.txt
import csv, glob
f = glob.glob("$O/stats/**/s_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("NO_CONV_BSTATS=$e kernel time total %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%8.2f ms %6d calls %9.1f us avg  %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:100]))
PY
  rm -rf $O/stats
done
