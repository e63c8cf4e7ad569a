#!/usr/bin/env python3
"""GPU idle gaps inside a repeated step, from a rocprofv3 kernel trace (tools/profile_cmd.sh TAG ... writes gpurun_out/prof_cmd_TAG/stats/s_kernel_trace.csv):
    python tools/dev/gap_report.py gpurun_out/prof_cmd_TAG/stats/s_kernel_trace.csv [marker kernel substring = fwd_asm_bf16_kernel] [min gap us = 30]
A step = the kernels between two occurrences of the marker kernel; the LAST complete step is listed."""
import csv
import re
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "fwd_asm_bf16_kernel"
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
marks = [i for i, e in enumerate(ev) if marker in e[2]]


def short(n):
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("s2l::", "").replace("void ", "")
    return n[-48:]


for a, b in list(zip(marks[:-1], marks[1:]))[-4:]:
    seg = ev[a:b]
    span, busy = (seg[-1][1] - seg[0][0]) / 1e3, sum(e[1] - e[0] for e in seg) / 1e3
    print(f"step: {len(seg)} kernels, span {span:.0f} us, busy {busy:.0f} us, idle {span - busy:.0f} us")
seg = ev[marks[-2]:marks[-1]]
t0 = seg[0][0]
for i in range(len(seg) - 1):
    g = (seg[i + 1][0] - seg[i][1]) / 1e3
    if g > min_gap:
        print(f"{(seg[i][1] - t0) / 1e3:7.0f} us  gap {g:5.0f} us  after {short(seg[i][2])}  -> before {short(seg[i + 1][2])}")
