#!/usr/bin/env python3
"""Launches whose workgroups do not fill whole rounds of the chip's slots (a rocprofv3 kernel trace: tools/profile_cmd.sh TAG ...):
    python tools/dev/tail_report.py gpurun_out/prof_cmd_TAG/stats/s_kernel_trace.csv
slots per CU = min(LDS, registers, 32 waves / waves per workgroup, 16 workgroups... ); rounds = workgroups / (256 CUs x slots).  A launch at
1.1 - 1.6 rounds spends its second round on a mostly empty chip."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    n = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    waves = (wg + 63) // 64
    regs = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
    regs = max(regs, 1)
    waves_per_simd = min(8, 512 // ((regs + 7) // 8 * 8))
    by_regs = waves_per_simd * 4 // waves if waves <= waves_per_simd * 4 else 0
    lds = int(r["LDS_Block_Size"])
    by_lds = (160 * 1024) // lds if lds else 99
    slots = max(1, min(by_regs, by_lds, 32 // waves if waves else 1))
    rounds = n / (256 * slots)
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("s2l::", "").replace("void ", "")[-44:]
    key = (name, n, slots)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0, rounds])
    a[0] += 1
    a[1] += d
for (name, n, slots), (calls, tot, rounds) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    frac = rounds - int(rounds)
    flag = "  <-- tail" if rounds > 1.0 and 0.0 < frac < 0.6 and rounds < 4 else ""
    if tot / calls > 30:
        print(f"{tot / calls:8.1f} us x {calls:4d}  {n:6d} workgroups, {slots:2d} per CU -> {rounds:5.2f} rounds  {name}{flag}")
