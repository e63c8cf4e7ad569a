"""Per-kernel table of the LAST occurrence of a kernel sequence in a rocprofv3 kernel trace: time, share, grid.
    python tools/kernel_table.py <kernel_trace.csv> <name of the sequence's first kernel (substring)>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
seg = rows[idx[-1]:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(dur(r) for r in seg)
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"{len(seg)} kernels, {tot:.1f} us busy, {wall:.1f} us wall")
agg = collections.OrderedDict()
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").replace("s2l::", "").split("(")[0]
    a = agg.setdefault(n, [0, 0.0, []])
    a[0] += 1; a[1] += dur(r)
    a[2].append((round(dur(r)), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
for n, (c, t, l) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{n:32s} {c:3d} {t:9.1f} us {100 * t / tot:5.1f}%  {l[:6]}")
