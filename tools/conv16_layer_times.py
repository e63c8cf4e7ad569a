"""Per-layer times of the U-Net's split-mode forward from a rocprofv3 kernel trace of tools/cmp_conv16.py: the C++ persistent kernel
(selector 0) next to the generated-assembly kernel (selector 2), against the MFMA time of each layer's tiles
(3 x 9 taps x 4 blocks = 108 MFMAs of 32 cycles per 16-channel chunk and 16x16x64 tile and wave).
    python tools/conv16_layer_times.py <kernel_trace.csv> [frames=16] [GHz=2.1]"""
import csv, sys
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.1
H = W = 500
LAYERS = [("L1 64->64 +pool", 64, 64, 1), ("L2 64->128", 64, 128, 2), ("L3 128->128 +pool", 128, 128, 2), ("L4 128->128", 128, 128, 4),
          ("L5 128->128", 128, 128, 4), ("L6 256->128", 256, 128, 2), ("L7 128->64", 128, 64, 2), ("L8 128->64", 128, 64, 1), ("L9 64->64 +out", 64, 64, 1)]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("conv3x3_split", "conv16_asm", "conv3x3_bf16"))]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# the last 10 timed forwards: 5 of selector 0 then 5 of selector 2 (9 conv launches each)
per = 9
runs = [rows[i:i + per] for i in range(len(rows) - 10 * per, len(rows), per)]
cpp, asm = runs[4], runs[9]
tot = [0, 0, 0]
for (name, cin, cout, sc), rc, ra in zip(LAYERS, cpp, asm):
    h, w = H // sc, W // sc
    tiles = F * ((h + 15) // 16) * ((w + 15) // 16) * (cout // 64)
    ideal = -(-tiles // 256) * (cin // 16) * 108 * 32 / (ghz * 1e3)
    kc, ka = ("asm" if "conv16" in r["Kernel_Name"] else "c++" for r in (rc, ra))
    print(f"{name:20s} {kc} {dur(rc):8.1f} us | {ka} {dur(ra):8.1f} us | MFMA time of the tiles {ideal:8.1f} us ({ideal / dur(ra):.3f} of the asm launch)")
    tot[0] += dur(rc); tot[1] += dur(ra); tot[2] += ideal
print(f"{'nine layers':20s}     {tot[0]:8.1f} us |     {tot[1]:8.1f} us | {tot[2]:8.1f} us ({tot[2] / tot[1]:.3f})")
