"""Per-layer efficiency of the U-Net eval forward from a rocprofv3 kernel trace of tools/ab_unet.py (its last forward: 16 frames
of 500x500).  For every 3x3 convolution: measured time against the MFMA time of its tiles (576 MFMAs of 32 cycles per 16-channel
chunk and 16x16x64 tile, one tile at a time per CU) at the clock given.
    python tools/unet_layer_times.py gpurun_out/<dir>/*_kernel_trace.csv [GHz]"""
import csv, sys

F, H, W = 16, 500, 500
LAYERS = [  # (name, cin, cout, scale)
    ("inc.2  64->64  +pool", 64, 64, 1), ("down1.1 64->128", 64, 128, 2), ("down1.2 128->128 +pool", 128, 128, 2),
    ("down2.1 128->128", 128, 128, 4), ("down2.2 128->128", 128, 128, 4), ("up1.1  256->128", 256, 128, 2),
    ("up1.2  128->64", 128, 64, 2), ("up2.1  128->64", 128, 64, 1), ("up2.2  64->64 +outc", 64, 64, 1)]


def main(path, ghz=2.3, n_cu=256):
    rows = [r for r in csv.DictReader(open(path)) if "conv3x3" in r["Kernel_Name"]]
    rows = rows[-len(LAYERS):]
    tot_t = tot_ideal = tot_alg = 0.0
    for (name, cin, cout, sc), r in zip(LAYERS, rows):
        h, w = H // sc, W // sc
        tiles = F * ((h + 15) // 16) * ((w + 15) // 16) * (cout // 64)
        per_cu = -(-tiles // n_cu)
        ideal = per_cu * (cin // 16) * 576 * 32 / (ghz * 1e3)              # us
        alg = F * h * w * cin * cout * 18 / (157.3e6 * ghz / 2.4)          # us at the fp32 MFMA peak of this clock
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        kind = "asm" if "asm" in r["Kernel_Name"] else "c++"
        print(f"{name:26s} {kind} {t:9.1f} us   tiles' MFMA time {ideal:9.1f} us ({ideal / t:.3f})   true pixels only {alg / t:.3f}")
        tot_t += t; tot_ideal += ideal; tot_alg += alg
    print(f"{'all nine':26s}     {tot_t:9.1f} us   {tot_ideal / tot_t:.3f}   {tot_alg / tot_t:.3f}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.3)
