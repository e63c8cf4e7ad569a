"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<tag>_rocprofv3_summary.txt."""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
out = []
def pmc(dirname, kernel_substr):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    path = f"{src}/{dirname}/s_counter_collection.csv"
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: sum(d.values()) / len(d) for c, d in agg.items()}
def stats(dirname):
    path = f"{src}/{dirname}/s_kernel_stats.csv"
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []
out.append(f"# rocprofv3 evidence, {tag}, MI355X (collected by tools/profile_round.sh; PMC groups are separate runs)\n")
for name in ("bench_line.json", "composite_line.json"):
    p = f"{src}/{name}"
    if os.path.exists(p):
        out.append(f"\n## {name}\n{open(p).read().strip()}\n")
out.append("\n## render path: rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline\n")
for r in stats("stats"):
    out.append("%-92s calls %4s avg_ns %16s pct %7s\n" % (r["Name"][:92], r["Calls"], r["AverageNs"], r["Percentage"]))
out.append("\n## render kernel PMC, per dispatch (1000 frames 96x96)\n")
vals = {}
for d in ("pmc_mfma", "pmc_fetch", "pmc_write", "pmc_sq"):
    vals.update(pmc(d, "render_tiles_kernel"))
for k, v in vals.items():
    out.append("%-34s %.6g\n" % (k, v))
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
    out.append("MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = %.4f of the kernel's cycles\n"
               % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (vals["GRBM_GUI_ACTIVE"] / 8)))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    traffic = vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024
    out.append("HBM traffic per dispatch = 2*FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE = %.4g bytes\n" % traffic)
    # the figure bench.py's roofline.traffic carries: tracked next to the profile it comes from, read by bench.py.  Earlier PMC figures of the
    # SAME kernel text stay in `previous` (the counter moved 2.03e8 -> 2.44e8 between two profiles of one build: bench.py quotes the range)
    digest = __import__("hashlib").sha256(open("speech2lip_amd/build/render_body.inc", "rb").read()).hexdigest()[:16]
    previous = []
    try:
        old = json.load(open("profiles/render_traffic.json"))
        if old.get("kernel_text_sha256_16") == digest and old.get("profile") != f"profiles/{tag}_rocprofv3_summary.txt":
            previous = ([{"profile": old["profile"], "hbm_bytes_per_dispatch": old["hbm_bytes_per_dispatch"]}] + list(old.get("previous", [])))[:4]
        elif old.get("profile") == f"profiles/{tag}_rocprofv3_summary.txt":
            previous = list(old.get("previous", []))
    except (OSError, ValueError, KeyError):
        pass
    json.dump({"profile": f"profiles/{tag}_rocprofv3_summary.txt", "kernel": "s2l::render_tiles_kernel",
               "frames_per_dispatch": 1000, "height": 96, "width": 96, "render_shape": "long (3 x (1 x 12)), auto-picked for 1000 frames",
               "kernel_text_sha256_16": digest,
               "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
               "hbm_bytes_per_dispatch": round(traffic), "formula": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024", "previous": previous},
              open("profiles/render_traffic.json", "w"), indent=1)
p = f"{src}/render_split_line.json"
if os.path.exists(p):
    out.append(f"\n## split-half speed mode of the renderer (opt-in; csrc/render16.hip): python tools/bench_render_split.py 1000\n{open(p).read().strip()}\n")
    for r in stats("sstats"):
        if "render" in r["Name"]:
            out.append("%-92s calls %4s avg_ns %16s pct %7s\n" % (r["Name"][:92], r["Calls"], r["AverageNs"], r["Percentage"]))
    sv = {}
    for d in ("spmc_mfma", "spmc_fetch", "spmc_write", "spmc_sq"):
        sv.update(pmc(d, "render16_tiles_kernel"))
    out.append("render16_tiles_kernel PMC, per dispatch (1000 frames 96x96):\n")
    for k, v in sv.items():
        out.append("%-34s %.6g\n" % (k, v))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sv and "GRBM_GUI_ACTIVE" in sv:
        out.append("MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = %.4f of the kernel's cycles\n"
                   % (sv["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (sv["GRBM_GUI_ACTIVE"] / 8)))
    if "FETCH_SIZE" in sv and "WRITE_SIZE" in sv:
        out.append("HBM traffic per dispatch = 2*FETCH_SIZE + WRITE_SIZE = %.4g bytes (algorithmic: 1000 x 112,448 = 1.124e8)\n"
                   % (sv["FETCH_SIZE"] * 1024 * 2 + sv["WRITE_SIZE"] * 1024))
out.append("\n## composite: rocprofv3 --kernel-trace --stats -- python tools/bench_composite.py 256\n")
for r in stats("cstats"):
    out.append("%-92s calls %4s avg_ns %16s pct %7s\n" % (r["Name"][:92], r["Calls"], r["AverageNs"], r["Percentage"]))
out.append("\n## composite span_kernel PMC, per dispatch (256 frames 500x500, 128x128 lip)\n")
cv = {}
for d in ("cpmc_fetch", "cpmc_write", "cpmc_sq"):
    cv.update(pmc(d, "span_kernel"))
for k, v in cv.items():
    out.append("%-34s %.6g\n" % (k, v))
if "FETCH_SIZE" in cv and "WRITE_SIZE" in cv:
    out.append("HBM traffic per dispatch = 2*FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE = %.4g bytes (algorithmic: 256 x 8,196,608 = 2.098e9)\n"
               % (cv["FETCH_SIZE"] * 1024 * 2 + cv["WRITE_SIZE"] * 1024))
for name, title in (("train_bf16", "training step, bf16 mode: python tools/bench_train.py 64 bf16"),
                    ("train_fp32", "training step, fp32 parity mode: python tools/bench_train.py 64 fp32"),
                    ("unet", "post-fusion U-Net: exact fp32, split-bf16 (the inference speed mode) and plain bf16 operands: python tools/bench_unet.py 16 --bf16"),
                    ("small_clips", "one frame per call / 16-frame clips (tile shapes of the renderer): python tools/bench_small_clips.py"),
                    ("config3_split", "lip 128x128 + composite + U-Net in split-bf16 mode: python tools/bench_config3.py 1000 100 --split"),
                    ("syncnet", "sync loss (T3): python tools/bench_syncnet.py 16"),
                    ("syncnet_split", "sync loss with the split-operand convolutions (what bf16-precision steps use): python tools/bench_syncnet.py 16 split"),
                    ("warp", "pose -> warp grid: python tools/bench_warp.py 256"),
                    ("config3", "lip 128x128 + composite + U-Net: python tools/bench_config3.py 1000 100 --unet"),
                    ("config3_nounet", "BASELINE config 3: lip 128x128 + composite, 5000 frames: python tools/bench_config3.py 5000 500"),
                    ("stage1_sync", "BASELINE config 5 with the sync loss: python tools/bench_train.py 64 bf16 --sync=8"),
                    ("stage1_full", "full stage-1 iteration (MSE + LPIPS on lip and face, U-Net, sync window): python tools/bench_train.py 8 bf16 --full"),
                    ("stage1_early", "early-phase iteration (it <= 100000: the U-Net trains with the MLP, no sync loss): python tools/bench_train.py 8 bf16 --full --early"),
                    ("stage1_sync_trainbn", "config 5 with the sync loss, frozen U-Net in TRAIN-mode BatchNorm (the reference's loop, G16): python tools/bench_train.py 64 bf16 --sync=8 --trainbn")):
    rows = stats("x_" + name)
    if not rows:
        continue
    out.append(f"\n## {title}\n")
    p = f"{src}/{name}_line.txt"
    if os.path.exists(p):
        out.append(open(p).read().strip() + "\n")
    for r in rows[:12]:
        out.append("%-92s calls %4s avg_ns %16s pct %7s\n" % (r["Name"][:92], r["Calls"], r["AverageNs"], r["Percentage"]))
# HBM traffic of the bf16 training step, per kernel (PMC passes of tools/bench_train.py 64 bf16; bytes per dispatch)
tf, tw = {}, {}
for d, dst in (("tpmc_fetch", tf), ("tpmc_write", tw)):
    path = f"{src}/{d}/s_counter_collection.csv"
    if os.path.exists(path):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                agg[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, dd in agg.items():
            dst[k] = (sum(dd.values()) / len(dd) * 1024, len(dd))
if tf or tw:
    out.append("\n## bf16 training step: HBM bytes per dispatch (2*FETCH_SIZE | WRITE_SIZE; separate PMC passes of tools/bench_train.py 64 bf16)\n")
    steps = max(1, min((n for _, n in tf.values()), default=1))
    total = 0.0
    for k in sorted(set(tf) | set(tw), key=lambda k: -(2 * tf.get(k, (0, 0))[0] + tw.get(k, (0, 0))[0]) * max(tf.get(k, (0, 1))[1], 1)):
        fb, n = tf.get(k, (0.0, 0))
        wb, n2 = tw.get(k, (0.0, 0))
        n = max(n, n2)
        if 2 * fb + wb < 1e6:
            continue
        out.append("%-70s dispatches %4d  read %9.3f GB  written %9.3f GB\n" % (k[:70], n, 2 * fb / 1e9, wb / 1e9))
# the split-bf16 U-Net convolution kernel (tools/bench_unet.py 16: ten launches per forward, 16 frames 500x500)
uv = {}
for d in ("upmc_mfma", "upmc_fetch", "upmc_write"):
    uv.update(pmc(d, "conv3x3_split_kernel"))
if uv:
    out.append("\n## U-Net split-bf16 convolution kernel (conv3x3_split_kernel), PMC averages per dispatch: python tools/bench_unet.py 16\n")
    for k, v in uv.items():
        out.append("%-34s %.6g\n" % (k, v))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in uv and "GRBM_GUI_ACTIVE" in uv:
        out.append("MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = %.4f of the kernel's cycles\n"
                   % (uv["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (uv["GRBM_GUI_ACTIVE"] / 8)))
    if "FETCH_SIZE" in uv and "WRITE_SIZE" in uv:
        out.append("HBM traffic per dispatch = 2*FETCH_SIZE + WRITE_SIZE = %.4g bytes\n" % (uv["FETCH_SIZE"] * 2048 + uv["WRITE_SIZE"] * 1024))
# the half-width convolution kernel (tools/bench_convh.py 20: the 18 convolutions of forward + input gradient, 20 frames 500x500)
hv = {}
for d in ("hpmc_mfma", "hpmc_fetch", "hpmc_write", "hpmc_sq"):
    hv.update(pmc(d, "convh_asm_kernel"))
if hv:
    out.append("\n## half-width convolution kernel (convh_asm_kernel), PMC averages per dispatch: python tools/bench_convh.py 20\n")
    p_ = f"{src}/convh_line.txt"
    if os.path.exists(p_):
        out.append(open(p_).read())
    for k, v in hv.items():
        out.append("%-34s %.6g\n" % (k, v))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in hv and "GRBM_GUI_ACTIVE" in hv:
        out.append("MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = %.4f of the kernel's cycles\n"
                   % (hv["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (hv["GRBM_GUI_ACTIVE"] / 8)))
    if "FETCH_SIZE" in hv and "WRITE_SIZE" in hv:
        out.append("HBM traffic per dispatch = 2*FETCH_SIZE + WRITE_SIZE = %.4g bytes\n" % (hv["FETCH_SIZE"] * 2048 + hv["WRITE_SIZE"] * 1024))
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_rocprofv3_summary.txt", "w").writelines(out)
print("".join(out))
