#!/usr/bin/env python3
"""SyncNet, exact fp32 form (conv_gemm_kernel + conv_reduce_kernel, csrc/conv_gemm.h): per layer at batch 16, measured microseconds against
the layer's floor (flops / 157.3 TFLOP/s of fp32 MFMA + weight bytes / 5 TB/s), its launches and its split-K factor -- the table VERDICT r05
item 7 asks for before deciding on a persistent per-encoder launch.  (src/face_simple/models/syncnet.py:7-67, conv.py:5-19.)

Two steps, because the per-launch times come from rocprofv3's kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sn_trace -o s -- python tools/dev/syncnet_layer_table.py run 16
    python tools/dev/syncnet_layer_table.py report gpurun_out/sn_trace 16 > profiles/r06_syncnet_layers.txt
`run` = 3 warm-up + 10 timed passes of the loss's forward (s2l_syncnet_forward_pair: 2 B face windows -- generated + negative -- and B mel
windows, each encoder once) and of the face encoder's input gradient; `report` maps the trace's conv launches to layers with the launch plan
below, which restates launch_conv's (the C++ is the authority: the counts are asserted against the trace)."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speech2lip_amd import weights as W      # noqa: E402

FP32_PEAK, L2_HBM_BW = 157.3e12, 5.0e12
K_PARTIAL = 1 << 23


def ceil_to(a, b):
    return (a + b - 1) // b * b


def shapes(blocks, h, w):
    out = []
    for cin, cout, (kh, kw), (sy, sx), (py, px), res in blocks:
        ho, wo = (h + 2 * py - kh) // sy + 1, (w + 2 * px - kw) // sx + 1
        out.append(((h, w), (ho, wo)))
        h, w = ho, wo
    return out


def plan(spec, in_hw, out_hw, batch, dgrad=False):
    """launch_conv<DGRAD> (csrc/conv_gemm.h) for the exact form: tile, grid, split-K factor, launches; flops and weight bytes of the layer"""
    cin, cout, (kh, kw), _, _, res = spec
    ncols = batch * (in_hw[0] * in_hw[1] if dgrad else out_hw[0] * out_hw[1])
    rows = cin if dgrad else cout
    kc = cout if dgrad else cin
    kcp = ceil_to(kc, 16)
    nchunks = kh * kw * kcp // 16
    TM = 16 if rows <= 16 else 32 if rows <= 32 else 64
    TN = 4096 // TM
    PR = ceil_to(rows, TM)
    tiles = ((ncols + TN - 1) // TN) * ((rows + TM - 1) // TM)
    splits = 1
    if tiles < 1024 and nchunks >= 16:
        splits = min(min(256 // tiles, nchunks // 8), 64) if tiles < 128 else min((1024 + tiles - 1) // tiles, nchunks // 32)
        splits = max(splits, 1)
        while splits > 1 and splits * ncols * PR > K_PARTIAL:
            splits -= 1
    cps = (nchunks + splits - 1) // splits
    splits = (nchunks + cps - 1) // cps
    macs = batch * out_hw[0] * out_hw[1] * cout * cin * kh * kw      # the same for the data gradient
    return {"ncols": ncols, "rows": rows, "K": kh * kw * kc, "tile": f"{TM}x{TN}", "workgroups": tiles * splits, "splits": splits,
            "launches": 1 + (splits > 1), "flops": 2 * macs, "weight_bytes": cout * cin * kh * kw * 4}


def layer_plans(B):
    face_sh, audio_sh = shapes(W.SYNCNET_FACE, 48, 96), shapes(W.SYNCNET_AUDIO, 80, 16)
    fwd = [("face", i, s, plan(s, *face_sh[i], 2 * B)) for i, s in enumerate(W.SYNCNET_FACE)]
    fwd += [("audio", i, s, plan(s, *audio_sh[i], B)) for i, s in enumerate(W.SYNCNET_AUDIO)]
    # the face encoder's input gradient (generated windows only: B of them), last layer first
    bwd = [("face", i, W.SYNCNET_FACE[i], plan(W.SYNCNET_FACE[i], *face_sh[i], B, dgrad=True)) for i in reversed(range(len(W.SYNCNET_FACE)))]
    return fwd, bwd


def run(B):
    import torch
    import speech2lip_amd as s2l
    from speech2lip_amd.syncnet import sync_window
    dev = torch.device("cuda:0")
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    mel, pos, neg = (torch.from_numpy(x).to(dev) for x in W.synthetic_sync_batch(B, seed=1))
    sl = s2l.SyncLoss(net)
    for k in range(13):
        sl.get_sync_contrastive_loss(mel, pos, neg, want_grad=True)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        sl.get_sync_contrastive_loss(mel, pos, neg, want_grad=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"batch {B}: loss + window gradient {e0.elapsed_time(e1) / 10:.3f} ms per call (back to back)")


def report(trace_dir, B):
    paths = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)
    assert paths, f"no kernel trace under {trace_dir}"
    rows = list(csv.DictReader(open(paths[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    conv = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
            if "conv_gemm_kernel" in r["Kernel_Name"] or "conv_reduce_kernel" in r["Kernel_Name"]]
    fwd, bwd = layer_plans(B)
    seq = fwd + bwd
    per_call = sum(p["launches"] for _, _, _, p in seq)
    assert len(conv) % per_call == 0 and len(conv) // per_call >= 10, (len(conv), per_call)
    calls = len(conv) // per_call
    use = conv[(calls - 10) * per_call:]                     # the last ten calls (warm)
    # every launch's mean over the ten calls, in launch order
    mean = [sum(use[c * per_call + k][1] for c in range(10)) / 10 / 1e3 for k in range(per_call)]
    names = [use[k][0] for k in range(per_call)]
    k = 0
    print(f"# SyncNet exact fp32 (conv_gemm_kernel / conv_reduce_kernel), batch {B}: forward of the contrastive loss (2B = {2 * B} face windows, {B} mel windows) and the face")
    print("# encoder's input gradient (B windows); us = mean of 10 warm calls from rocprofv3 --kernel-trace; floor = flops / 157.3 TF + fp32 weight bytes / 5 TB/s")
    print(f"{'pass':4s} {'layer':9s} {'cin->cout k':16s} {'cols x rows x K':22s} {'tile':7s} {'split-K':7s} {'wgs':5s} {'launches':8s} {'GFLOP':7s} {'w MB':6s} {'gemm us':8s} {'reduce us':9s} {'floor us':8s} {'x floor':7s}")
    tot = {"fwd": [0.0, 0.0, 0], "bwd": [0.0, 0.0, 0]}
    for which, lst in (("fwd", fwd), ("bwd", bwd)):
        for enc, i, (cin, cout, (kh, kw), _, _, res), p in lst:
            assert "conv_gemm_kernel" in names[k], (k, names[k])
            g = mean[k]
            k += 1
            r = 0.0
            if p["launches"] == 2:
                assert "conv_reduce_kernel" in names[k], (k, names[k])
                r = mean[k]
                k += 1
            floor = (p["flops"] / FP32_PEAK + p["weight_bytes"] / L2_HBM_BW) * 1e6
            tot[which][0] += g + r
            tot[which][1] += floor
            tot[which][2] += p["launches"]
            print(f"{which:4s} {enc + ' ' + str(i):9s} {f'{cin}->{cout} {kh}x{kw}':16s} {str(p['ncols']) + ' x ' + str(p['rows']) + ' x ' + str(p['K']):22s} {p['tile']:7s} "
                  f"{p['splits']:<7d} {p['workgroups']:<5d} {p['launches']:<8d} {p['flops'] / 1e9:<7.3f} {p['weight_bytes'] / 1e6:<6.2f} {g:<8.1f} {r:<9.1f} {floor:<8.2f} {(g + r) / floor:<7.1f}")
    assert k == per_call
    for which in ("fwd", "bwd"):
        t, f, n = tot[which]
        print(f"# {which}: {n} launches, {t:.0f} us in kernels, floors add up to {f:.0f} us ({t / f:.1f} x)")
    t, f = tot["fwd"][0] + tot["bwd"][0], tot["fwd"][1] + tot["bwd"][1]
    print(f"# loss + window gradient: {t:.0f} us in the convolution launches, sum of floors {f:.0f} us")


if __name__ == "__main__":
    B = int(sys.argv[3] if sys.argv[1] == "report" and len(sys.argv) > 3 else sys.argv[2] if sys.argv[1] == "run" and len(sys.argv) > 2 else 16)
    if sys.argv[1] == "run":
        run(B)
    else:
        report(sys.argv[2], B)
