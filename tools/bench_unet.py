"""Post-fusion U-Net throughput at the reference's 500x500 face frame.
FLOPs per frame: 2 * 78.7 GMAC (ten 3x3 convs + 1x1).   python tools/bench_unet.py [frames=16] [--train | --backward]
--backward: eval-mode forward with saved activations + input gradient (the frozen net inside loss.backward());
--train: train mode (BatchNorm batch statistics) forward + full backward (input, weight and BatchNorm gradients): 3x the FLOPs."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
F = int(args[0]) if args else 16
H = Wd = 500
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(F, H, Wd, 3, device=dev)
out = torch.empty_like(x)
for _ in range(2):
    u.forward_nhwc(x, out=out)
torch.cuda.synchronize()
evs = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); u.forward_nhwc(x, out=out); e1.record(); evs.append((e0, e1))
torch.cuda.synchronize()
ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
macs = 0
for (name, cin, cout), (h, w) in zip(W.UNET_CONVS, [(500, 500)] * 2 + [(250, 250)] * 2 + [(125, 125)] * 2 + [(250, 250)] * 2 + [(500, 500)] * 2):
    macs += cin * cout * 9 * h * w
macs += 64 * 3 * 500 * 500
tf = 2 * macs * F / (ms * 1e-3) / 1e12
print(json.dumps({"kernel": "s2l_unet_forward (conv3x3_kernel + ...)", "frames": F, "ms": round(ms, 3),
                  "frames_per_s": round(F / ms * 1e3, 1), "gflop_per_frame": round(2 * macs / 1e9, 2),
                  "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4)}}))

ref = out.clone()
modes = [("split", "s2l_unet_forward_split, split operands hi + lo as IEEE halves (conv3x3_split_kernel): the inference speed mode")]
if "--bf16" in sys.argv:      # plain bf16 operands: the training chain's precision (outside the inference tolerance)
    modes.append(("bf16", "s2l_unet_forward, bf16 operands (conv3x3_bf16_kernel)"))
for prec, label in modes:
    for _ in range(2):
        u.forward_nhwc(x, out=out, precision=prec)
    torch.cuda.synchronize()
    evs = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); u.forward_nhwc(x, out=out, precision=prec); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    ms16 = float(np.median([a.elapsed_time(b) for a, b in evs]))
    mse = float(((out.double() - ref.double()) ** 2).mean())
    print(json.dumps({"kernel": label, "frames": F, "ms": round(ms16, 3),
                      "frames_per_s": round(F / ms16 * 1e3, 1), "speedup_vs_fp32": round(ms / ms16, 2),
                      "psnr_db_vs_fp32": round(10 * np.log10(1.0 / max(mse, 1e-30)), 1), "rmse_vs_fp32": float(np.sqrt(mse)),
                      "rel_l2_vs_fp32": float((out - ref).norm() / ref.norm())}))

if "--backward" in sys.argv or "--train" in sys.argv:
    train = "--train" in sys.argv
    if train:
        u.train()
    d = torch.randn(F, H, Wd, 3, device=dev)
    def fb():
        if train:
            o, ctx = u.forward_train_nhwc(x, update_running=False)
            u.backward_train(ctx, d)
        else:
            o, ctx = u.forward_saved_nhwc(x)
            u.backward_input(ctx, d)
    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    evs = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fb(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    passes = 3 if train else 2
    tf = passes * 2 * macs * F / (ms * 1e-3) / 1e12
    print(json.dumps({"kernel": "U-Net " + ("train mode: forward + input/weight/BatchNorm gradients" if train else "frozen: saved forward + input gradient"),
                      "frames": F, "ms": round(ms, 3), "ms_per_frame": round(ms / F, 3), "gflop_per_frame": round(passes * 2 * macs / 1e9, 1),
                      "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4)}}))
