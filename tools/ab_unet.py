"""A/B aid for the U-Net's convolution kernels: sha256 of the eval forward's output on a few sizes, and its time at 500x500.
    S2L_LIB=ab/other.so python tools/ab_unet.py      # another build of the same ABI must print the same digests"""
import hashlib, json, sys
import numpy as np, torch
from benchlib import W, _median_ms
import speech2lip_amd as s2l
dev = torch.device("cuda:0")
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
res = {}
for (F, H, Wd) in ((2, 64, 80), (1, 37, 53), (3, 500, 500), (1, 412, 364)):
    x = torch.from_numpy(W.synthetic_image((F, H, Wd, 3), 5, "x")).to(dev)
    out = u.forward_nhwc(x)
    torch.cuda.synchronize()
    res[f"{F}x{H}x{Wd}"] = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
# the backward passes (input-gradient convolutions with / without gate; raw convolutions of the train-mode forward)
h = lambda a: hashlib.sha256(a.detach().cpu().numpy().tobytes()).hexdigest()[:12]
for (F, H, Wd) in ((2, 64, 80), (1, 101, 77)):
    x = torch.from_numpy(W.synthetic_image((F, H, Wd, 3), 7, "x")).to(dev)
    g = torch.from_numpy(W.synthetic_image((F, H, Wd, 3), 8, "x")).to(dev) - 0.5
    out, ctx = u.forward_saved_nhwc(x)
    res[f"bwd_{F}x{H}x{Wd}"] = h(out) + "/" + h(u.backward_input(ctx, g))
    u.train()
    out, ctx = u.forward_train_nhwc(x, update_running=False)
    dx, grads = u.backward_train(ctx, g)
    res[f"train_{F}x{H}x{Wd}"] = h(out) + "/" + h(dx) + "/" + h(torch.cat([grads[k].reshape(-1) for k in sorted(grads)]))
    u.eval()
x = torch.rand(16, 500, 500, 3, device=dev)
out = torch.empty_like(x)
ms = _median_ms(lambda: u.forward_nhwc(x, out=out), reps=5, inner=1)
res["ms_16_frames"] = round(ms, 3)
res["fps"] = round(16 / ms * 1e3, 1)
print(json.dumps(res))
