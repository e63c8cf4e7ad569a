"""A/B sweep for the U-Net's convolution kernels over many frame shapes (tile borders, single-tile frames, exact multiples of
16, more tiles than CUs and fewer): one line of digests per shape -- eval forward, saved forward + input gradient, train-mode
forward + all gradients.  Two builds whose kernels perform the same arithmetic in the same order print identical output:
    python tools/ab_unet_sweep.py > a.txt;  S2L_LIB=ab/unet_cpp.so python tools/ab_unet_sweep.py > b.txt;  cmp a.txt b.txt"""
import hashlib, sys
import numpy as np, torch
from benchlib import W
import speech2lip_amd as s2l
dev = torch.device("cuda:0")
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
h = lambda a: hashlib.sha256(a.detach().cpu().numpy().tobytes()).hexdigest()[:10]
shapes = [(1, 4, 4), (1, 5, 7), (2, 8, 8), (1, 15, 15), (1, 16, 16), (1, 17, 17), (3, 16, 48), (1, 31, 33), (1, 32, 32), (2, 33, 31),
          (1, 47, 65), (1, 64, 64), (1, 65, 63), (5, 20, 36), (1, 96, 96), (2, 128, 130), (1, 257, 63), (1, 70, 300), (1, 4, 200),
          (1, 200, 4), (7, 44, 52), (1, 500, 500), (24, 40, 40)]
rng = np.random.default_rng(3)
shapes += [(int(rng.integers(1, 4)), int(rng.integers(4, 150)), int(rng.integers(4, 150))) for _ in range(12)]
for (F, H, Wd) in shapes:
    x = torch.from_numpy(W.synthetic_image((F, H, Wd, 3), 11, "x")).to(dev)
    g = torch.from_numpy(W.synthetic_image((F, H, Wd, 3), 12, "x")).to(dev) - 0.5
    line = [h(u.forward_nhwc(x))]
    out, ctx = u.forward_saved_nhwc(x)
    line += [h(out), h(u.backward_input(ctx, g))]
    u.train()
    out, ctx = u.forward_train_nhwc(x, update_running=False)
    dx, grads = u.backward_train(ctx, g)
    line += [h(out), h(dx), h(torch.cat([grads[k].reshape(-1) for k in sorted(grads)]))]
    u.eval()
    print(f"{F}x{H}x{Wd}", *line, flush=True)
