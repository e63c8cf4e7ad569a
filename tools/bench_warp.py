#!/usr/bin/env python3
"""Time the pose -> warp-grid kernels at the reference's 500x500 face frame.
    python tools/bench_warp.py [FRAMES]
Algorithmic bytes per frame: 8 B/pixel of grid written (+4 B/pixel of depth read when the depth is per frame)."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech2lip_amd import geometry as G


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    H = W = 500
    g = torch.Generator(device="cpu").manual_seed(0)
    ce, ct = torch.tensor([[0.03, 0.01, -0.02]]), torch.tensor([[0.2, 0.1, -9.0]])
    eul = (ce + 0.1 * torch.randn(F, 3, generator=g)).to(dev)
    trn = (ct + 0.3 * torch.randn(F, 3, generator=g)).to(dev)
    ce, ct = ce.to(dev), ct.to(dev)
    d1 = (9.0 + 0.4 * torch.randn(H, W, generator=g)).to(dev)
    dF = d1[None].repeat(F, 1, 1).contiguous()
    out = torch.empty(F, H, W, 2, device=dev)
    T = G.compute_rel_pose_from_obs2can(ce, ct, eul, trn)
    t_pose = timed(lambda: G.compute_rel_pose_from_obs2can(ce, ct, eul, trn))
    for name, d, bpp in (("shared depth", d1, 8), ("per-frame depth", dF, 12)):
        t = timed(lambda: G.warp_grid(d, T, 1200.0, clamp=True, out=out))
        print(f"warp_grid {name:16s}: {t / F * 1e6:7.3f} us/frame  {F / t:10.0f} frames/s  {F * H * W * bpp / t / 1e9:7.1f} GB/s algorithmic")
    print(f"rel_pose ({F} frames): {t_pose * 1e6:.1f} us per call")
    src = torch.rand(H, W, 3, device=dev)
    t = timed(lambda: G.grid_sample(src, out, "border"))
    print(f"grid_sample border   : {t / F * 1e6:7.3f} us/frame  {F * H * W * 20 / t / 1e9:7.1f} GB/s (grid 8 B + out 12 B per pixel)")


if __name__ == "__main__":
    main()
