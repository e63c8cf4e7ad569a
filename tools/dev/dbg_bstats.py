#!/usr/bin/env python3
"""Debug aid for the input-gradient convolution's backward statistics (csrc/gen_convh8_body.py bstats_block): per-tile comparison against a
double-precision reference, and run-to-run differences.  usage: dbg_bstats.py layer F H W"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import speech2lip_amd as s2l                         # noqa: E402
from speech2lip_amd import _abi, weights as W      # noqa: E402
from speech2lip_amd.unet import c32_to_nhwc, nhwc_to_c32      # noqa: E402

CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128), (256, 128), (128, 64), (128, 64), (64, 64)]


def p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def main():
    layer, F, H, Wd = (int(a) for a in sys.argv[1:5])
    dev = torch.device("cuda:0")
    lib = _abi.load()
    net = s2l.SimpleUnetLight().to(dev).train()
    net.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
    tensors = net._tensors()
    _, raw16 = net._raw_blobs(tensors, net._table(tensors), True)
    g = torch.Generator(device="cpu").manual_seed(13 * layer + H)
    cin, cout = CONVS[layer]
    dz = nhwc_to_c32((0.25 * torch.randn(F, H, Wd, cout, generator=g)).to(torch.bfloat16).to(dev))
    z_nhwc = torch.randn(F, H, Wd, cin, generator=g).to(torch.bfloat16)
    z = nhwc_to_c32(z_nhwc.to(dev))
    sc = (torch.randn(F, cin, generator=g) + 0.3).to(torch.bfloat16).float()
    sh = (0.5 * torch.randn(F, cin, generator=g)).to(torch.bfloat16).float()
    rows = torch.zeros(F, 512)
    rows[:, :cin], rows[:, cin:2 * cin] = sc, sh
    rows = rows.to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.full((F, cin // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
    _abi.check(lib.s2l_convh_layer(p(raw16), layer, 1, p(dz), cout, None, 0, None, p(ref), H, Wd, F, st), "s2l_convh_layer")
    parts = []
    for _ in range(4):
        out = torch.full_like(ref, -1)
        stat = torch.full((F * 1024 * 2 * cin,), float("nan"), device=dev)
        blocks = ctypes.c_int(0)
        _abi.check(lib.s2l_debug_convh_layer_bstats(p(raw16), layer, p(dz), p(z), p(rows), p(out), p(stat), ctypes.byref(blocks), H, Wd, F, st), "bstats")
        torch.cuda.synchronize()
        print("blocks", blocks.value, "out equal", bool(torch.equal(out, ref)))
        parts.append(stat[:F * blocks.value * 2 * cin].reshape(F, blocks.value, 2, cin).clone().cpu())
    tx, ty = (Wd + 15) // 16, (H + 31) // 32
    gy = c32_to_nhwc(ref).view(torch.bfloat16).double().cpu()
    zd = z_nhwc.double()
    mask = (zd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]) > 0
    gm = gy * mask
    pad = torch.zeros(F, ty * 32, tx * 16, cin, dtype=torch.double)
    padq = pad.clone()
    pad[:, :H, :Wd], padq[:, :H, :Wd] = gm, gm * zd
    want = torch.stack([pad.reshape(F, ty, 32, tx, 16, cin).sum((2, 4)), padq.reshape(F, ty, 32, tx, 16, cin).sum((2, 4))], 3).reshape(F, ty * tx, 2, cin)
    for r in range(1, 4):
        d = (parts[r] != parts[0]) & ~(torch.isnan(parts[r]) & torch.isnan(parts[0]))
        print(f"run {r} vs 0: {int(d.sum())} differing entries; tiles: {sorted(set((int(i[0]), int(i[1])) for i in d.nonzero()[:2000]))[:40]}")
    err = (parts[0].double() - want).abs()
    tol = 1e-4 * want.abs().max()
    bad = (err > tol) | torch.isnan(parts[0])
    print(f"vs reference: {int(bad.sum())} of {bad.numel()} entries off (tol {float(tol):.3g}); worst {float(err.nan_to_num(1e9).max()):.4g}")
    if bad.any():
        idx = bad.nonzero()
        tiles = sorted(set((int(i[0]), int(i[1]) // tx, int(i[1]) % tx) for i in idx))
        print("bad tiles (frame, ty, tx):", tiles[:60], "...", len(tiles))
        for i in idx[:12]:
            f, b, k, c = (int(v) for v in i)
            print(f"  frame {f} tile ({b // tx},{b % tx}) k {k} ch {c}: got {float(parts[0][f, b, k, c]):.5f} want {float(want[f, b, k, c]):.5f}")


if __name__ == "__main__":
    main()
