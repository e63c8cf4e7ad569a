"""convh_asm_kernel (bf16 tensors, csrc/convh.hip) against the fp32-tensor bf16-operand kernel it replaces: on bf16-representable inputs
its output must be the round-to-nearest-even bf16 of that kernel's fp32 output, bit for bit; then the half-width train-mode chain
(forward + input gradient, per-frame statistics) against the fp32-tensor chain, and timings.

    python tools/cmp_convh.py [frames=8] [H=500] [W=500]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.unet import c32_to_nhwc, nhwc_to_c32
from speech2lip_amd.unet import UNET_CONVS

CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128), (256, 128), (128, 64), (128, 64), (64, 64)]
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())


def bf16_bits(x):
    return x.to(torch.bfloat16).view(torch.int16)


def layer_case(lib, u, raw, raw16, dev, layer, transposed, F, H, Wd, gate, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    cin, cout = CONVS[layer]
    if transposed:
        cin, cout = cout, cin
    split = layer in (6, 8) and not transposed
    CA, CB = (cin // 2, cin // 2) if split else (cin, 0)
    a = torch.randn(F, H, Wd, CA, generator=g).to(torch.bfloat16).to(dev)
    b = torch.randn(F, H, Wd, CB, generator=g).to(torch.bfloat16).to(dev) if CB else None
    gt = (torch.randn(F, H, Wd, cout, generator=g).clamp_min(0)).to(torch.bfloat16).to(dev) if gate else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.full((F, H, Wd, cout), float("nan"), device=dev)
    a32, b32, g32 = a.float(), (b.float() if b is not None else None), (gt.float() if gt is not None else None)
    _abi.check(lib.s2l_debug_conv_layer_f32(p(raw), p(raw16), layer, int(transposed), p(a32), CA, p(b32), CB, p(g32), p(ref), H, Wd, F, st), "ref")
    outs = []
    ah, bh, gh = nhwc_to_c32(a), (nhwc_to_c32(b) if b is not None else None), (nhwc_to_c32(gt) if gt is not None else None)
    for rep in range(2):
        out = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
        _abi.check(lib.s2l_convh_layer(p(raw16), layer, int(transposed), p(ah), CA, p(bh), CB, p(gh), p(out), H, Wd, F, st), "convh")
        outs.append(c32_to_nhwc(out))
    torch.cuda.synchronize()
    want = bf16_bits(ref)
    eq = bool(torch.equal(outs[0], want))
    det = bool(torch.equal(outs[0], outs[1]))
    if not eq:
        d = (outs[0].view(torch.bfloat16).float() - ref)
        bad = (outs[0] != want)
        idx = bad.nonzero()[:5].tolist()
        print("   mismatches", int(bad.sum()), "of", bad.numel(), "max abs diff", float(d.abs().max()), "first", idx, flush=True)
    return eq, det


def main():
    dev = torch.device("cuda:0")
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    Wd = int(sys.argv[3]) if len(sys.argv) > 3 else 500
        _abi.reference_kernels().__enter__()      # the non-default forms live in libs2l_hip_ref.so
    lib = _abi.load()
    u = s2l.SimpleUnetLight().to(dev).train()
    u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
    tensors = u._tensors()
    raw, raw16 = u._raw_blobs(tensors, u._table(tensors), True)
    bad = 0
    cases = [(1, 0, 1, 32, 16, False), (1, 0, 2, 40, 40, False), (9, 1, 1, 33, 17, True), (2, 0, 3, 20, 36, False), (3, 0, 1, 64, 64, False),
             (6, 0, 2, 37, 53, False), (8, 0, 1, 70, 30, False), (6, 1, 1, 37, 53, False), (8, 1, 2, 50, 34, False), (7, 1, 1, 31, 31, True),
             (5, 0, 2, 13, 9, False), (4, 1, 1, 5, 4, False), (1, 1, 2, 131, 77, True), (9, 0, 1, 250, 250, False), (3, 1, 1, 125, 125, True),
             (8, 0, 2, 500, 500, False), (1, 1, 1, 500, 500, True)]
    for k, (layer, tr, f, h, w, gate) in enumerate(cases):
        eq, det = layer_case(lib, u, raw, raw16, dev, layer, tr, f, h, w, gate, k)
        bad += not (eq and det)
        print(f"layer {layer} {'dgrad' if tr else 'fwd  '} F={f} {h}x{w} gate={int(gate)}: bits == bf16(fp32 kernel): {eq}  deterministic: {det}", flush=True)
    # the chain
    x = torch.rand(F, H, Wd, 3, device=dev)
    d = torch.randn(F, H, Wd, 3, device=dev)
    res = {}
    for prec in ("fp32", "bf16", "bf16h"):
        out, ctx = u.forward_train_frames_nhwc(x, update_running=False, precision=prec)
        dx = u.backward_train_frames(ctx, d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out, ctx = u.forward_train_frames_nhwc(x, update_running=False, precision=prec)
            dx = u.backward_train_frames(ctx, d)
        torch.cuda.synchronize()
        res[prec] = (out.clone(), dx.clone(), (time.perf_counter() - t0) / 3 * 1e3)
        del ctx
    for prec in ("bf16", "bf16h"):
        eo = float((res[prec][0] - res["fp32"][0]).norm() / res["fp32"][0].norm())
        ed = float((res[prec][1] - res["fp32"][1]).norm() / res["fp32"][1].norm())
        print(f"{prec:6s}: out rel err {eo:.3e}  dx rel err {ed:.3e}  fwd+bwd {res[prec][2]:.2f} ms per {F} frames (fp32 {res['fp32'][2]:.2f})", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
