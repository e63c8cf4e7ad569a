"""Race screens for the kernels that wait with hand-counted `s_waitcnt vmcnt(n)` on requests issued as assembly text
(DESIGN.md §4.1, §4.5, §4.6): random shapes, every result compared BIT FOR BIT with an independent kernel that performs the same
arithmetic, and with a second run of itself.

  conv    conv3x3_split_kernel (persistent, bf16 operands) and conv16_asm_kernel (generated assembly): split-bf16 forward and the
          plain-bf16 training chain, against the one-tile-per-workgroup kernels (the soak that found round 3's ~1 %-of-shapes copy race);
  render  render_tiles_kernel / render16_tiles_kernel<long | wide | single>: random (H, W, F), the three tile shapes against each other and the
          auto-picked one, each twice (every sample column sees the same MFMA sequence in every shape);
  convh   convh_asm_kernel (bf16 planes): random layer / direction / gate / size against the fp32-tensor kernel's rounded output;
  bf16    fwd_asm_bf16_kernel / bwd_asm_bf16_kernel: random (h, w, B) rows, assembly vs the C++ kernels (activation images,
          ReLU mask words, rgb, dz images), each twice;
  rows    rows_fs_kernel (rgb_forward's feature-split tile, generated assembly) against the column form rows_fwd_kernel: random row counts,
          each form twice and the automatic choice.

    python tools/soak_conv_kernels.py [rounds=150] [conv|render|bf16|convh|rows|all]

`tests/test_gpu_concurrency.py` runs a 100-shape slice of the three under `-m gpu`."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.talking_face import _ptr, _stream


def with_reference_library(fn):
    """the non-default kernel forms these soaks compare against live in libs2l_hip_ref.so: inside the call `_abi.load()` is that library"""
    def run(*a, **k):
        with _abi.reference_kernels():
            return fn(*a, **k)
    run.__name__, run.__doc__ = fn.__name__, fn.__doc__
    return run


@with_reference_library
def soak_conv(dev, rounds, seed=0, log=print):
    u = s2l.SimpleUnetLight().to(dev).eval()
    u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
    lib = _abi.load()
    rng = np.random.default_rng(seed)
    bad = []
    try:
        for it in range(rounds):
            big = it % 3 == 0      # a third: few large frames; the rest: many small ones (both reach >= 4 tiles per workgroup, the plain form's threshold)
            F = int(rng.integers(1, 4)) if big else int(rng.integers(1, 64))
            H = int(rng.integers(200, 520)) if big else int(rng.integers(4, 140))
            Wd = int(rng.integers(200, 520)) if big else int(rng.integers(4, 140))
            x = torch.rand(F, H, Wd, 3, device=dev)
            d = torch.randn(F, H, Wd, 3, device=dev)
            res = []
            for kind in (1, 0, 0, 2, 2):      # one tile per workgroup; the persistent C++ kernel twice; the generated-assembly kernel twice
                _abi.check(lib.s2l_set_unet_split_kernel(kind), "kind")
                a = u.forward_nhwc(x, precision="split").clone()
                o, ctx = u.forward_saved_nhwc(x, precision="bf16")
                res.append((a, o.clone(), u.backward_input(ctx, d).clone()))
            ok = all(torch.equal(res[0][j], res[k][j]) for j in range(3) for k in (1, 2, 3, 4))
            if not ok:
                bad.append(("conv", F, H, Wd))
                log("MISMATCH conv", F, H, Wd, [[torch.equal(res[0][j], res[k][j]) for j in range(3)] for k in (1, 2, 3, 4)])
    finally:
        lib.s2l_set_unet_split_kernel(0)
    return bad


@with_reference_library
def soak_render(dev, rounds, seed=0, log=print):
    lib = _abi.load()
    rng = np.random.default_rng(seed + 1)
    models = {}
    bad = []
    try:
        for it in range(rounds):
            kind = it % 4
            if kind == 0:        # the clip regime: many frames, mid-size crops (several tiles per persistent workgroup)
                H, Wd, F = int(rng.integers(24, 100)), int(rng.integers(24, 100)), int(rng.integers(40, 400))
            elif kind == 1:      # the per-frame regime
                H, Wd, F = int(rng.integers(1, 132)), int(rng.integers(1, 132)), int(rng.integers(1, 4))
            else:                # ragged everything
                H, Wd, F = int(rng.integers(1, 70)), int(rng.integers(1, 70)), int(rng.integers(1, 60))
            m = models.get("m")
            if m is None:
                m = models["m"] = s2l.TalkingFace(dev, s2l.may_config(16, 16), mode="eval").eval()
                m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
            audio = torch.from_numpy(W.synthetic_audio(F, seed=int(rng.integers(1 << 20))).astype(np.float32)).to(dev)
            idx = torch.from_numpy(rng.integers(0, 40000, F)).to(dev)
            for precision in ("fp32", "split"):             # the exact kernel and the split-half speed mode (render16.hip)
                outs = []
                for shape in (0, 1, 2, 3, 4, 0, 1, 2, 3, 4):   # 0 = auto; 1 + shape forces long / wide / single / feature-split
                    _abi.check(lib.s2l_set_render_shape(shape), "s2l_set_render_shape")
                    outs.append(m.render_clip(audio, idx, H, Wd, precision=precision).clone())
                if not all(torch.equal(outs[0], o) for o in outs[1:]):
                    bad.append(("render", precision, H, Wd, F))
                    log("MISMATCH render", precision, H, Wd, F, [bool(torch.equal(outs[0], o)) for o in outs[1:]])
            m._tables = {}       # the pixel tables of this crop size are not needed again
    finally:
        lib.s2l_set_render_shape(0)
    return bad


@with_reference_library
def soak_bf16(dev, rounds, seed=0, log=print):
    lib = _abi.load()
    rng = np.random.default_rng(seed + 2)
    m = s2l.TalkingFace(dev, s2l.may_config(16, 16), mode="eval").eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
    pb, pf = m.packed_weights_bf16(), m.packed_weights()
    bad = []
    try:
        for it in range(rounds):
            if it % 3 == 0:      # several tiles per persistent workgroup (> 256 tiles of 256 rows)
                h, w, B = int(rng.integers(40, 100)), int(rng.integers(40, 100)), int(rng.integers(3, 9))
            else:
                h, w, B = int(rng.integers(1, 48)), int(rng.integers(1, 48)), int(rng.integers(1, 6))
            P = h * w
            N = 4 * P * B
            Np = int(lib.s2l_bf16_rows_padded(N))
            feat = m.audio_merge_forward(torch.from_numpy(W.synthetic_audio(B, seed=int(rng.integers(1 << 20))).astype(np.float32)).to(dev))
            coords = s2l.get_coords(w, h, dev)
            xT = torch.zeros(Np * 128, dtype=torch.int16, device=dev)
            areas = torch.empty(N, device=dev)
            t_idx = torch.from_numpy(rng.integers(0, 40000, B)).to(dev)
            t_u = torch.from_numpy(rng.random(B, dtype=np.float32)).to(dev)
            _abi.check(lib.s2l_ensemble_rows_bf16(_ptr(pf), _ptr(coords), _ptr(feat), _ptr(t_idx), _ptr(t_u), w, h, _ptr(xT), _ptr(areas),
                                                  P, B, _stream()), "s2l_ensemble_rows_bf16")
            drgb = (torch.randn(N, 3, device=dev) * 1e-3)
            fwd = []
            for kind in (1, 0, 0):                           # 1 = the C++ forward, 0 = the generated assembly (default)
                _abi.check(lib.s2l_set_bf16_forward_kernel(kind), "s2l_set_bf16_forward_kernel")
                hT = torch.zeros(8 * Np * 256, dtype=torch.int16, device=dev)
                masks = torch.zeros(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
                rgb = torch.zeros(N, 3, device=dev)
                _abi.check(lib.s2l_train_forward_bf16(_ptr(pb), _ptr(pf), _ptr(xT), _ptr(hT), _ptr(masks), _ptr(rgb), N, _stream()), "fwd16")
                fwd.append((hT, masks, rgb))
            ok = all(torch.equal(fwd[0][j], fwd[k][j]) for j in range(3) for k in (1, 2))
            masks = fwd[0][1]
            dz_c = torch.zeros(8 * Np * 256, dtype=torch.int16, device=dev)
            dxa = torch.zeros(Np, 64, device=dev)
            _abi.check(lib.s2l_train_backward_bf16(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dz_c), _ptr(dxa), N, _stream()), "bwd16")
            tiles0 = None
            for rep in range(2):
                dz_a = torch.full((8 * Np * 256,), 0x7fc0, dtype=torch.int16, device=dev)
                tiles = torch.full((Np // 256, 64), float("nan"), device=dev)
                _abi.check(lib.s2l_train_backward_bf16_tiles(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dz_a), _ptr(tiles), N, _stream()), "bwd16 asm")
                ok = ok and torch.equal(dz_c, dz_a) and (tiles0 is None or torch.equal(tiles0, tiles))
                tiles0 = tiles
            if not ok:
                bad.append(("bf16", h, w, B))
                log("MISMATCH bf16", h, w, B)
    finally:
        lib.s2l_set_bf16_forward_kernel(0)
    return bad


@with_reference_library
def soak_convh(dev, rounds, seed=0, log=print):
    """convh_asm_kernel (bf16 planes, csrc/convh.hip): random layer / direction / size / gate against the fp32-tensor kernel of the same
    arithmetic (its output rounded to bf16), each twice."""
    from speech2lip_amd.unet import c32_to_nhwc, nhwc_to_c32
    convs = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128), (256, 128), (128, 64), (128, 64), (64, 64)]
    u = s2l.SimpleUnetLight().to(dev).train()
    u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
    tensors = u._tensors()
    raw, raw16 = u._raw_blobs(tensors, u._table(tensors), True)
    lib = _abi.load()
    rng = np.random.default_rng(seed + 3)
    p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    bad = []
    for it in range(rounds):
        layer, tr, gate = int(rng.integers(1, 10)), int(rng.integers(0, 2)), bool(rng.integers(0, 2))
        big = it % 4 == 0
        F = int(rng.integers(1, 3)) if big else int(rng.integers(1, 24))
        H = int(rng.integers(150, 420)) if big else int(rng.integers(1, 90))
        Wd = int(rng.integers(150, 420)) if big else int(rng.integers(1, 90))
        cin, cout = convs[layer]
        if tr:
            cin, cout = cout, cin
        cat = layer in (6, 8) and not tr
        CA, CB = (cin // 2, cin // 2) if cat else (cin, 0)
        a = torch.randn(F, H, Wd, CA, device=dev).to(torch.bfloat16)
        b = torch.randn(F, H, Wd, CB, device=dev).to(torch.bfloat16) if CB else None
        gt = torch.randn(F, H, Wd, cout, device=dev).clamp_min(0).to(torch.bfloat16) if (gate and tr) else None
        a32, b32, g32 = a.float(), (b.float() if cat else None), (gt.float() if gt is not None else None)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ref = torch.zeros(F, H, Wd, cout, device=dev)
        _abi.check(lib.s2l_debug_conv_layer_f32(p(raw), p(raw16), layer, tr, p(a32), CA, p(b32), CB, p(g32), p(ref), H, Wd, F, st), "ref")
        want = ref.to(torch.bfloat16).view(torch.int16)
        ah, bh, gh = nhwc_to_c32(a), (nhwc_to_c32(b) if cat else None), (nhwc_to_c32(gt) if gt is not None else None)
        ok = True
        for kind in (0, 0, 1, 2):      # the eight-wave form twice, the four-wave form, the alternating-roles form (gated launches: eight waves)
            _abi.check(lib.s2l_set_unet_half_kernel(kind), "s2l_set_unet_half_kernel")
            out = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
            _abi.check(lib.s2l_convh_layer(p(raw16), layer, tr, p(ah), CA, p(bh), CB, p(gh), p(out), H, Wd, F, st), "convh")
            ok = ok and torch.equal(c32_to_nhwc(out), want)
        lib.s2l_set_unet_half_kernel(0)
        if not tr and (H + 31) // 32 * ((Wd + 15) // 16) <= 1024:
            # the forward launch that also leaves its tiles' batch statistics (the train-mode chain's): the same output bits, and the
            # partial sums add up to the stored tensor's per-frame, per-channel sum / sum of squares; twice: the partials are the same bits
            parts = []
            for _ in range(2):
                out = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
                stat = torch.full((F * 1024 * 2 * cout,), float("nan"), device=dev)
                blocks = ctypes.c_int(0)
                _abi.check(lib.s2l_debug_convh_layer_stats(p(raw16), layer, p(ah), CA, p(bh), CB, p(out), p(stat), ctypes.byref(blocks), H, Wd, F, st),
                           "convh stats")
                ok = ok and torch.equal(c32_to_nhwc(out), want) and blocks.value == (H + 31) // 32 * ((Wd + 15) // 16)
                parts.append(stat[:F * blocks.value * 2 * cout].reshape(F, blocks.value, 2, cout).clone())
            z = want.view(torch.bfloat16).double()
            got = parts[0].double().sum(1)
            ok = ok and torch.equal(parts[0], parts[1])
            ok = ok and float((got[:, 0] - z.sum((1, 2))).abs().max()) <= 2e-5 * max(1.0, float(z.abs().sum((1, 2)).max()))
            ok = ok and float((got[:, 1] - (z * z).sum((1, 2))).abs().max()) <= 2e-5 * max(1.0, float((z * z).sum((1, 2)).max()))
        if tr and not gate and (H + 31) // 32 * ((Wd + 15) // 16) <= 1024:
            # the input-gradient launch that also leaves stage 1 of the BatchNorm backward of the layer below (the frozen train-mode chain's):
            # the same output bits; the partial sums add up to sum g' / sum g' z over the stored tensor; twice: the same partial bits
            zt = torch.randn(F, H, Wd, cout, device=dev).to(torch.bfloat16)
            sc = (torch.randn(F, cout, device=dev) + 0.3).to(torch.bfloat16).float()      # (8-bit mantissas: the mask is the same in any arithmetic)
            sh = (0.5 * torch.randn(F, cout, device=dev)).to(torch.bfloat16).float()
            rows = torch.zeros(F, 512, device=dev)
            rows[:, :cout], rows[:, cout:2 * cout] = sc, sh
            zh = nhwc_to_c32(zt)
            parts = []
            for _ in range(2):
                out = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
                stat = torch.full((F * 1024 * 2 * cout,), float("nan"), device=dev)
                blocks = ctypes.c_int(0)
                _abi.check(lib.s2l_debug_convh_layer_bstats(p(raw16), layer, p(ah), p(zh), p(rows), p(out), p(stat), ctypes.byref(blocks), H, Wd, F, st),
                           "convh bstats")
                ok = ok and torch.equal(c32_to_nhwc(out), want) and blocks.value == (H + 31) // 32 * ((Wd + 15) // 16)
                parts.append(stat[:F * blocks.value * 2 * cout].reshape(F, blocks.value, 2, cout).clone())
            gy, zd = want.view(torch.bfloat16).double(), zt.double()
            gm = gy * ((zd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]) > 0)
            got = parts[0].double().sum(1)
            ok = ok and torch.equal(parts[0], parts[1])
            ok = ok and float((got[:, 0] - gm.sum((1, 2))).abs().max()) <= 2e-5 * max(1.0, float(gm.abs().sum((1, 2)).max()))
            ok = ok and float((got[:, 1] - (gm * zd).sum((1, 2))).abs().max()) <= 2e-5 * max(1.0, float((gm * zd).abs().sum((1, 2)).max()))
        if not ok:
            bad.append(("convh", layer, tr, F, H, Wd, gate))
            log("MISMATCH convh", layer, tr, F, H, Wd, gate)
    return bad


def soak_rows(dev, rounds, seed=0, log=print):
    """rows_fs_kernel vs rows_fwd_kernel through TalkingFace.rgb_forward: the same bits for any row count (ragged tiles, one to ~8 tiles per workgroup)"""
    lib = _abi.load()
    rng = np.random.default_rng(seed + 5)
    m = s2l.TalkingFace(dev, s2l.may_config(16, 16), mode="eval").eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
    bad = []
    try:
        for it in range(rounds):
            n = int(rng.integers(1, 40)) if it % 5 == 0 else int(rng.integers(1, 6000)) if it % 5 < 4 else int(rng.integers(6000, 34000))
            rows = torch.cat([torch.rand(n, 2, device=dev) * 2 - 1, torch.randn(n, 64, device=dev)], -1)
            t = torch.tensor([int(rng.integers(0, 40000))], device=dev)
            outs = []
            with torch.no_grad():
                for kind in (1, 2, 0, 2, 1):
                    _abi.check(lib.s2l_set_rows_kernel(kind), "s2l_set_rows_kernel")
                    outs.append(m.rgb_forward(rows, time_pts=t).clone())
            if not all(torch.equal(outs[0], o) for o in outs[1:]) or not bool(torch.isfinite(outs[0]).all()):
                bad.append(("rows", n))
                log("MISMATCH rows", n, [bool(torch.equal(outs[0], o)) for o in outs[1:]])
    finally:
        lib.s2l_set_rows_kernel(0)
    return bad


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    which = sys.argv[2] if len(sys.argv) > 2 else "conv"
    bad = []
    for name, fn in (("conv", soak_conv), ("render", soak_render), ("bf16", soak_bf16), ("convh", soak_convh), ("rows", soak_rows)):
        if which in (name, "all"):
            b = fn(dev, rounds)
            torch.cuda.synchronize()
            print(f"{name}: {rounds} shapes, {len(b)} mismatches", flush=True)
            bad += b
    sys.exit(1 if bad else 0)
