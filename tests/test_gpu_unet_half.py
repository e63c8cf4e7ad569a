"""The half-width train-mode chain of the frozen post-fusion U-Net (csrc/unet_half.inc, csrc/convh.hip, gen_convh_body.py): bf16
tensors between the kernels, bf16 operands, fp32 accumulation / statistics.

The reference side is `SimpleUnetLight.forward` in .train() on one frame per call (SimpleUnetLight.py:99-111 at tf_nerf.py:387, the
frozen net of training.py:436-459 after `it > 100000`) and autograd's backward through it; its fp32 restatement is the exact fp32
chain already pinned by G13 / G16 (tests/test_gpu_training_chain.py).  Pinned here:
  (a) the generated-assembly convolution alone: on bf16-representable inputs its output is, BIT FOR BIT, the round-to-nearest-even
      bf16 of the fp32-tensor kernel's output (same operands, same accumulation order) -- forward layers, input-gradient layers with
      and without the ReLU gate, concatenated inputs, ragged sizes, sizes below one tile, each run twice;
  (b) F frames in one call == F one-frame calls (every frame its own statistics group; running statistics in frame order);
  (c) the chain against the exact fp32 chain and against the fp32-tensor bf16-operand chain, at the bounds bf16 storage allows;
  (d) the mode-following pair routes a frozen train-mode net with precision "bf16" through it (and not when switched off)."""
import ctypes

import numpy as np
import pytest
import torch

import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.unet import c32_to_nhwc, nhwc_to_c32

pytestmark = pytest.mark.gpu
T = torch.from_numpy
CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128), (256, 128), (128, 64), (128, 64), (64, 64)]
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def net(dev):
    u = s2l.SimpleUnetLight().to(dev).train()
    u.load_state_dict({k[len("post_fusion_unet."):]: T(v) for k, v in W.make_unet_state_dict(0).items()})
    return u


@pytest.fixture(scope="module")
def blobs(dev):
    u = net(dev)
    tensors = u._tensors()
    raw, raw16 = u._raw_blobs(tensors, u._table(tensors), True)
    return u, raw, raw16


CASES = [(1, 0, 1, 32, 16, False), (1, 0, 2, 40, 40, False), (9, 1, 1, 33, 17, True), (2, 0, 3, 20, 36, False), (3, 0, 1, 64, 64, False),
         (6, 0, 2, 37, 53, False), (8, 0, 1, 70, 30, False), (6, 1, 1, 37, 53, False), (8, 1, 2, 50, 34, False), (7, 1, 1, 31, 31, True),
         (5, 0, 2, 13, 9, False), (4, 1, 1, 5, 4, False), (1, 1, 2, 131, 77, True), (9, 0, 1, 250, 250, False), (3, 1, 1, 125, 125, True),
         (8, 0, 2, 500, 500, False), (1, 1, 1, 500, 500, True), (9, 0, 7, 1, 1, False), (2, 1, 40, 16, 33, True)]


@pytest.mark.parametrize("layer,transposed,F,H,Wd,gate", CASES)
def test_convolution_is_the_rounded_fp32_tensor_kernel(blobs, dev, layer, transposed, F, H, Wd, gate):
    _, raw, raw16 = blobs
    lib = _abi.load()
    g = torch.Generator(device="cpu").manual_seed(layer * 1000 + H)
    cin, cout = CONVS[layer]
    if transposed:
        cin, cout = cout, cin
    cat = layer in (6, 8) and not transposed
    CA, CB = (cin // 2, cin // 2) if cat else (cin, 0)
    a = torch.randn(F, H, Wd, CA, generator=g).to(torch.bfloat16).to(dev)
    b = torch.randn(F, H, Wd, CB, generator=g).to(torch.bfloat16).to(dev) if CB else None
    gt = torch.randn(F, H, Wd, cout, generator=g).clamp_min(0).to(torch.bfloat16).to(dev) if gate else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.full((F, H, Wd, cout), float("nan"), device=dev)
    a32, b32, g32 = a.float(), (b.float() if cat else None), (gt.float() if gate else None)      # (held: the call is asynchronous)
    _abi.check(lib.s2l_debug_conv_layer_f32(p(raw), p(raw16), layer, transposed, p(a32), CA, p(b32), CB, p(g32), p(ref), H, Wd, F, st),
               "s2l_debug_conv_layer_f32")
    want = ref.to(torch.bfloat16).view(torch.int16)
    ah, bh, gh = nhwc_to_c32(a), (nhwc_to_c32(b) if cat else None), (nhwc_to_c32(gt) if gate else None)      # the chain's plane layout
    for rep in range(2):
        out = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
        _abi.check(lib.s2l_convh_layer(p(raw16), layer, transposed, p(ah), CA, p(bh), CB, p(gh), p(out), H, Wd, F, st), "s2l_convh_layer")
        assert torch.equal(c32_to_nhwc(out), want), rep
    assert float(ref.abs().max()) > 0 and bool(torch.isfinite(ref).all())
    if gate:
        assert float((ref == 0).float().mean()) > 0.3      # the gate really closed


@pytest.mark.parametrize("layer,transposed,F,H,Wd,gate", [(1, 0, 2, 40, 40, False), (9, 1, 1, 33, 17, True), (6, 0, 2, 37, 53, False), (3, 1, 1, 125, 125, True),
                                                          (8, 0, 2, 500, 500, False), (2, 0, 3, 250, 250, False), (4, 1, 2, 125, 125, False), (7, 0, 1, 16, 32, False),
                                                          (5, 0, 5, 70, 41, False)])
def test_four_wave_and_eight_wave_forms_give_the_same_bits(blobs, dev, layer, transposed, F, H, Wd, gate):
    """s2l_set_unet_half_kernel: 0 = eight waves interleaving loads and MFMAs (default), 1 = four waves, 2 = eight waves in alternating
    roles (gated launches run as 0); same arithmetic in the same order."""
    _, raw, raw16 = blobs
    lib = _abi.load()
    g = torch.Generator(device="cpu").manual_seed(7 * layer + H)
    cin, cout = CONVS[layer]
    if transposed:
        cin, cout = cout, cin
    cat = layer in (6, 8) and not transposed
    CA, CB = (cin // 2, cin // 2) if cat else (cin, 0)
    ah = nhwc_to_c32(torch.randn(F, H, Wd, CA, generator=g).to(torch.bfloat16).to(dev))
    bh = nhwc_to_c32(torch.randn(F, H, Wd, CB, generator=g).to(torch.bfloat16).to(dev)) if CB else None
    gh = nhwc_to_c32(torch.randn(F, H, Wd, cout, generator=g).clamp_min(0).to(torch.bfloat16).to(dev)) if gate else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []

    def run(L):
        out = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
        _abi.check(L.s2l_convh_layer(p(raw16), layer, transposed, p(ah), CA, p(bh), CB, p(gh), p(out), H, Wd, F, st), "s2l_convh_layer")
        return out
    outs.append(run(lib))                                       # the product library: one kernel per job
    assert lib.s2l_set_unet_half_kernel(1) == -5 and lib.s2l_set_unet_half_kernel(2) == -5      # S2L_E_UNSUPPORTED
    rlib = _abi.load_reference()                                # libs2l_hip_ref.so holds the other forms
    try:
        for kind in (0, 1, 2, 0):
            assert rlib.s2l_set_unet_half_kernel(kind) == 0
            outs.append(run(rlib))
        torch.cuda.synchronize()
    finally:
        rlib.s2l_set_unet_half_kernel(0)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert lib.s2l_set_unet_half_kernel(3) == -2


def test_convolution_argument_errors(blobs, dev):
    _, raw, raw16 = blobs
    lib = _abi.load()
    a = torch.zeros(1, 2, 8, 8, 32, dtype=torch.int16, device=dev)
    o = torch.zeros(1, 2, 8, 8, 32, dtype=torch.int16, device=dev)
    assert lib.s2l_convh_layer(p(raw16), 0, 0, p(a), 64, None, 0, None, p(o), 8, 8, 1, None) == -2       # layer 0 is the fp32-input convolution
    assert lib.s2l_convh_layer(p(raw16), 1, 0, p(a), 32, None, 0, None, p(o), 8, 8, 1, None) == -2       # channel count of the layer
    assert lib.s2l_convh_layer(None, 1, 0, p(a), 64, None, 0, None, p(o), 8, 8, 1, None) == -1
    assert lib.s2l_convh_layer(p(raw16), 1, 0, ctypes.c_void_p(a.data_ptr() + 2), 64, None, 0, None, p(o), 8, 8, 1, None) == -3


def rel(a, b):
    return float((a - b).norm() / b.norm())


def cos(a, b):
    return float((a.flatten() @ b.flatten()) / (a.norm() * b.norm()))


@pytest.mark.parametrize("F,fh,fw", [(5, 36, 44), (3, 500, 500)])
def test_frames_in_one_call_equal_one_call_per_frame(dev, F, fh, fw):
    ua, ub = net(dev), net(dev)
    rng = np.random.default_rng(F * fh)
    x = T(rng.random((F, fh, fw, 3), dtype=np.float32)).to(dev)
    d = T(rng.standard_normal((F, fh, fw, 3)).astype(np.float32)).to(dev)
    outs, dxs = [], []
    for f in range(F):
        o, c = ua.forward_train_frames_nhwc(x[f:f + 1], precision="bf16h")
        outs.append(o)
        dxs.append(ua.backward_train_frames(c, d[f:f + 1]))
    o_b, c_b = ub.forward_train_frames_nhwc(x, precision="bf16h")
    dx_b = ub.backward_train_frames(c_b, d)
    assert torch.equal(o_b, torch.cat(outs, 0)) and torch.equal(dx_b, torch.cat(dxs, 0))
    sa, sb = ua.state_dict(), ub.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert int(sb["inc.double_conv.1.num_batches_tracked"]) == 100 + F
    # a second run of the same call: the same bits (no atomics anywhere in the chain)
    uc = net(dev)
    o_c, c_c = uc.forward_train_frames_nhwc(x, precision="bf16h")
    assert torch.equal(o_c, o_b) and torch.equal(uc.backward_train_frames(c_c, d), dx_b)


@pytest.mark.parametrize("F,fh,fw", [(1, 4, 4), (3, 37, 45), (5, 64, 48), (2, 130, 70), (3, 500, 500)])
def test_normalise_inside_the_consuming_convolution_gives_the_same_bits(dev, F, fh, fw):
    """`fuse_norm` (s2l_unet_train_forward_frames_h_fused, the frozen net's forward): BatchNorm + ReLU of a0, a2, a4, a6, a8 folded into the
    convolution that reads them (convh8_norm_asm_kernel normalises the pre-BatchNorm halo tile in LDS: bn_relu_h_kernel's fma / max / round to
    nearest even, zero padding applied to the ACTIVATION) == the two-kernel route, bit for bit: the output, the input gradient (whose backward
    never reads those activations), the running statistics -- at sizes with partial tiles on every side, sub-tile images, three resolutions,
    several frames (each its own statistics group: the table changes at frame boundaries inside a workgroup's tile range)."""
    ua, ub = net(dev), net(dev)
    rng = np.random.default_rng(7 * F + fh)
    x = T(rng.random((F, fh, fw, 3), dtype=np.float32)).to(dev)
    d = T(rng.standard_normal((F, fh, fw, 3)).astype(np.float32)).to(dev)
    o_a, c_a = ua.forward_train_frames_nhwc(x, precision="bf16h", fuse_norm=False)
    o_b, c_b = ub.forward_train_frames_nhwc(x, precision="bf16h", fuse_norm=True)
    assert torch.equal(o_a, o_b), float((o_a - o_b).abs().max())
    assert torch.equal(ua.backward_train_frames(c_a, d), ub.backward_train_frames(c_b, d))
    sa, sb = ua.state_dict(), ub.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    with pytest.raises(s2l._abi.S2LError):      # the fused state has no activations for the weight gradients
        ub.backward_train_frames(c_b, d, want_param_grads=True)
    with pytest.raises(ValueError):
        ub.forward_train_frames_nhwc(x, precision="bf16", fuse_norm=True)
    # the default: fused exactly when the net is frozen
    for p_ in ub.parameters():
        p_.requires_grad = False
    assert ub.forward_train_frames_nhwc(x, precision="bf16h", update_running=False)[1][5] is True
    assert ua.forward_train_frames_nhwc(x, precision="bf16h", update_running=False)[1][5] is False


@pytest.mark.parametrize("F,fh,fw,grads", [(2, 64, 80, False), (2, 130, 70, True), (2, 500, 500, False)])
def test_backward_with_stage_one_from_the_convolution_matches_the_reduction_pass(dev, F, fh, fw, grads):
    """The frozen (and the training) backward take stage 1 of five layers' BatchNorm backward from the input-gradient convolution that produces
    their output gradient (gen_convh8_body.py bstats_block) instead of bn_bwd_reduce_h_kernel's pass over g and z (S2L_NO_CONV_BSTATS=1 restores
    it; read per launch).  Same sums in another order, on a chain whose tensors are bf16: the two agree to a few bf16 roundings of dz (rel. L2
    <= 2e-2, cosine >= 0.9999), and against the fp32-tensor chain (the parity reference of test_chain_against_the_fp32_chain; autograd of
    SimpleUnetLight.py:16-111 in train mode behind it) the convolution's sums are no worse than the pass's."""
    import os
    u, u32 = net(dev), net(dev)
    rng = np.random.default_rng(3 * F + fw)
    x = T(W.synthetic_image((F, fh, fw, 3), 5, "x")).to(dev)      # (the image of test_chain_against_the_fp32_chain: the bf16 chain's own error is ~1e-2 there)
    d = T(rng.standard_normal((F, fh, fw, 3)).astype(np.float32)).to(dev)

    def flat(r):
        return [t.double().flatten() for t in ([r[0]] + [r[1][k] for k in sorted(r[1])] if grads else [r])]
    _, c32 = u32.forward_train_frames_nhwc(x, update_running=False)
    ref = flat(u32.backward_train_frames(c32, d, want_param_grads=True) if grads else u32.backward_train_frames(c32, d))
    res = []
    for off in (False, True):
        if off:
            os.environ["S2L_NO_CONV_BSTATS"] = "1"
        else:
            os.environ.pop("S2L_NO_CONV_BSTATS", None)
        try:
            _, c = u.forward_train_frames_nhwc(x, precision="bf16h", fuse_norm=False, update_running=False)
            res.append(flat(u.backward_train_frames(c, d, want_param_grads=True) if grads else u.backward_train_frames(c, d)))
            torch.cuda.synchronize()
        finally:
            os.environ.pop("S2L_NO_CONV_BSTATS", None)
    assert len(res[0]) == len(res[1]) == len(ref) >= 1
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for a, b, r in zip(res[0], res[1], ref):
        assert bool(torch.isfinite(a).all())
        e_on, e_off = rel(a, r), rel(b, r)
        # (white-noise d: ReLUs within rounding of zero resolve the other way in a bf16 chain -- its own bound against the fp32 chain is 0.4, see
        #  test_chain_against_the_fp32_chain; what matters here: the convolution's sums are no worse than the pass's, and the two agree closely)
        assert e_on <= 1.25 * e_off + 1e-4 and e_on <= 0.4, (e_on, e_off, a.numel())
        assert rel(a, b) <= 2e-2 and float(torch.dot(a, b) / (a.norm() * b.norm())) >= 0.9999, (rel(a, b), e_off)


@pytest.mark.parametrize("fh,fw", [(64, 80), (500, 500)])
def test_chain_against_the_fp32_chain(dev, fh, fw):
    u32, u16, uh = net(dev), net(dev), net(dev)
    x = T(W.synthetic_image((2, fh, fw, 3), 5, "x")).to(dev)
    d = T(np.random.default_rng(2).standard_normal((2, fh, fw, 3)).astype(np.float32)).to(dev)
    o32, c32 = u32.forward_train_frames_nhwc(x)
    o16, c16 = u16.forward_train_frames_nhwc(x, precision="bf16")
    oh, ch = uh.forward_train_frames_nhwc(x, precision="bf16h")
    g32, g16, gh = u32.backward_train_frames(c32, d), u16.backward_train_frames(c16, d), uh.backward_train_frames(ch, d)
    # forward: bf16 operands alone hold 2e-2 here (tests/test_gpu_training_chain.py); rounding every stored tensor to bf16 as well
    # adds its 2^-9 per tensor on top
    assert rel(oh, o32) <= 3.5e-2 and cos(oh, o32) >= 0.9995, (rel(oh, o32), cos(oh, o32), rel(o16, o32))
    assert bool(torch.isfinite(oh).all()) and bool(torch.isfinite(gh).all()) and not torch.equal(oh, o16)
    # input gradient (white-noise d: ReLUs within rounding of zero resolve the other way; the fp32-tensor bf16 chain's bound is 0.3 / 0.97)
    assert cos(gh, g32) >= 0.95 and rel(gh, g32) <= 0.4, (rel(gh, g32), cos(gh, g32), rel(g16, g32), cos(g16, g32))
    sd32, sdh = u32.state_dict(), uh.state_dict()
    for k in sd32:
        if k.endswith("num_batches_tracked"):
            assert int(sd32[k]) == int(sdh[k]) == 102
        elif "running" in k:
            assert rel(sdh[k], sd32[k]) <= 3e-2, (k, rel(sdh[k], sd32[k]))


def test_frozen_train_mode_net_takes_the_half_width_route(dev):
    u, v = net(dev), net(dev)
    for m in (u, v):
        for q in m.parameters():
            q.requires_grad_(False)
    v.half_width_tensors = False
    x = T(W.synthetic_image((3, 40, 56, 3), 7, "x")).to(dev)
    d = torch.ones(3, 40, 56, 3, device=dev)
    o_u, c_u = u.forward_for_backward(x, precision="bf16")
    o_v, c_v = v.forward_for_backward(x, precision="bf16")
    assert c_u[0] == c_v[0] == "train_frames"
    assert c_u[1][0][2][2].dtype == torch.int16 and c_v[1][0][2][2].dtype == torch.float32      # the saved state: bf16 vs fp32 tensors
    assert not torch.equal(o_u, o_v) and rel(o_u, o_v) <= 3e-2
    g_u, g_v = u.backward_to_input(c_u, d), v.backward_to_input(c_v, d)
    assert cos(g_u, g_v) >= 0.9      # (two rounded chains against each other, constant d: 0.94 measured; each is >= 0.94 against fp32)
    # fp32 precision never takes it
    _, c_w = net(dev).requires_grad_(False).forward_for_backward(x, precision="fp32")
    assert c_w[1][0][2][2].dtype == torch.float32
    # several groups under a small memory budget: the same bits as one group
    w = net(dev)
    for q in w.parameters():
        q.requires_grad_(False)
    lib = _abi.load()
    w.train_frames_budget_bytes = 2 * 2 * (int(lib.s2l_unet_train_frames_h_saved_halves(40, 56, 1)) + int(lib.s2l_unet_train_frames_h_work_halves(40, 56, 1)))
    o_w, c_w = w.forward_for_backward(x, precision="bf16")
    assert len(c_w[1]) == 2 and torch.equal(o_w, o_u) and torch.equal(w.backward_to_input(c_w, d), g_u)


@pytest.mark.parametrize("fh,fw,F", [(64, 80, 2), (500, 500, 1)])
def test_eval_mode_chain_against_the_fp32_tensor_chain(dev, fh, fw, F):
    """precision "bf16h" of the frozen EVAL-mode net (s2l_unet_forward_saved_h / s2l_unet_backward_h): the same bf16 operands as
    precision "bf16", bf16 tensors between the kernels.  Against the exact fp32 pair at the bounds of tests/test_gpu_training_chain.py's
    bf16 test (relaxed by the extra storage rounding), against the fp32-tensor bf16 pair, deterministic, and equal on a crop window."""
    u = s2l.SimpleUnetLight().to(dev).eval()
    u.load_state_dict({k[len("post_fusion_unet."):]: T(v) for k, v in W.make_unet_state_dict(0).items()})
    x = T(W.synthetic_image((F, fh, fw, 3), 5, "x")).to(dev)
    d = T(np.random.default_rng(2).standard_normal((F, fh, fw, 3)).astype(np.float32)).to(dev)
    o32, c32 = u.forward_saved_nhwc(x)
    g32 = u.backward_input(c32, d)
    o16, c16 = u.forward_saved_nhwc(x, precision="bf16")
    g16 = u.backward_input(c16, d)
    oh, ch = u.forward_saved_nhwc(x, precision="bf16h")
    gh = u.backward_input(ch, d)
    assert ch[0].dtype == torch.int16 and c16[0].dtype == torch.float32
    assert rel(oh, o32) <= 1.5e-2 and cos(oh, o32) >= 0.9998, (rel(oh, o32), cos(oh, o32), rel(o16, o32))
    assert rel(oh, o16) <= 1e-2 and not torch.equal(oh, o16)
    assert cos(gh, g32) >= 0.97 and rel(gh, g32) <= 0.3, (rel(gh, g32), cos(gh, g32), rel(g16, g32), cos(g16, g32))
    oh2, ch2 = u.forward_saved_nhwc(x, precision="bf16h")
    assert torch.equal(oh2, oh) and torch.equal(u.backward_input(ch2, d), gh)
    if fh == 500:      # a crop evaluated as a window of the full frame (the sync chain's use): interior values are the full frame's bits
        win = (fh, fw, 48, 68)
        crop = x[:, 48:460, 68:432].contiguous()
        ow, cw = u.forward_saved_nhwc(crop, window=win, precision="bf16h")
        assert torch.equal(ow[:, 40:-40, 40:-40], oh[:, 88:420, 108:392])
        dw = torch.zeros_like(d)
        dw[:, 120:380, 140:360] = d[:, 120:380, 140:360]      # a gradient whose cone stays inside the crop
        g_full = u.backward_input(ch, dw)
        g_win = u.backward_input(cw, dw[:, 48:460, 68:432].contiguous())
        assert torch.equal(g_win[:, 40:-40, 40:-40], g_full[:, 88:420, 108:392])
    # the mode-following pair takes this route for an eval-mode net in bf16 precision (and not when switched off)
    _, ctx = u.forward_for_backward(x, precision="bf16")
    assert ctx[0].dtype == torch.int16
    u.half_width_tensors = False
    _, ctx = u.forward_for_backward(x, precision="bf16")
    assert ctx[0].dtype == torch.float32


def test_training_net_on_half_width_tensors(dev):
    """A net that still trains (before `it > 100000`) in bf16 precision: forward_for_backward takes the half-width frames route with
    parameter gradients (s2l_unet_train_backward_frames_h_grads: BatchNorm gradients summed over the frames, 3x3 weight gradients on
    bf16 MFMAs straight from the bf16 planes, first / output layer gradients fp32).  Against the fp32-tensor bf16 chain and the exact fp32
    chain on the same input; F frames in one call == the per-frame gradients summed (to summation-order accuracy)."""
    def fresh():
        return net(dev)
    F, fh, fw = 3, 64, 80
    x = T(W.synthetic_image((F, fh, fw, 3), 5, "x")).to(dev)
    d = T(np.random.default_rng(2).standard_normal((F, fh, fw, 3)).astype(np.float32)).to(dev)
    res = {}
    for name, prec, half in (("fp32", "fp32", True), ("bf16", "bf16", False), ("half", "bf16", True)):
        u = fresh()
        u.half_width_tensors = half
        o, ctx = u.forward_for_backward(x, precision=prec)
        g = {}
        dx = u.backward_to_input(ctx, d, param_grads=g)
        res[name] = (o, dx, g, ctx, u)
    assert res["half"][3][0] == "train_frames_grads" and res["half"][3][1][0][2][2].dtype == torch.int16
    assert res["bf16"][3][1][0][2][2].dtype == torch.float32
    o32, dx32, g32 = res["fp32"][:3]
    oh, dxh, gh = res["half"][:3]
    o16, dx16, g16 = res["bf16"][:3]
    assert rel(oh, o32) <= 3.5e-2 and cos(dxh, dx32) >= 0.95
    assert set(gh) == set(g32) and len(gh) == 32
    for k in g32:      # parameter gradients: those of the rounded forward, like the fp32-tensor bf16 chain's (its bound there: cos >= 0.97)
        assert cos(gh[k], g32[k]) >= 0.95, (k, cos(gh[k], g32[k]), cos(g16[k], g32[k]))
        assert bool(torch.isfinite(gh[k]).all())
    # frames in one call == per-frame calls, gradients summed
    ua, ub = fresh(), fresh()
    ga, gb = {}, {}
    oa, ca = ua.forward_for_backward(x, precision="bf16")
    dxa = ua.backward_to_input(ca, d, param_grads=ga)
    dxs = []
    for f in range(F):
        o1, c1 = ub.forward_for_backward(x[f:f + 1], precision="bf16")
        assert torch.equal(o1, oa[f:f + 1])
        dxs.append(ub.backward_to_input(c1, d[f:f + 1], param_grads=gb))
    assert torch.equal(torch.cat(dxs, 0), dxa)
    for k in ga:
        err = float((ga[k] - gb[k]).abs().max()) / (float(gb[k].abs().max()) + 1e-30)
        assert err <= 5e-5, (k, err)
    for k, v in ua.state_dict().items():
        assert torch.equal(v, ub.state_dict()[k]), k


def test_odd_tiny_and_ragged_sizes(dev):
    """4 x 4 to 501 x 499, odd sizes, one to five frames: the three half-width routes (frozen train mode, training with parameter
    gradients, eval mode) against the exact fp32 chains -- finite, consistent with each other, inside the bf16 bounds."""
    from tools import odd_sizes_half
    notes = []
    assert odd_sizes_half.screen(log=notes.append) == 0, notes


@pytest.mark.parametrize("F,H,Wd,CA,CB,cout", [(1, 5, 7, 64, 0, 64), (2, 37, 53, 64, 0, 128), (1, 3, 130, 128, 128, 128), (3, 19, 64, 64, 64, 64),
                                               (1, 70, 65, 128, 0, 64), (2, 1, 200, 64, 0, 64)])
def test_weight_gradient_kernel_alone_vs_fp64_correlation(dev, F, H, Wd, CA, CB, cout):
    """conv_wgrad_h_kernel (LDS transpose reads, a ring of three input rows down a 64-pixel column, split K) on its own: dW[co][ci][t] of
    bf16 planes against the fp64 correlation of the SAME bf16 values -- ragged widths (one partial chunk, several chunks), heights of one
    to many rows (column changes inside a workgroup's range), several frames, concatenated inputs.  Only the fp32 summation order differs:
    1e-5 of the tensor's largest value.  (The reference's counterpart is autograd's conv2d weight gradient, SimpleUnetLight.py:16-45.)"""
    lib = _abi.load()
    g = torch.Generator(device="cpu").manual_seed(H * 131 + Wd)
    dz = torch.randn(F, H, Wd, cout, generator=g).to(torch.bfloat16)
    a = torch.randn(F, H, Wd, CA, generator=g).to(torch.bfloat16)
    b = torch.randn(F, H, Wd, CB, generator=g).to(torch.bfloat16) if CB else None
    x = torch.cat([a, b], -1) if CB else a
    want = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (cout, CA + CB, 3, 3), dz.permute(0, 3, 1, 2).double(), padding=1)
    dzp, ap, bp = nhwc_to_c32(dz.to(dev)), nhwc_to_c32(a.to(dev)), (nhwc_to_c32(b.to(dev)) if CB else None)
    part = torch.empty(64 * 256 * 128 * 9, dtype=torch.float32, device=dev)
    outs = []
    for _ in range(2):
        dw = torch.full((cout, CA + CB, 9), float("nan"), device=dev)
        _abi.check(lib.s2l_debug_conv_wgrad_h(p(dzp), p(ap), CA, p(bp), CB, cout, p(part), p(dw), H, Wd, F,
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_debug_conv_wgrad_h")
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[1])
    got = outs[0].cpu().double().reshape(cout, CA + CB, 3, 3)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()), float((got - want).abs().max()) / float(want.abs().max())
    # argument errors
    assert lib.s2l_debug_conv_wgrad_h(p(dzp), p(ap), 32, None, 0, cout, p(part), p(outs[0]), H, Wd, F, None) == -2
    assert lib.s2l_debug_conv_wgrad_h(None, p(ap), CA, None, 0, cout, p(part), p(outs[0]), H, Wd, F, None) == -1


@pytest.mark.parametrize("layer,F,H,Wd", [(9, 2, 40, 40), (1, 1, 33, 17), (7, 2, 37, 53), (3, 1, 125, 125), (9, 2, 500, 500), (5, 3, 70, 41), (1, 2, 4, 4)])
def test_input_gradient_convolution_leaves_stage_one_of_the_batchnorm_backward(blobs, dev, layer, F, H, Wd):
    """The input-gradient convolution's backward statistics (csrc/gen_convh8_body.py bstats_block: what the frozen train-mode backward consumes
    instead of bn_bwd_reduce_h_kernel's pass over g and z; autograd's BatchNorm backward of SimpleUnetLight.py:16-40): per tile and channel
    sum g' and sum g' z with g' = fma(z, scale, shift) > 0 ? g : 0 on the STORED bf16 g -- ragged tiles excluded rows / columns -- to fp32
    summation accuracy; the stored tensor is the same bits as the plain launch's (unmasked); two runs: the same partials bit for bit."""
    _, raw, raw16 = blobs
    lib = _abi.load()
    g = torch.Generator(device="cpu").manual_seed(13 * layer + H)
    cin, cout = CONVS[layer]
    dz = nhwc_to_c32((0.25 * torch.randn(F, H, Wd, cout, generator=g)).to(torch.bfloat16).to(dev))
    z_nhwc = torch.randn(F, H, Wd, cin, generator=g).to(torch.bfloat16)
    z = nhwc_to_c32(z_nhwc.to(dev))
    # scale / shift with 8-bit mantissas: fma(z, scale, shift) is then exact in fp32 and the mask is the same in any arithmetic
    sc = (torch.randn(F, cin, generator=g) + 0.3).to(torch.bfloat16).float()
    sh = (0.5 * torch.randn(F, cin, generator=g)).to(torch.bfloat16).float()
    rows = torch.zeros(F, 512)
    rows[:, :cin], rows[:, cin:2 * cin] = sc, sh
    rows = rows.to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.full((F, cin // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
    _abi.check(lib.s2l_convh_layer(p(raw16), layer, 1, p(dz), cout, None, 0, None, p(ref), H, Wd, F, st), "s2l_convh_layer")
    parts = []
    for _ in range(2):
        out = torch.full_like(ref, -1)
        stat = torch.full((F * 1024 * 2 * cin,), float("nan"), device=dev)
        blocks = ctypes.c_int(0)
        _abi.check(lib.s2l_debug_convh_layer_bstats(p(raw16), layer, p(dz), p(z), p(rows), p(out), p(stat), ctypes.byref(blocks), H, Wd, F, st),
                   "s2l_debug_convh_layer_bstats")
        torch.cuda.synchronize()
        assert blocks.value == ((Wd + 15) // 16) * ((H + 31) // 32) and torch.equal(out, ref)
        parts.append(stat[:F * blocks.value * 2 * cin].reshape(F, blocks.value, 2, cin).clone())
    assert torch.equal(parts[0], parts[1]) and bool(torch.isfinite(parts[0]).all())
    gy = c32_to_nhwc(ref).view(torch.bfloat16).double().cpu()          # [F,H,W,cin]
    zd = z_nhwc.double()
    mask = (zd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]) > 0
    gm = gy * mask
    s_ref, q_ref = gm.sum((1, 2)), (gm * zd).sum((1, 2))
    got = parts[0].double().sum(1).cpu()                               # [F,2,cin]
    assert 0.2 < float(mask.double().mean()) < 0.8
    assert float((got[:, 0] - s_ref).abs().max()) <= 2e-5 * float(gm.abs().sum((1, 2)).max())
    assert float((got[:, 1] - q_ref).abs().max()) <= 2e-5 * float((gm * zd).abs().sum((1, 2)).max())


@pytest.mark.parametrize("layer,F,H,Wd", [(1, 2, 40, 40), (2, 1, 33, 17), (6, 2, 37, 53), (3, 1, 125, 125), (9, 2, 500, 500), (8, 3, 70, 41)])
def test_convolution_leaves_its_tiles_batch_statistics(blobs, dev, layer, F, H, Wd):
    """The forward convolution's own per-tile partial sums (csrc/gen_convh8_body.py stats_block: what the train-mode chain's BatchNorm
    consumes instead of a pass over z): summed over the tiles they are the per-frame, per-channel sum and sum of squares of the STORED
    bf16 values -- ragged tiles, rows and columns outside the image excluded -- to fp32 summation accuracy; the stored tensor is the
    same bits as without statistics; two runs give the same partials bit for bit (fixed order, no atomics)."""
    _, raw, raw16 = blobs
    lib = _abi.load()
    g = torch.Generator(device="cpu").manual_seed(11 * layer + H)
    cin, cout = CONVS[layer]
    cat = layer in (6, 8)
    CA, CB = (cin // 2, cin // 2) if cat else (cin, 0)
    ah = nhwc_to_c32(torch.randn(F, H, Wd, CA, generator=g).to(torch.bfloat16).to(dev))
    bh = nhwc_to_c32(torch.randn(F, H, Wd, CB, generator=g).to(torch.bfloat16).to(dev)) if CB else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
    _abi.check(lib.s2l_convh_layer(p(raw16), layer, 0, p(ah), CA, p(bh), CB, None, p(ref), H, Wd, F, st), "s2l_convh_layer")
    parts = []
    for _ in range(2):
        out = torch.full_like(ref, -1)
        stat = torch.full((F * 1024 * 2 * cout,), float("nan"), device=dev)
        blocks = ctypes.c_int(0)
        _abi.check(lib.s2l_debug_convh_layer_stats(p(raw16), layer, p(ah), CA, p(bh), CB, p(out), p(stat), ctypes.byref(blocks), H, Wd, F, st),
                   "s2l_debug_convh_layer_stats")
        torch.cuda.synchronize()
        assert blocks.value == ((Wd + 15) // 16) * ((H + 31) // 32) and torch.equal(out, ref)
        parts.append(stat[:F * blocks.value * 2 * cout].reshape(F, blocks.value, 2, cout).clone())
    assert torch.equal(parts[0], parts[1]) and bool(torch.isfinite(parts[0]).all())
    z = c32_to_nhwc(ref).view(torch.bfloat16).double()                  # [F,H,W,cout]
    s_ref, q_ref = z.sum((1, 2)), (z * z).sum((1, 2))
    got = parts[0].double().sum(1)                                     # [F,2,cout]
    assert float((got[:, 0] - s_ref).abs().max()) <= 2e-5 * float(z.abs().sum((1, 2)).max())
    assert float((got[:, 1] - q_ref).abs().max()) <= 2e-5 * float(q_ref.max())
