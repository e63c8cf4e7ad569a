"""The U-Net's fp32 3x3 convolutions exist as generated gfx950 assembly (csrc/gen_conv_body.py: four tile-end variants) and as
the C++ kernel it replaced (csrc/unet.hip conv3x3_kernel).  Parity with the reference network is tested on whichever runs
(tests/test_gpu_parity.py, test_gpu_training_chain.py: goldens G7 / G13 and the oracle); here: both forms perform the same
arithmetic in the same order, so every output, input gradient and weight gradient is the same bit pattern -- over frame shapes
that hit the tile borders, single-tile frames, exact multiples of 16, fewer tiles than CUs and more, ragged pooled sizes."""
import numpy as np
import pytest
import torch

import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W

pytestmark = pytest.mark.gpu

SHAPES = [(1, 4, 4), (1, 5, 7), (2, 8, 8), (1, 15, 15), (1, 16, 16), (1, 17, 17), (3, 16, 48), (1, 31, 33), (1, 32, 32), (2, 33, 31),
          (1, 47, 65), (1, 65, 63), (5, 20, 36), (2, 128, 130), (1, 257, 63), (1, 4, 200), (1, 200, 4), (7, 44, 52), (24, 40, 40),
          (1, 500, 500)]


@pytest.fixture(scope="module")
def unet():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    dev = torch.device("cuda:0")
    u = s2l.SimpleUnetLight().to(dev).eval()
    u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
    yield u
    assert _abi.load().s2l_set_unet_conv_kernel(0) == 0


def all_outputs(u, x, g):
    res = {"eval": u.forward_nhwc(x)}
    out, ctx = u.forward_saved_nhwc(x)
    res["saved"], res["d_x"] = out, u.backward_input(ctx, g)
    u.train()
    try:
        out, ctx = u.forward_train_nhwc(x, update_running=False)
        dx, grads = u.backward_train(ctx, g)
    finally:
        u.eval()
    res["train"], res["train_d_x"] = out, dx
    res.update({"grad " + k: v for k, v in grads.items()})
    return {k: v.clone() for k, v in res.items()}


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_assembly_and_cpp_convolutions_agree_bit_for_bit(unet, shape):
    lib = _abi.load()
    dev = next(unet.parameters()).device
    x = torch.from_numpy(W.synthetic_image(shape + (3,), 11, "x")).to(dev)
    g = torch.from_numpy(W.synthetic_image(shape + (3,), 12, "x")).to(dev) - 0.5
    assert lib.s2l_set_unet_conv_kernel(0) == 0
    a = all_outputs(unet, x, g)
    assert lib.s2l_set_unet_conv_kernel(1) == 0
    try:
        b = all_outputs(unet, x, g)
    finally:
        assert lib.s2l_set_unet_conv_kernel(0) == 0
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert float(a["eval"].abs().max()) > 0 and float(a["d_x"].abs().max()) > 0


def test_conv_kernel_switch_rejects_other_kinds():
    assert _abi.load().s2l_set_unet_conv_kernel(2) == -2 and _abi.load().s2l_set_unet_conv_kernel(-1) == -2


@pytest.mark.parametrize("shape", [(1, 4, 4), (1, 33, 17), (3, 37, 501), (2, 64, 48), (1, 500, 500), (7, 200, 333), (40, 96, 96),
                                   (16, 160, 256)], ids=lambda s: "x".join(map(str, s)))
def test_split_persistent_kernel_equals_the_one_tile_form(unet, shape):
    """The split-bf16 layers run in a persistent kernel (tiles as one chunk stream per workgroup, two swizzled halo buffers, LDS-DMA
    weights, counted vmcnt waits, one barrier per chunk, the activation leaving through LDS): the same products in the same
    order as conv3x3_bf16_kernel<.., SPLIT, 8>, so the SAME BITS -- over single-tile frames, partial tiles, fewer tiles than
    workgroups and several tiles per workgroup (the streams' tile crossings), repeated (a counted wait that is one short shows
    as run-to-run differences)."""
    F, H, Wd = shape
    dev = next(unet.parameters()).device
    x = torch.from_numpy(np.random.default_rng(H * 1000 + Wd).random((F, H, Wd, 3), dtype=np.float32)).to(dev)
    g = torch.from_numpy(np.random.default_rng(7).standard_normal((F, H, Wd, 3)).astype(np.float32)).to(dev)
    lib = _abi.load()

    def chain():      # the plain-bf16 training chain through the same kernel (launches of >= 4 tiles per workgroup): forward with
        out, ctx = unet.forward_saved_nhwc(x, precision="bf16")      # saved state (pooled copies, fused output layer), gated input gradients
        return out.clone(), unet.backward_input(ctx, g).clone()
    try:
        assert lib.s2l_set_unet_split_kernel(1) == 0
        ref = unet.forward_nhwc(x, precision="split").clone()
        ref_out, ref_dx = chain()
        assert lib.s2l_set_unet_split_kernel(0) == 0
        for _ in range(3):
            assert torch.equal(unet.forward_nhwc(x, precision="split"), ref)
        for _ in range(2):
            out, dx = chain()
            assert torch.equal(out, ref_out) and torch.equal(dx, ref_dx)
    finally:
        lib.s2l_set_unet_split_kernel(0)
    assert float(ref.abs().max()) > 0 and float(ref_dx.abs().max()) > 0
    assert lib.s2l_set_unet_split_kernel(3) == -2


def test_persistent_kernel_random_shapes_race_screen(unet):
    """tools/soak_conv_kernels.py in small: 36 random frame shapes (a third of them few large frames, the rest many small ones, so that
    workgroups own one tile, several tiles, and tile ranges that cross channel tiles and frames), split forward + plain-bf16 saved
    forward + gated input gradient, persistent kernel twice against the one-tile kernels: every output the same bits."""
    dev = next(unet.parameters()).device
    lib = _abi.load()
    rng = np.random.default_rng(123)
    try:
        for it in range(36):
            big = it % 3 == 0
            F = int(rng.integers(1, 4)) if big else int(rng.integers(1, 48))
            H = int(rng.integers(200, 520)) if big else int(rng.integers(4, 140))
            Wd = int(rng.integers(200, 520)) if big else int(rng.integers(4, 140))
            x = torch.rand(F, H, Wd, 3, device=dev)
            d = torch.randn(F, H, Wd, 3, device=dev)
            res = []
            for kind in (1, 0, 0):
                assert lib.s2l_set_unet_split_kernel(kind) == 0
                a = unet.forward_nhwc(x, precision="split").clone()
                o, ctx = unet.forward_saved_nhwc(x, precision="bf16")
                res.append((a, o.clone(), unet.backward_input(ctx, d).clone()))
            for j in range(3):
                assert torch.equal(res[0][j], res[1][j]) and torch.equal(res[0][j], res[2][j]), (F, H, Wd, j)
    finally:
        lib.s2l_set_unet_split_kernel(0)


@pytest.mark.parametrize("shape", [(1, 4, 4), (1, 16, 16), (1, 33, 17), (2, 20, 36), (3, 64, 64), (1, 131, 77), (3, 37, 501), (2, 500, 500),
                                   (40, 96, 96), (7, 200, 333)], ids=lambda s: "x".join(map(str, s)))
def test_generated_assembly_split_convolution_equals_the_cpp_kernel(unet, shape):
    """conv16_asm_kernel (csrc/gen_conv16_body.py: 32 x 16 tiles, four waves, operands and accumulators in AGPRs, one chunk stream
    per workgroup, stores through an LDS transpose; selector 2) performs conv3x3_split_kernel's arithmetic in its order: the
    split-mode forward is THE SAME BITS -- single-tile frames, partial tiles on every border, one tile per workgroup and many
    (the streams' tile crossings and the store staging that aliases a halo buffer), virtual-concat layers, both pooled layers --
    and it is repeatable (a counted wait that is one short shows as run-to-run differences)."""
    F, H, Wd = shape
    dev = next(unet.parameters()).device
    x = torch.from_numpy(np.random.default_rng(H * 977 + Wd).random((F, H, Wd, 3), dtype=np.float32)).to(dev)
    lib = _abi.load()
    assert lib.s2l_set_unet_split_kernel(0) == 0
    ref = unet.forward_nhwc(x, precision="split").clone()      # the product library's kernel
    assert lib.s2l_set_unet_split_kernel(2) == -5              # ... which does not hold the assembly form (S2L_E_UNSUPPORTED)
    with _abi.reference_kernels() as rlib:                     # libs2l_hip_ref.so (-DS2L_WITH_REFERENCE_KERNELS)
        try:
            assert rlib.s2l_set_unet_split_kernel(2) == 0
            for _ in range(3):
                assert torch.equal(unet.forward_nhwc(x, precision="split"), ref)
        finally:
            rlib.s2l_set_unet_split_kernel(0)
    assert float(ref.abs().max()) > 0


def test_split_mode_out_of_range_operands_stay_finite(unet):
    """The split speed mode carries every operand as two IEEE halves: valid for |x| < 65504 (include/s2l_hip.h).  Beyond that range
    the parts SATURATE -- hi by v_cvt_pkrtz_f16_f32, the residual x - hi by a clamp before its conversion -- so that the result is
    wrong but FINITE (an inf part would make the MFMA's sum NaN); in range the mode keeps its accuracy."""
    dev = next(unet.parameters()).device
    x = torch.from_numpy(np.random.default_rng(3).random((1, 40, 56, 3), dtype=np.float32)).to(dev)
    ok = unet.forward_nhwc(x, precision="split")
    ref = unet.forward_nhwc(x)
    assert float((ok - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    big = unet.forward_nhwc(x * 3e6, precision="split")           # activations of ~1e6 after the first layers
    assert bool(torch.isfinite(big).all())
