"""The opt-in split-half speed mode of the clip renderer (`TalkingFace.render_clip(precision="split")`,
`s2l_render_lip_split`, csrc/render16.hip): the same function as the exact kernel -- tf_nerf.py:225-285 over every pixel of
every frame, inference.py:140-159 -- on v_mfma_f32_16x16x32_f16 with every operand carried as hi + lo halves.

Bars (written here, per the north star: RMSE <= 1e-4 / PSNR >= 50 dB; VERDICT round 3 asked for RMSE <= 1e-5 against
`oracle.render_clip`): RMSE <= 4e-6, max |err| <= 4e-5, PSNR >= 105 dB against the CPU oracle and the reference's own golden
frames (G3) on outputs of RMS ~0.4 -- the exact kernel sits at 6.5e-7 on the same inputs.  The exact fp32 kernel remains the
default; this file is the only place that turns the speed mode on (besides bench.py's `extra.render_split`).
"""
import numpy as np
import pytest
import torch

from oracle import s2l_oracle as O
from speech2lip_amd import _abi, weights as W
from tests.test_gpu_parity import make_model

pytestmark = pytest.mark.gpu
T = torch.from_numpy
RMSE, MAXERR, PSNR = 4e-6, 4e-5, 105.0


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def sd():
    return O.to_sd(W.make_state_dict(0, "he"))


def near(got, ref):
    got, ref = got.detach().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    r, mx, ps = O.rmse(got, ref), float((got.double() - ref.double()).abs().max()), O.psnr(got, ref)
    assert r <= RMSE and mx <= MAXERR and ps >= PSNR, f"rmse {r:.3e} max {mx:.3e} psnr {ps:.1f} dB"
    return r


def test_split_render_golden_frames_of_the_reference(golden, dev):
    """G3: frames the REFERENCE rendered (tools/make_goldens.py)."""
    g = golden("g3_rgb.npz")
    for h, w, idx in [(16, 16, 7), (64, 64, 7), (12, 20, 597)]:
        m = make_model(dev, h, w)
        out = m.render_clip(T(g["window"])[None].to(dev), [idx], h, w, precision="split")
        near(out.reshape(-1, 3), g[f"frame_{h}x{w}_idx{idx}"])


@pytest.mark.parametrize("h,w,f", [(16, 16, 12), (12, 20, 7), (64, 64, 3), (5, 7, 11), (1, 1, 1), (2, 3, 200), (24, 24, 30)])
def test_split_render_vs_oracle(sd, dev, h, w, f):
    m = make_model(dev, h, w)
    win = T(W.synthetic_audio(f, seed=5).astype(np.float32))
    idx = [(37 * i) % 4001 for i in range(f)]
    with torch.no_grad():
        ref = O.render_clip(sd, win, idx, h, w)
    got = m.render_clip(win.to(dev), idx, h, w, precision="split")
    near(got, ref)
    exact = m.render_clip(win.to(dev), idx, h, w)
    assert O.rmse(got.cpu(), exact.cpu()) <= RMSE and not torch.equal(got, exact) or h * w * f < 8


def test_split_render_config2_size(sd, dev):
    """BASELINE config 2: 96x96, 1000 frames (83 x 12 + 4: a partial last frame tile).  Sampled frames against the oracle, the
    whole clip against the exact kernel, determinism, and frame independence (any sub-clip renders to the same bits)."""
    h = w = 96
    f = 1000
    m = make_model(dev, h, w)
    win = T(W.synthetic_audio(f, seed=1).astype(np.float32)).to(dev)
    idx = torch.arange(f, device=dev)
    clip = m.render_clip(win, idx, h, w, precision="split")
    assert torch.equal(clip, m.render_clip(win, idx, h, w, precision="split"))
    exact = m.render_clip(win, idx, h, w)
    assert O.rmse(clip.cpu(), exact.cpu()) <= RMSE and float((clip - exact).abs().max()) <= MAXERR
    with torch.no_grad():
        for k in (0, 517, 999):
            near(clip[k], O.render_clip(sd, win[k:k + 1].cpu(), [k], h, w)[0])
    for a, b in ((100, 160), (996, 1000), (7, 8)):
        assert torch.equal(m.render_clip(win[a:b], idx[a:b], h, w, precision="split"), clip[a:b]), (a, b)


@pytest.mark.parametrize("h,w,frames", [(96, 96, (1, 5, 16)), (12, 20, (1, 13)), (5, 7, (1, 3))])
def test_split_render_tile_shapes_give_the_same_bits(dev, h, w, frames):
    """The three tile shapes (shared with the exact kernel: s2l_set_render_shape) render every frame to the same bits in the
    split mode too: per sample column the MFMA sequence is the same."""
    lib = _abi.load()
    m = make_model(dev, h, w)
    F = max(frames)
    win = T(W.synthetic_audio(F, seed=21).astype(np.float32)).to(dev)
    idx = torch.arange(100, 100 + F, device=dev)
    try:
        _abi.check(lib.s2l_set_render_shape(1), "s2l_set_render_shape")
        ref = m.render_clip(win, idx, h, w, precision="split")
        for f in frames:
            for mode in (2, 3, 0):
                _abi.check(lib.s2l_set_render_shape(mode), "s2l_set_render_shape")
                got = m.render_clip(win[:f], idx[:f], h, w, precision="split")
                assert torch.equal(got, ref[:f]), (h, w, f, mode, float((got - ref[:f]).abs().max()))
    finally:
        lib.s2l_set_render_shape(0)


def test_split_render_follows_weight_updates_and_rejects_bad_arguments(dev):
    m = make_model(dev, 16, 16)
    win = T(W.synthetic_audio(2, seed=9).astype(np.float32)).to(dev)
    a = m.render_clip(win, [0, 1], 16, 16, precision="split").clone()
    with torch.no_grad():
        m.output_linear.weight.mul_(0.5)
        m.output_linear.bias.mul_(0.5)
    b = m.render_clip(win, [0, 1], 16, 16, precision="split")
    assert float((b - 0.5 * a).abs().max()) <= 1e-5            # the bf16-style pack is rebuilt with the fp32 blob
    with pytest.raises(ValueError):
        m.render_clip(win, [0, 1], 16, 16, precision="bf16")
    lib = _abi.load()
    null = None
    assert lib.s2l_render_lip_split(null, null, null, null, null, null, null, 16, 1, null) == -1
    assert lib.s2l_render_lip_split(null, null, null, null, null, null, null, 16, 0, null) == 0
    assert lib.s2l_pack_render16(null, null, null) == -1 and lib.s2l_render16_packed_halves() == 113 * 8192
