"""BASELINE.json's configurations at their stated sizes, through the C-ABI on the GPU.

  C2  96x96, 1000 synthetic audio frames (the bench workload): frames incl. the partial last frame tile vs the oracle,
      sub-clip bit-equality.
  C3  128x128 lip + paste/head-pose-warp composite into 500x500, per-clip constants (table path, XCD regions),
      then the U-Net on the same frames, against the oracle chain.
  C4  frame indices near 40 000 (the 40k-frame clip of the 8-GPU configuration).
  C5  the bf16 training step at batch 64, 96x96: vs the fp32 parity-mode step, a teacher-forced sample of its rows vs
      the CPU emulation, and -- at a size the oracle's autograd finishes in seconds -- bf16 gradients directly vs the oracle.

Sizes the oracle cannot finish in seconds are covered through size-independent properties (determinism, frame
independence: any sub-clip renders to the same bits) plus sampled frames.  Tolerances as tests/test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch

import speech2lip_amd as s2l
from oracle import s2l_oracle as O
from speech2lip_amd import weights as W
from tests.test_gpu_parity import close, make_model

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def sd():
    return O.to_sd(W.make_state_dict(0, "he"))


def test_config2_1000_frames_96(sd, dev):
    """1000 = 83 x 12 + 4: the last frame tile of every pixel group is partial."""
    h = w = 96
    f = 1000
    m = make_model(dev, h, w)
    win = T(W.synthetic_audio(f, seed=1).astype(np.float32)).to(dev)
    idx = torch.arange(f, device=dev)
    clip = m.render_clip(win, idx, h, w)
    assert torch.equal(clip, m.render_clip(win, idx, h, w))                         # deterministic
    assert torch.equal(m.render_clip(win[990:], idx[990:], h, w), clip[990:])       # 10-frame tail: other tiling, same bits
    assert torch.equal(m.render_clip(win[492:504], idx[492:504], h, w), clip[492:504])
    with torch.no_grad():
        for k in (0, 499, 995, 996, 999):
            ref = O.render_clip(sd, win[k:k + 1].cpu(), [k], h, w)[0]
            close(clip[k], ref)
            assert O.psnr(clip[k].cpu(), ref) >= 90.0


@pytest.mark.parametrize("h,w,frames", [(96, 96, (1, 5, 12, 16, 25)), (64, 64, (1, 16)), (5, 7, (1, 3, 13)), (1, 1, (1, 2)), (12, 20, (1, 7))])
def test_render_tile_shapes_give_the_same_bits(sd, dev, h, w, frames):
    """The tile shapes of the renderer (16 px x 12 frames, 192 px x 1 frame, 64 px x 1 frame: csrc/gen_render_body.py; and the feature-split
    tile, 16 px x 1 frame with the four waves owning 64 features each: csrc/gen_render_fs_body.py, mode 4)
    render every frame to the same bits: each sample column sees the same MFMA sequence, only the tiling differs.  Includes
    images smaller than one tile block (5x7 = 3 pixel groups of a 12-group block: table rows are clamped, stores masked), the
    auto-selection, and -- for one frame per call, the reference's mode (inference.py:140-159) -- the CPU oracle."""
    from speech2lip_amd import _abi
    lib = _abi.load()
    m = make_model(dev, h, w)
    F = max(frames)
    win = T(W.synthetic_audio(F, seed=21).astype(np.float32)).to(dev)
    idx = torch.arange(100, 100 + F, device=dev)
    try:
        _abi.check(lib.s2l_set_render_shape(1), "s2l_set_render_shape")
        ref = m.render_clip(win, idx, h, w)
        for f in frames:
            for mode in (2, 3, 4, 0):
                _abi.check(lib.s2l_set_render_shape(mode), "s2l_set_render_shape")
                got = m.render_clip(win[:f], idx[:f], h, w)
                assert torch.equal(got, ref[:f]), (h, w, f, mode, float((got - ref[:f]).abs().max()))
    finally:
        lib.s2l_set_render_shape(0)
    assert lib.s2l_set_render_shape(5) != 0
    with torch.no_grad():
        o = O.render_clip(sd, win[:1].cpu(), [100], h, w)[0]
    close(m.render_clip(win[:1], idx[:1], h, w)[0], o)


def test_config4_frame_indices_near_40000(sd, dev):
    """The time encoding at the far end of a 40 000-frame clip (sin/cos arguments up to 4e4): golden G1 pins the
    oracle there on the CPU; this is the device path."""
    h = w = 96
    m = make_model(dev, h, w)
    idx = list(range(39990, 40000))
    win = T(W.synthetic_audio(len(idx), seed=4).astype(np.float32))
    out = m.render_clip(win.to(dev), idx, h, w)
    with torch.no_grad():
        for k in (0, 4, 9):
            ref = O.render_clip(sd, win[k:k + 1], [idx[k]], h, w)[0]
            close(out[k], ref)
    # the general-row path at the same index
    feat = m.audio_merge_forward(win[9:10].to(dev))
    rows = torch.cat([s2l.get_coords(w, h, dev), feat.expand(h * w, -1)], -1)
    close(m.rgb_forward(rows, time_pts=torch.tensor([39999])), out[9].reshape(-1, 3).cpu())


@pytest.fixture(scope="module")
def one_rank_rccl(dev):
    import os
    import socket
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(dev)
    created = not dist.is_initialized()
    if created:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    yield dev
    if created:
        dist.destroy_process_group()


def test_config4_rank_blocks_at_full_size_from_audio_npy_on_disk(sd, one_rank_rccl, tmp_path):
    """BASELINE config 4 as far as ONE GPU can take it: the clip's wire format -- a 40 000-window float64 `audio.npy`
    (deepspeech_features.py:65-75; read and cast as someones_lip_dataset.py:246 does) -- goes to disk, comes back through
    `speech2lip_amd.data` (the `--use_new_audio` split, someones_lip_dataset.py:156-161), and the 5 000-frame blocks that rank 0 and
    rank 7 of an 8-GPU job own (sharded.shard_range) are rendered through the product entry `sharded.render_clip_sharded` on a
    one-rank RCCL group: determinism, sub-block bit equality across the tile-shape boundary (5 000 = 416 x 12 + 8), the oracle on
    sampled frames including index 39 999, and the 8-bit gather.  What is left untested of config 4 is the N > 1 collective itself."""
    import os
    from speech2lip_amd import sharded
    from tools.benchlib import write_synthetic_dataset
    dev = one_rank_rccl
    h = w = 96
    N, G = 40_000, 8
    root = str(tmp_path / "may_face_crop_lip")
    write_synthetic_dataset(root, 3, FH=160, FW=176, lh=h, lw=w, x0=40, y0=30, train=False, workers=2)
    wire = W.synthetic_audio(N, seed=1)
    assert wire.dtype == np.float64 and wire.shape == (N, 16, 29)
    assert not wire[0, :4].any() and not wire[-1, -4:].any() and wire[N // 2].all()      # the clip-end zero padding of the windowing
    np.save(os.path.join(root, "audio_test", "audio.npy"), wire)
    assert os.path.getsize(os.path.join(root, "audio_test", "audio.npy")) >= N * 16 * 29 * 8
    ds = s2l.SomeonesLipClip(root, "test", s2l.may_config(h, w, data_path=root))
    assert len(ds) == N and ds.aud_features.dtype == np.float64 and (ds.lip_h, ds.lip_w) == (h, w)
    audio = T(ds.aud_features.astype(np.float32))                      # HOST tensor: 74 MB, every rank holds the whole clip's windows
    idx = torch.arange(N)
    m = make_model(dev, h, w)
    first7 = None
    for r in (0, G - 1):
        first, count, per = sharded.shard_range(N, r, G)
        assert (first, count, per) == (r * 5000, 5000, 5000)
        a_r, i_r = audio[first:first + count], idx[first:first + count]
        block = sharded.render_clip_sharded(m, a_r, i_r, h, w, force_collective=True)
        again = sharded.render_clip_sharded(m, a_r, i_r, h, w, force_collective=True, n_chunks=4, quantum=48)
        torch.cuda.synchronize()
        assert block.shape == (count, h, w, 3) and block.dtype == torch.float32
        assert torch.equal(block, again)                                # deterministic, and chunked gathers land in frame order
        # sub-blocks: other launch sizes, other tile shapes (12-frame tiles | the 8-frame tail | single frames), same bits
        for lo, hi in ((0, 12), (4984, 5000), (4992, 5000), (4999, 5000), (2500, 2600), (1234, 1235)):
            sub = m.render_clip(a_r[lo:hi].to(dev), i_r[lo:hi].to(dev), h, w)
            assert torch.equal(sub, block[lo:hi]), (r, lo, hi)
        u8 = sharded.render_clip_sharded(m, a_r, i_r, h, w, gather="u8", force_collective=True)
        assert u8.dtype == torch.uint8 and torch.equal(u8, s2l.to8b(block))
        with torch.no_grad():
            for k in ((0, 7, 2499) if r == 0 else (0, 4991, 4999)):
                ref = O.render_clip(sd, a_r[k:k + 1], [int(i_r[k])], h, w)[0]
                close(block[k], ref)
                assert O.psnr(block[k].cpu(), ref) >= 90.0
        if r == G - 1:
            assert int(i_r[-1]) == 39_999
            first7 = block[:4].clone()
        del block, again, u8
    # the two blocks are different frames of one clip: same audio seed, so only a wrong offset could make them equal
    assert not torch.equal(first7, m.render_clip(audio[:4].to(dev), idx[:4].to(dev), h, w))


def _config3_inputs(dev, F, seed=0):
    h = w = 128
    FH = FW = 500
    x0, y0 = 186, 300
    face = T(W.synthetic_image((1, FH, FW, 3), 2, "face"))
    gt = T(W.synthetic_image((F, FH, FW, 3), 3, "gt"))
    mask = torch.zeros(1, FH, FW, 3)
    mask[:, y0:y0 + h, x0:x0 + w] = 1
    # a soft band like the JPEG-decoded mask of the dataset (someones_lip_dataset.py:72): the blend must lerp
    mask[:, y0:y0 + 3, x0:x0 + w] = T(W.synthetic_image((3, w, 3), 5, "soft"))
    coord = T(W.synthetic_warp_coords(F, FH, FW, seed=4 + seed))
    return h, w, FH, FW, x0, y0, face, gt, mask, coord


def test_config3_composite_500_with_128_lip(sd, dev):
    """128x128 lip pasted at (186,300) into 500x500 faces, 8 frames, per-clip face/mask -> fused-table path with the
    XCD-region block mapping at its real size."""
    F = 8
    h, w, FH, FW, x0, y0, face, gt, mask, coord = _config3_inputs(dev, F)
    m = make_model(dev, h, w)
    win = T(W.synthetic_audio(F, seed=1).astype(np.float32))
    idx = list(range(100, 100 + F))
    lip = m.render_clip(win.to(dev), idx, h, w)
    with torch.no_grad():
        ref_lip = O.render_clip(sd, win, idx, h, w)
    close(lip, ref_lip)
    new, can = m.composite_clip(lip, face.to(dev), gt.to(dev), mask.to(dev), x0, y0, coord.to(dev), want_canonical=True)
    ref_new, ref_can = O.composite(lip.cpu(), face.expand(F, -1, -1, -1), gt, mask.expand(F, -1, -1, -1), x0, y0, coord)
    assert torch.equal(can.cpu(), ref_can)                       # elementwise: bit-exact
    close(new, ref_new, 1e-6, 2e-5)                              # bilinear weights of fp32 coordinates at 500x500
    assert O.psnr(new.cpu(), ref_new) >= 100.0
    # batched == single frame, table path == per-frame-constants path
    one, _ = m.composite_clip(lip[5:6], face.to(dev), gt[5:6].to(dev), mask.to(dev), x0, y0, coord[5:6].to(dev))
    assert torch.equal(one[0], new[5])
    slow, _ = m.composite_clip(lip, face.expand(F, -1, -1, -1).contiguous().to(dev), gt.to(dev),
                               mask.expand(F, -1, -1, -1).contiguous().to(dev), x0, y0, coord.to(dev))
    assert torch.equal(slow, new)
    # every pixel outside the warped rectangle is the observed frame, untouched
    changed = (new.cpu() != gt).any(-1)
    assert 0.05 < float(changed.float().mean()) < 0.25


def test_config3_chain_render_composite_unet_vs_oracle(sd, dev):
    """The whole inference output of inference.py:140-178 for two frames at config-3 size: render -> composite -> U-Net
    through the drop-in `post_fusion2_onlylip`, against the oracle chain."""
    F = 2
    h, w, FH, FW, x0, y0, face, gt, mask, coord = _config3_inputs(dev, F, seed=1)
    m = make_model(dev, h, w)
    m.load_state_dict({k: T(v) for k, v in W.make_unet_state_dict(0).items()})
    win = T(W.synthetic_audio(F, seed=2).astype(np.float32))
    lip = m.render_clip(win.to(dev), [7, 8], h, w)
    recon, new, can = m.post_fusion2_onlylip(lip, face.to(dev), gt.to(dev), mask.to(dev), x0, y0, coord.to(dev))
    usd = O.to_sd(W.make_unet_state_dict(0))
    with torch.no_grad():
        ref_lip = O.render_clip(sd, win, [7, 8], h, w)
        ref_new, ref_can = O.composite(ref_lip, face.expand(F, -1, -1, -1), gt, mask.expand(F, -1, -1, -1), x0, y0, coord)
        ref_recon = O.unet_forward(usd, ref_new)
    close(new, ref_new, 2e-6, 3e-5)
    close(can, ref_can, 1e-6, 1e-4)          # the lip itself carries the MLP's fp32 re-association here
    close(recon, ref_recon, 1e-5, 1e-4)
    assert O.psnr(recon.cpu(), ref_recon) >= 90.0


def test_config5_bf16_step_batch64_96(dev):
    """BASELINE config 5 at its stated size: 64 frames, 96x96 (9 216 row tiles of 256, 20 GB of saved state) in bf16,
    against the fp32 parity-mode step on the same inputs (itself pinned to oracle autograd + the reference's gradients)."""
    h = w = 96
    B = 64
    m = make_model(dev, h, w)
    rng = np.random.default_rng(64)
    win = T(W.synthetic_audio(B, seed=17).astype(np.float32)).to(dev)
    idx = [3 + 7 * b for b in range(B)]
    u01 = [float(v) for v in rng.random(B)]
    targets = T(rng.random((B, h * w, 3), dtype=np.float32)).to(dev)
    loss16, g16, aux16 = s2l.LipTrainStep(m, h, w, precision="bf16").loss_and_grads(win, idx, targets, u01)
    pred16 = aux16["pred"].cpu()
    g16 = {k: v.cpu() for k, v in g16.items()}
    del aux16
    torch.cuda.empty_cache()
    loss32, g32, aux32 = s2l.LipTrainStep(m, h, w).loss_and_grads(win, idx, targets, u01)
    assert abs(float(loss16) - float(loss32)) <= 5e-3 * abs(float(loss32))
    assert O.rmse(pred16, aux32["pred"].cpu()) <= 1e-2
    assert set(g16) == set(g32)
    for k in g32:
        a, b = g16[k].double().flatten(), g32[k].cpu().double().flatten()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert rel <= 5e-2 and cos >= 0.999, f"{k}: rel {rel:.3e} cos {cos:.6f}"
    # frames are independent in the forward: frame 40 of the batch == the same frame alone (bit for bit)
    _, _, aux1 = s2l.LipTrainStep(m, h, w, precision="bf16").loss_and_grads(win[40:41], idx[40:41], targets[40:41], u01[40:41])
    assert torch.equal(aux1["pred"].cpu()[0], pred16[40])


def test_config5_bf16_kernels_batch64_sampled_rows(dev):
    """Forward and backward bf16 kernels at the batch-64 row count (2.36 M rows): 64-row tiles sampled from the whole range
    against the step-wise CPU emulation of the same arithmetic (as test_bf16_kernels_multi_tile_workgroups at 2 frames)."""
    from speech2lip_amd import _abi
    from speech2lip_amd.talking_face import _ptr, _stream
    from tests import bf16_util as U
    from tests.test_gpu_parity import _bf16_inputs
    m, lib, x, N = _bf16_inputs(dev, 96, 96, 64)
    Np = int(lib.s2l_bf16_rows_padded(N))
    assert Np // 256 == 9216
    hT = torch.zeros(8 * Np * 256, dtype=torch.int16, device=dev)
    dzT = torch.zeros(8 * Np * 256, dtype=torch.int16, device=dev)
    masks = torch.zeros(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
    rgb, dxa = torch.empty(N, 3, device=dev), torch.empty(N, 64, device=dev)
    pb, pf = m.packed_weights_bf16(), m.packed_weights()
    drgb = (torch.randn(N, 3, generator=torch.Generator().manual_seed(3)) * 1e-3).to(dev)
    _abi.check(lib.s2l_train_forward_bf16(_ptr(pb), _ptr(pf), _ptr(_bf16_inputs.last[0]), _ptr(hT), _ptr(masks), _ptr(rgb), N, _stream()), "fwd")
    _abi.check(lib.s2l_train_backward_bf16(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dzT), _ptr(dxa), N, _stream()), "bwd")
    tiles = [0, 255 * 4 + 1, 256 * 4 + 2, 5000 * 4 + 3, 9215 * 4, Np // 64 - 1]
    rows = torch.cat([torch.arange(t * 64, t * 64 + 64) for t in tiles])

    def sample(buf, nl):     # rows `rows` of every layer without converting the whole 9.7 GB buffer
        lay = Np * 256
        out = []
        for L in range(nl):
            per = []
            for t in tiles:
                g32 = (t * 64) // 32
                seg = buf[L * lay + g32 * 8192: L * lay + (g32 + 2) * 8192]
                per.append(U.tiles_to_rows(seg, 1, 64)[0])
            out.append(torch.cat(per))
        return torch.stack(out)

    h_d, g_d = sample(hT, 8), sample(dzT, 8)
    mk_all = masks.reshape(8, Np // 64, 256)
    mk = torch.cat([U.masks_to_rows(mk_all[:, t].reshape(-1), 64) for t in tiles], dim=1)        # [8, rows, 256]
    sd_ = O.to_sd(W.make_state_dict(0, "he"))
    fold = U.folded_from_blob(pf)
    with torch.no_grad():
        h_tf = U.forward_teacher_forced(sd_, x[rows.to(dev)].cpu(), h_d, fold)
        g_e, dxa_e = U.backward_teacher_forced(sd_, drgb[rows.to(dev)].cpu(), mk, g_d, fold)
    for L in range(8):
        U.assert_bf16_close(h_d[L], h_tf[L], f"h{L}")
        d = (g_d[L] - g_e[L]).abs()
        scale = float(g_e[L].abs().max())
        assert bool((d <= 2.0 ** -7 * g_e[L].abs() * 1.01 + 1e-6 * scale).all()), (L, float(d.max()), scale)
    assert bool((mk == (h_d > 0)).all())
    assert float((dxa.cpu()[rows] - dxa_e).abs().max()) <= 1e-5 * float(dxa_e.abs().max())


@pytest.mark.parametrize("h,w,B", [(12, 20, 3)])
def test_bf16_step_gradients_directly_vs_oracle_autograd(dev, h, w, B):
    """bf16 mode one hop from the oracle (not via the HIP fp32 mode): loss, prediction and all 42 gradients against torch
    autograd through the CPU oracle.  bf16 carries 8 significant bits: relative L2 <= 5 %, cosine >= 0.999."""
    from tests.test_gpu_parity import _oracle_grads
    np_sd = W.make_state_dict(0, "he")
    m = make_model(dev, h, w)
    rng = np.random.default_rng(21)
    win = T(W.synthetic_audio(B, seed=17).astype(np.float32))
    idx = [5 + 11 * b for b in range(B)]
    u01 = [0.37, 0.81, 0.05][:B]
    targets = T(rng.random((B, h * w, 3), dtype=np.float32))
    ref_loss, ref, ref_pred = _oracle_grads(np_sd, win, idx, targets, u01, h, w, weight=1.0)
    loss, g, aux = s2l.LipTrainStep(m, h, w, precision="bf16").loss_and_grads(win.to(dev), idx, targets.to(dev), u01)
    assert abs(float(loss) - ref_loss) <= 5e-3 * abs(ref_loss)
    assert O.rmse(aux["pred"].cpu(), ref_pred) <= 1e-2
    assert set(ref) <= set(g)
    for k in ref:
        a, b = g[k].cpu().double().flatten(), ref[k].double().flatten()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert rel <= 5e-2 and cos >= 0.999, f"{k}: rel {rel:.3e} cos {cos:.6f}"


def test_parity_on_weights_moved_by_training(dev):
    """Every other parity test runs on the seeded G0 weights.  Here the weights are first MOVED by 40 Adam steps of the bf16
    training step (lr 1e-3 on a fixed target image: activations saturate differently, the output range grows), then the
    renderer, the general-row path and the fp32 training gradients are checked against the oracle ON THOSE WEIGHTS."""
    from tests.test_gpu_parity import _oracle_grads
    h, w, B = 24, 32, 4
    m = make_model(dev, h, w).train()
    opt = torch.optim.Adam([p for n, p in m.named_parameters() if not n.startswith("coord_linears")], lr=1e-3)
    rng = np.random.default_rng(77)
    win = T(W.synthetic_audio(B, seed=31).astype(np.float32)).to(dev)
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    img = np.stack([0.5 + 0.5 * np.sin(9 * xx + k) * np.cos(7 * yy - k) for k in range(3)], -1).reshape(1, h * w, 3)
    targets = T(np.repeat(img, B, 0).astype(np.float32) * 3.0 - 1.0).to(dev)          # outside [0,1] on purpose
    step = s2l.LipTrainStep(m, h, w, precision="bf16")
    first = None
    for it in range(40):
        loss, g, _ = step.loss_and_grads(win, list(range(B)), targets, [float(v) for v in rng.random(B)])
        first = float(loss) if first is None else first
        s2l.training.apply_grads(m, g)
        opt.step()
    assert float(loss) < 0.5 * first                                                    # it did train
    np_sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items() if k in W.make_state_dict(0, "he")}
    moved = max(float(np.abs(np_sd[k] - v).max()) for k, v in W.make_state_dict(0, "he").items())
    assert moved > 0.02
    sd = O.to_sd(np_sd)
    m.eval()
    idx = [3, 50, 999, 12345]
    with torch.no_grad():
        ref = O.render_clip(sd, win.cpu(), idx, h, w)
    out = m.render_clip(win, idx, h, w)
    assert float(ref.abs().max()) > 1.0                                                 # a different magnitude regime than G0's
    scale = float(ref.pow(2).mean().sqrt())
    assert O.rmse(out.cpu(), ref) <= 2e-5 * scale and float((out.cpu() - ref).abs().max()) <= 2e-4 * scale
    feat = m.audio_merge_forward(win[1:2])
    rows = torch.cat([s2l.get_coords(w, h, dev), feat.detach().expand(h * w, -1)], -1)
    with torch.no_grad():
        close(m.rgb_forward(rows, time_pts=50), ref[1].reshape(-1, 3), 2e-5 * scale, 2e-4 * scale)
    # gradients of the fp32 parity step on the moved weights
    u01 = [0.3, 0.9]
    ref_loss, ref_g, _ = _oracle_grads(np_sd, win[:2].cpu(), [7, 8], targets[:2].cpu(), u01, h, w)
    loss, g, _ = s2l.LipTrainStep(m.train(), h, w).loss_and_grads(win[:2], [7, 8], targets[:2], u01)
    assert abs(float(loss) - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss))
    for k in ref_g:
        sc = float(ref_g[k].abs().max()) + 1e-12
        assert float((g[k].cpu() - ref_g[k]).abs().max()) <= 3e-4 * sc, k


def test_render_clip_bits_are_pinned(dev):
    """The renderer has been rewritten three times (C++ with compiler-scheduled loads, C++ on the LDS-DMA ring, generated
    assembly) with the arithmetic -- operands, accumulation order, roundings -- unchanged: the sha256 of the 1000-frame 96x96 clip
    of tools/ab_render.py is the same for all of them.  A schedule change must keep it; an arithmetic change must say so here."""
    import hashlib
    from tools.benchlib import make_model as bench_model
    m = bench_model(dev, 96, 96)
    for frames, digest in ((24, "cac76c6f37cf20de"), (1000, "4eb292b93e4ac3c7")):
        audio = T(W.synthetic_audio(frames, 1).astype(np.float32)).to(dev)
        with torch.no_grad():
            clip = m.render_clip(audio, torch.arange(frames, device=dev), 96, 96)
        assert hashlib.sha256(clip.cpu().numpy().tobytes()).hexdigest()[:16] == digest, frames


# ------------------------------------------------------------------------------------------------ composite geometry (property)
def test_composite_random_geometry_property(dev):
    """SURVEY.md §4's property test for the paste + warp composite: random frame sizes, lip sizes, lip offsets (incl. boxes that
    touch the frame border on every side), both pad modes, expand_lip_mask on / off, per-frame and per-clip constants, F = 1..3,
    coordinates that leave [-1, 1] -- every geometry the reference evaluates must equal the oracle (tf_nerf.py:320-386), INCLUDING
    lip boxes that leave the face frame (F.pad's negative amounts crop the lip, :343-350) and rectangles whose python slice wraps
    (:362; golden G17 pins both against the reference itself); a box entirely outside the frame makes the reference's F.pad
    raise, and must raise S2L_E_GEOMETRY here instead of rendering something."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from speech2lip_amd import _abi

    @st.composite
    def cases(draw):
        FH, FW = draw(st.integers(8, 70)), draw(st.integers(8, 90))
        lh, lw = draw(st.integers(1, min(24, FH))), draw(st.integers(1, min(30, FW)))
        edge = draw(st.sampled_from(["free", "free", "left", "top", "right", "bottom", "corner", "over", "outside"]))
        x0 = {"left": 0, "right": FW - lw, "corner": FW - lw, "outside": draw(st.sampled_from([-lw - 2, FW + 2, -lw, FW]))}.get(
            edge, draw(st.integers(-lw + 1, FW) if edge == "over" else st.integers(-2, FW - lw + 2)))
        y0 = {"top": 0, "bottom": FH - lh, "corner": FH - lh}.get(
            edge, draw(st.integers(-lh + 1, FH) if edge == "over" else st.integers(-2, FH - lh + 2)))
        return dict(FH=FH, FW=FW, lh=lh, lw=lw, x0=x0, y0=y0, may=draw(st.booleans()), expand=draw(st.booleans()),
                    F=draw(st.integers(1, 3)), per_clip=draw(st.booleans()), seed=draw(st.integers(0, 2 ** 16)))

    seen = {"ok": 0, "geometry": 0, "cropped": 0}

    @settings(max_examples=80, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(cases())
    def run(c):
        rng = np.random.default_rng(c["seed"])
        FH, FW, lh, lw, x0, y0, F = c["FH"], c["FW"], c["lh"], c["lw"], c["x0"], c["y0"], c["F"]
        path = "dataset/may_face_crop_lip" if c["may"] else "dataset/someone_else"
        m = make_model(dev, lh, lw, path=path)
        m.expand_lip_mask = c["expand"]
        nc = 1 if c["per_clip"] else F
        lip = T(rng.random((F, lh, lw, 3), dtype=np.float32))
        face = T(rng.random((nc, FH, FW, 3), dtype=np.float32))
        gt = T(rng.random((F, FH, FW, 3), dtype=np.float32))
        mask = T(rng.random((nc, FH, FW, 3), dtype=np.float32))
        coord = T((rng.random((F, FH, FW, 2), dtype=np.float32) * 2.6 - 1.3))
        ox, oy = (x0, y0) if c["may"] else (x0 - 1, y0 - 1)
        outside = ox + lw < 0 or oy + lh < 0 or ox > FW or oy > FH              # F.pad would have to crop more than the lip has
        args = (lip.to(dev), face.to(dev), gt.to(dev), mask.to(dev), x0, y0, coord.to(dev))
        if outside:
            with pytest.raises(_abi.S2LError, match="geometry|GEOMETRY"):
                m.composite_clip(*args)
            with pytest.raises(RuntimeError):                                    # ... as the reference's F.pad does (oracle = the same call)
                O.composite(lip[:1], face[:1], gt[:1], mask[:1], x0, y0, coord[:1], pad_mode=O.PAD_MODE_MAY if c["may"] else O.PAD_MODE_DEFAULT)
            seen["geometry"] += 1
            return
        if ox < 0 or oy < 0 or ox + lw > FW or oy + lh > FH:
            seen["cropped"] += 1
        new, can = m.composite_clip(*args, want_canonical=True)
        with torch.no_grad():
            ref = [O.composite(lip[f:f + 1], face[min(f, nc - 1)][None], gt[f:f + 1], mask[min(f, nc - 1)][None], x0, y0, coord[f:f + 1],
                               pad_mode=O.PAD_MODE_MAY if c["may"] else O.PAD_MODE_DEFAULT, expand_lip_mask=c["expand"]) for f in range(F)]
        ref_new, ref_can = torch.cat([r[0] for r in ref]), torch.cat([r[1] for r in ref])
        assert torch.equal(can.cpu(), ref_can), c                                   # merged_canonical is bit-exact
        d = (new.cpu() - ref_new).abs().amax(-1)
        # a coordinate within rounding of a pixel boundary of the expanded rectangle may fall on the other side: whole pixels, rarely
        assert int((d > 1e-5).sum()) <= max(2, d.numel() // 2000), (c, int((d > 1e-5).sum()), float(d.max()))
        if F >= 2 and c["per_clip"]:      # the clip fast path (span kernel) gives the one-pixel kernel's bits
            one, _ = m.composite_clip(args[0][:1], args[1], args[2][:1], args[3], x0, y0, args[6][:1])
            assert torch.equal(one[0], new[0]), c
        seen["ok"] += 1

    run()
    assert seen["ok"] >= 40 and seen["geometry"] >= 3 and seen["cropped"] >= 8, seen


def test_composite_edge_geometry_golden_g17(golden, dev):
    """The reference's own outputs for lip boxes that leave the face frame (F.pad crops, tf_nerf.py:343-350) and rectangles whose
    python slice wraps or is clipped (:362) -- tools/make_golden_edges.py; both pad modes and the obama2 rule; the one-pixel
    kernel (one frame) and the clip fast path (the same frame twice with per-clip constants)."""
    g = golden("g17_composite_edges.npz")
    face, gt, mask, coord = (T(g[k]).to(dev) for k in ("face", "gt", "mask", "coord"))
    shown = {}
    for name in [str(n) for n in g["names"]]:
        lip = T(g[f"{name}/lip"]).to(dev)
        x0, y0, path = int(g[f"{name}/x0"]), int(g[f"{name}/y0"]), str(g[f"{name}/path"])
        m = make_model(dev, lip.shape[1], lip.shape[2], path=path)
        new, can = m.composite_clip(lip, face, gt, mask, x0, y0, coord, want_canonical=True)
        assert torch.equal(can.cpu(), T(g[f"{name}/merged_canonical"])), name          # elementwise: bit-exact
        close(new, g[f"{name}/merged_new"], 1e-6, 2e-6)
        two, _ = m.composite_clip(lip.expand(2, -1, -1, -1).contiguous(), face, gt.expand(2, -1, -1, -1).contiguous(), mask, x0, y0,
                                  coord.expand(2, -1, -1, -1).contiguous())
        assert torch.equal(two[0], new[0]) and torch.equal(two[1], new[0]), name         # span kernel + merged-box table
        shown[name] = float((new != gt).any(-1).float().mean())
        assert abs(shown[name] - float(g[f"{name}/shown"])) <= 2e-3, (name, shown[name])
        # the lip gradient follows the same geometry: finite, zero wherever the forward shows no warped pixel
        d_lip = m.composite_backward_lip(torch.ones_like(new), face, mask, x0, y0, coord, lip.shape[1], lip.shape[2])
        assert torch.isfinite(d_lip).all() and (shown[name] > 0 or float(d_lip.abs().max()) == 0.0)
    assert shown["default_rect_wraps"] == 0.0 and shown["obama2_rect_inside"] > 0.1      # the wrapped slice is EMPTY, as in the reference
    assert bool(g["outside_raises"])
    from speech2lip_amd import _abi
    m = make_model(dev, 16, 24)
    with pytest.raises(_abi.S2LError, match="geometry|GEOMETRY"):
        m.composite_clip(torch.zeros(1, 16, 24, 3, device=dev), face, gt, mask, 70, 10, coord)


def test_render_and_crop_resize_random_sizes_property(sd, dev):
    """SURVEY.md §4's property tests, the remaining two axes: (H, W, F) of the lip crop -- any size, any clip length, any tile
    shape the renderer picks -- against the oracle on sampled frames, and random crop boxes (incl. boxes that leave the frame, which
    python slicing clips, training.py:541-543) of crop + Resize([96,96]) and its adjoint against autograd through the oracle."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from speech2lip_amd import autograd as A

    @settings(max_examples=12, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.integers(1, 40), st.integers(1, 40), st.integers(1, 30), st.integers(0, 2 ** 16))
    def render(h, w, F, seed):
        m = make_model(dev, h, w)
        win = T(W.synthetic_audio(F, seed=seed % 97).astype(np.float32))
        idx = [seed % 1000 + 3 * k for k in range(F)]
        out = m.render_clip(win.to(dev), idx, h, w)
        assert out.shape == (F, h, w, 3)
        with torch.no_grad():
            for k in sorted({0, F // 2, F - 1}):
                close(out[k], O.render_clip(sd, win[k:k + 1], [idx[k]], h, w)[0])

    @settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.integers(4, 70), st.integers(4, 70), st.integers(0, 2 ** 16), st.sampled_from([(96, 96), (17, 23), (5, 4)]))
    def resize(H, Wd, seed, size):
        rng = np.random.default_rng(seed)
        x0, y0 = int(rng.integers(0, Wd - 1)), int(rng.integers(0, H - 1))
        x2, y2 = int(rng.integers(x0 + 1, Wd + 12)), int(rng.integers(y0 + 1, H + 12))        # may leave the frame: clipped
        x = T(rng.random((2, H, Wd, 3), dtype=np.float32))
        x_o = x.clone().requires_grad_(True)
        ref = O.crop_resize(x_o, (x0, y0, x2, y2), size)
        d = T(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
        (ref * d).sum().backward()
        x_d = x.to(dev).requires_grad_(True)
        got = A.crop_resize(x_d, (x0, y0, x2, y2), size, 0)
        close(got, ref.detach(), 2e-7, 1e-6)
        (got * d.to(dev)).sum().backward()
        assert float((x_d.grad.cpu() - x_o.grad).abs().max()) <= 3e-6 * max(1.0, float(x_o.grad.abs().max()))

    render()
    resize()
