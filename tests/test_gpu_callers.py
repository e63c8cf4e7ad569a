"""The callers either side of the hot path as the reference has them, un-bound from the host:
the drop-in Trainer in bf16 precision, `hole_noise="device"`, `Trainer.train_steps` (K frames per optimisation step) against
the reference's own step (goldens G11 / G14) and against K single `train_step` calls, the byte-wide loader (`from8b`,
`ClipStreamer`) against `SomeonesLipClip.load`, and the pipelined file writer against `write_frames`."""
import os

import numpy as np
import pytest
import torch

import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from tests.test_gpu_training_chain import _g11_device, _patched_draws, full_model, relerr

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def syncnet(dev):
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: T(v) for k, v in W.make_syncnet_state_dict(0).items()}, strict=True)
    return net


def _cfg(m):
    return {**m.cfg, "training": {**m.cfg["training"], "use_canonical_depth_loss_photo_v2": False, "use_perceptual_loss": False,
                                  "use_sync_contrastive_loss": True, "stage": "stage1", "batch_rays": 16 * 24}}


def _late_model(dev):
    m = full_model(dev, 16, 24).train()
    m.post_fusion_unet.eval()                                              # train.py:188-197
    for p in m.post_fusion_unet.parameters():
        p.requires_grad = False
    return m


def test_from8b_is_the_readers_conversion(dev):
    """s2l_from8b == (u8 / 255.0).astype(float32) (someones_lip_dataset.py:196-217), every byte value, ragged length."""
    u8 = np.concatenate([np.arange(256, dtype=np.uint8), np.random.default_rng(0).integers(0, 256, 1003, dtype=np.uint8)])
    got = s2l.from8b(T(u8).to(dev)).cpu().numpy()
    assert np.array_equal(got, (u8 / 255.0).astype(np.float32))
    with pytest.raises(s2l._abi.S2LError):
        s2l.from8b(T(u8))


def test_train_steps_one_frame_reproduces_the_reference_step(golden, syncnet, dev):
    """Trainer.train_stage1_frames (= train_steps without training.py:150's model.train(), as train_stage1 is to train_step) with K = 1 on the G11 batch (it > 100000: frozen U-Net, sync window) and on the G14 batch (the U-Net
    trains): the loss dictionary and the gradients the REFERENCE's train_stage1 produced, at train_stage1's own tolerances; the
    random draws are consumed in train_step's order (six eps draws, the coin, two noise fields)."""
    g, data, eps, _, face = _g11_device(golden, dev)
    m = _late_model(dev)
    tr = s2l.Trainer(m, torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0), cfg=_cfg(m), syncnet=syncnet, use_syncloss=True)
    restore = _patched_draws(eps, face["hole_noise"], dev)
    try:
        loss_rgb, loss = tr.train_stage1_frames([data], it=100001)
    finally:
        eq, fq = restore()
    assert not eq and not fq
    assert abs(float(loss["loss"]) - float(g["loss"])) <= 3e-6 and abs(float(loss["loss_sync"]) - float(g["loss_sync"])) <= 2e-6
    assert abs(float(loss_rgb) - float(g["loss_rgb"])) <= 2e-6
    params = dict(m.named_parameters())
    for key in g:
        if key.startswith("g_") and key != "g_pts5_cols":
            assert relerr(params[key[2:]].grad, g[key]) <= 5e-4, (key, relerr(params[key[2:]].grad, g[key]))
    # G14: it = 50000, train-mode U-Net trained with the MLP; its parameter gradients arrive under post_fusion_unet.*
    e = golden("g14_stage1_early.npz")
    m = full_model(dev, 16, 24).train()
    tr = s2l.Trainer(m, torch.optim.SGD(m.parameters(), lr=0.0), cfg=_cfg(m), syncnet=syncnet, use_syncloss=True)
    restore = _patched_draws(e["eps"], face["hole_noise"], dev)
    try:
        _, loss = tr.train_stage1_frames([dict(data, rgb_face_ori=T(e["rgb_face_ori"]))], it=50000)
    finally:
        eq, fq = restore()
    assert not fq and "loss_sync" not in loss          # (the window's five draws are not made before it > 100000)
    assert abs(float(loss["loss"]) - float(e["loss"])) <= 5e-6
    params = dict(m.named_parameters())
    n_unet = 0
    for key in e:
        if key.startswith("g_"):
            got, ref = params[key[2:]].grad.cpu(), T(e[key])
            assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), key
            n_unet += key[2:].startswith("post_fusion_unet.")
    assert n_unet > 0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_steps_K_frames_equal_K_single_calls(golden, syncnet, dev, precision):
    """K = 3 frames in one train_steps call against three train_step calls at the same weights (lr = 0): the per-step loss is the
    mean of the three single-call losses and the gradient their mean -- to summation-order accuracy in fp32 (the kernels and the
    per-frame BatchNorm statistics groups are the same), to bf16 accuracy in bf16 (tile boundaries move with the batch)."""
    g, data, eps, _, face = _g11_device(golden, dev)
    rng = np.random.default_rng(5)
    frames = []
    for k in range(3):
        d = dict(data)
        d["index"] = int(data["index"]) + 3 * k
        d["rgb"] = T(rng.random(tuple(data["rgb"].shape), dtype=np.float32))
        d["audio"] = T(W.synthetic_audio(1, seed=20 + k).astype(np.float32))
        d["rgb_face_ori"] = T(rng.random(tuple(data["rgb_face_ori"].shape), dtype=np.float32))
        frames.append(d)
    draws = [[0.1 * (k + 1) + 0.05 * j for j in range(6)] for k in range(3)]
    holes = [(torch.randn(1, *data["rgb_face_ori"].shape[1:3], generator=torch.Generator().manual_seed(40 + k)),
              torch.randn(1, *data["rgb_face_ori"].shape[1:3], generator=torch.Generator().manual_seed(50 + k))) for k in range(3)]

    def trainer():
        m = _late_model(dev)
        return m, s2l.Trainer(m, torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0), cfg=_cfg(m), syncnet=syncnet,
                              use_syncloss=True, precision=precision)
    m1, t1 = trainer()
    singles, grads = [], []
    for k in range(3):
        restore = _patched_draws(draws[k], (holes[k][0].to(dev), holes[k][1].to(dev)), dev)
        try:
            _, loss = t1.train_step(frames[k], it=100001)
        finally:
            restore()
        singles.append(float(loss["loss"]))
        grads.append({n: p.grad.clone() for n, p in m1.named_parameters() if p.grad is not None})
    mK, tK = trainer()
    # one call: the draws of frame 0, then frame 1, then frame 2 (train_step's order inside each frame)
    import random
    eq = [v for d in draws for v in d]
    fq = [h.cpu()[:, None].repeat(1, 3, 1, 1) for pair in holes for h in pair]
    # train_steps draws per frame: u01 main, coin, two fields, five window draws -> the patched queues must interleave the same way
    order = []
    for k in range(3):
        order += [draws[k][0]] + draws[k][1:]
    saved = (torch.rand, torch.randn, random.random)
    torch.rand = lambda *a, **kw: torch.full((1,), order.pop(0), device=dev)
    torch.randn = lambda *a, **kw: fq.pop(0)
    random.random = lambda: 0.9
    try:
        _, lossK = tK.train_steps(frames, it=100001)
    finally:
        torch.rand, torch.randn, random.random = saved
    assert not order and not fq
    tol = 2e-5 if precision == "fp32" else 2e-2
    assert abs(float(lossK["loss"]) - sum(singles) / 3) <= tol * max(1.0, abs(sum(singles) / 3))
    pK = dict(mK.named_parameters())
    for name in ("output_linear.weight", "pts_linears.3.weight", "pts_linears.5.weight", "fc_audio.weight", "encoder_conv.0.weight"):
        mean = sum(gr[name] for gr in grads) / 3
        assert relerr(pK[name].grad, mean.cpu()) <= (1e-4 if precision == "fp32" else 6e-2), (name, relerr(pK[name].grad, mean.cpu()))


def test_trainer_bf16_and_device_noise(golden, syncnet, dev):
    """Trainer(precision="bf16") -- train_step through the bf16 MLP kernels and the half-width U-Net chain -- stays within bf16
    accuracy of the fp32 step on the G11 batch; hole_noise="device" draws the fields on the GPU (the CPU generator is not
    touched) and punches about half of the face pixels' holes like the host stream does."""
    g, data, eps, _, face = _g11_device(golden, dev)
    m = _late_model(dev)
    tr = s2l.Trainer(m, torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0), cfg=_cfg(m), syncnet=syncnet, use_syncloss=True,
                     precision="bf16")
    restore = _patched_draws(eps, face["hole_noise"], dev)
    try:
        _, loss = tr.train_stage1(data, it=100001)      # (G11 is train_stage1 itself: the frozen U-Net in eval mode)
    finally:
        restore()
    assert abs(float(loss["loss"]) - float(g["loss"])) <= 2e-2 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    for key in ("g_output_linear.weight", "g_pts_linears.0.weight"):
        a, b = params[key[2:]].grad.flatten().double().cpu(), T(g[key]).flatten().double()
        assert float(a @ b / (a.norm() * b.norm())) >= 0.995, key
    with pytest.raises(ValueError):
        s2l.Trainer(m, None, cfg=_cfg(m), precision="fp16")
    # device noise
    m.hole_noise = "device"
    state = torch.get_rng_state()
    n1, n2 = m.draw_hole_noise(data["rgb_face_ori"].to(dev))
    assert torch.equal(torch.get_rng_state(), state) and n1.is_cuda and n1.shape == data["rgb_face_ori"].shape[:3]
    assert 0.4 < float((n1 < 1e-6).float().mean()) < 0.6 and not torch.equal(n1, n2)
    tr2 = s2l.Trainer(m, torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0), cfg=_cfg(m), syncnet=syncnet, use_syncloss=True,
                      hole_noise="device")
    _, loss2 = tr2.train_step(data, it=100001)
    assert np.isfinite(float(loss2["loss"]))
    m.hole_noise = "bogus"
    with pytest.raises(ValueError):
        m.draw_hole_noise(data["rgb_face_ori"].to(dev))


def test_clip_streamer_and_frame_writer_equal_the_serial_forms(dev, tmp_path):
    """ClipStreamer (thread-pool decode into pinned bytes, H2D on a side stream, s2l_from8b on the device) yields the tensors
    SomeonesLipClip.load returns, bit for bit, over ragged batches; FrameWriter writes the files write_frames writes."""
    from tools.benchlib import write_synthetic_dataset
    root = str(tmp_path / "may_face_crop_lip")
    write_synthetic_dataset(root, 30, FH=40, FW=56, lh=8, lw=12, x0=20, y0=22, train=False, workers=4)
    cfg = s2l.may_config(8, 12, data_path=root)
    ds = s2l.SomeonesLipClip(root, "train", cfg)
    n = len(ds)
    assert n == 27 and (ds.lefttop_x, ds.lefttop_y) == (20, 22)
    for mode in ("thread", "process"):      # decoders in this process's threads | in child processes writing shared memory (PIL holds the GIL)
        st = s2l.ClipStreamer(ds, dev, batch=8, first=3, count=21, workers=3, mode=mode)
        seen = 0
        for clip in st:
            ref = ds.load(dev, 3 + seen, len(clip.names))
            for f in ("audio", "index", "coord", "rgb_face_ori", "rgb_face_zero", "mask_lip_canonical"):
                assert torch.equal(getattr(clip, f), getattr(ref, f)), (mode, f)
            assert clip.names == ref.names and (clip.lip_lefttop_x, clip.height, clip.width) == (ref.lip_lefttop_x, ref.height, ref.width)
            seen += len(clip.names)
        st.close()
        assert seen == 21
    assert s2l.ClipStreamer(ds, dev, batch=8).mode == "thread"      # (short clips: the child processes' start-up would not pay)
    frames = torch.rand(11, 40, 56, 3, device=dev)
    names = ["%05d" % (k + 1) for k in range(11)]
    s2l.write_frames(frames, names, str(tmp_path / "a"))
    wr = s2l.FrameWriter(str(tmp_path / "b"), workers=3, depth=2)
    for s0 in range(0, 11, 4):
        wr.submit(s2l.to8b(frames[s0:s0 + 4]), names[s0:s0 + 4])
    wr.close()
    for nm in names:
        assert open(tmp_path / "a" / (nm + ".jpg"), "rb").read() == open(tmp_path / "b" / (nm + ".jpg"), "rb").read(), nm


def test_clip_streamer_abandoned_mid_clip_leaves_nothing_behind(dev, tmp_path):
    """An exception half way through a clip (what a failing kernel call or a full disk does to inference.py:140-178's loop): the
    streamer's shared-memory blocks go back to the process's bounded cache (the next streamer takes the same blocks: no new segment, no
    new mapping in the decode workers), whether it is closed or only collected; blocks beyond the cache's bound are detached and
    unlinked; once the process-wide pool is shut down no segment and no child process of this test is left; a worker that dies is
    replaced instead of poisoning the pool; the validity promise of a yielded batch holds for work queued one iteration late."""
    import gc
    import os
    import psutil
    from speech2lip_amd import data as D
    from tools.benchlib import write_synthetic_dataset
    root = str(tmp_path / "may_face_crop_lip")
    write_synthetic_dataset(root, 40, FH=40, FW=56, lh=8, lw=12, x0=20, y0=22, train=False, workers=4)
    ds = s2l.SomeonesLipClip(root, "train", s2l.may_config(8, 12, data_path=root))
    me = psutil.Process()
    from multiprocessing import resource_tracker
    resource_tracker.ensure_running()          # (python's own helper process for shared memory: it stays for the life of the interpreter)
    D.shutdown_decode_workers()                # (a clean slate: earlier tests of this process left workers and cached blocks)
    before = {c.pid for c in me.children(recursive=True)}
    shm = lambda name: os.path.exists("/dev/shm/" + name.lstrip("/"))

    class Boom(RuntimeError):
        pass

    with pytest.raises(Boom):
        with s2l.ClipStreamer(ds, dev, batch=4, workers=3, mode="process") as st:
            names = [b.name for b in (st.shm_frames + st.shm_coords)]
            assert len(names) == 6 and all(shm(n) for n in names)
            for k, clip in enumerate(st):
                if k == 3:
                    raise Boom()
    cached = lambda: sorted(b.name for v in D._SHM_CACHE.values() for b in v)
    assert st.closed and cached() == sorted(names) and D._cached_bytes() <= D._SHM_CACHE_MAX
    workers = [w.pid for w in D._DECODE_WORKERS]
    assert len(workers) == 3
    st.close()                               # idempotent
    assert cached() == sorted(names)
    with pytest.raises(RuntimeError):
        next(iter(st))

    # garbage collection alone releases an un-closed streamer -- and it had taken the SAME blocks: nothing new was created or mapped
    st = s2l.ClipStreamer(ds, dev, batch=4, workers=3, mode="process")
    assert sorted(b.name for b in (st.shm_frames + st.shm_coords)) == sorted(names) and cached() == []
    it = iter(st)
    next(it)
    del it, st
    gc.collect()
    assert cached() == sorted(names)
    # beyond the cache's bound a closed streamer's blocks are detached from the workers and unlinked
    old_max, D._SHM_CACHE_MAX = D._SHM_CACHE_MAX, 0
    try:
        with s2l.ClipStreamer(ds, dev, batch=6, workers=3, mode="process") as big:      # (another batch size: other block sizes, new blocks)
            extra = [b.name for b in (big.shm_frames + big.shm_coords)]
            assert not set(extra) & set(names) and sum(len(c.names) for c in big) == len(ds)
        assert not any(shm(n) for n in extra)
        for pid in workers:
            with open(f"/proc/{pid}/maps") as f:
                assert not [ln for ln in f if any(n.lstrip("/") in ln for n in extra)], pid
    finally:
        D._SHM_CACHE_MAX = old_max

    # a decode worker that dies is replaced; the clip still comes out right
    psutil.Process(D._DECODE_WORKERS[0].pid).kill()
    with s2l.ClipStreamer(ds, dev, batch=8, workers=3, mode="process") as st:
        seen = 0
        for clip in st:
            ref = ds.load(dev, seen, len(clip.names))
            assert torch.equal(clip.rgb_face_ori, ref.rgb_face_ori) and torch.equal(clip.coord, ref.coord)
            seen += len(clip.names)
    assert seen == len(ds) and len(D._DECODE_WORKERS) == 3 and all(w.poll() is None for w in D._DECODE_WORKERS)

    # validity: batch k may still be read by work queued during iteration k+1 (a consumer one batch behind)
    with s2l.ClipStreamer(ds, dev, batch=4, workers=3, mode="thread") as st:
        prev, sums, refs, seen = None, [], [], 0
        for clip in st:
            if prev is not None:
                torch.cuda._sleep(20_000_000)                               # a lagging consumer stream
                sums.append((prev.rgb_face_ori.double().sum(), prev.coord.double().sum()))
            refs.append(ds.load(dev, seen, len(clip.names)))
            seen += len(clip.names)
            prev = clip
        sums.append((prev.rgb_face_ori.double().sum(), prev.coord.double().sum()))
    torch.cuda.synchronize()
    for (a, b), ref in zip(sums, refs):
        assert float(a) == float(ref.rgb_face_ori.double().sum()) and float(b) == float(ref.coord.double().sum())

    D.shutdown_decode_workers()
    assert not ({c.pid for c in me.children(recursive=True)} - before)
    assert cached() == [] and not any(shm(n) for n in names)                         # the cache goes with the workers that map it
    with s2l.ClipStreamer(ds, dev, batch=8, first=0, count=8, workers=2, mode="process") as st:      # the pool comes back on demand
        assert sum(len(c.names) for c in st) == 8
    D.shutdown_decode_workers()
    assert not ({c.pid for c in me.children(recursive=True)} - before)


@pytest.mark.parametrize("it", [50000, 100001])
def test_fused_step_option_routes_train_step_through_the_fused_engine(golden, syncnet, dev, it):
    """Trainer(precision="bf16", fused_step=True).train_step(frame) == Trainer(precision="bf16").train_steps([frame]): the same
    generators, draws and kernels, both sides of it = 100000 -- losses bit for bit, gradients up to the composite adjoint's float
    atomics (csrc/composite.hip: the last bit of d lip is not fixed, as for ATen's grid_sample backward); a loss set the fused
    engine does not implement keeps the autograd route."""
    import random
    _, data, _, _, _ = _g11_device(golden, dev)

    def run(fused):
        m = _late_model(dev) if it > 100000 else full_model(dev, 16, 24).train()
        opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0)
        tr = s2l.Trainer(m, opt, cfg=_cfg(m), syncnet=syncnet, use_syncloss=True, precision="bf16", hole_noise="device", fused_step=fused)
        torch.manual_seed(7)
        random.seed(7)
        out = tr.train_step(data, it=it) if fused else tr.train_steps([data], it=it)
        return out, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, tr

    (la, da), ga, tra = run(True)
    (lb, db), gb, _ = run(False)
    assert isinstance(la, float) and la == float(lb)
    assert set(da) == set(db) and all(float(da[k]) == float(db[k]) for k in da if k != "rgb_window")
    assert set(ga) == set(gb) and len(ga) >= 42
    for k in ga:
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-6 * float(gb[k].abs().max()), k
    assert tra._fused_step_covers()
    tra.cfg = {**tra.cfg, "training": {**tra.cfg["training"], "use_canonical_depth_loss_photo_v2": True}}
    assert not tra._fused_step_covers()


def test_fused_adam_is_torch_adam_in_one_launch(dev):
    """speech2lip_amd.FusedAdam (s2l_adam_step: every tensor of the group in one launch) against torch.optim.Adam -- the optimizer the
    reference builds (train.py:166) -- over ragged tensor sizes, several steps, with and without weight decay: parameters and both
    moments to a few ulps (torch's kernels may contract a multiply-add the library is built not to), the same `state_dict()` layout
    (a state written by either loads into the other and continues identically), and the NaN flags of check_weights
    (src/common.py:56-64): a parameter that holds a NaN BEFORE a step is reported for that step."""
    sizes = [(1,), (5, 7), (4096,), (4097,), (256, 259), (3, 64, 3, 3), (12289,)]
    for wd in (0.0, 0.01):
        g = torch.Generator(device=dev).manual_seed(11)
        init = [torch.randn(s, device=dev, generator=g) for s in sizes]
        pa = [torch.nn.Parameter(t.clone()) for t in init]
        pb = [torch.nn.Parameter(t.clone()) for t in init]
        oa = torch.optim.Adam(pa, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        ob = s2l.FusedAdam(pb, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        for step in range(1, 6):
            grads = [torch.randn(s, device=dev, generator=g) * (0.1 if step % 2 else 3.0) for s in sizes]
            for p, q, gr in zip(pa, pb, grads):
                p.grad, q.grad = gr.clone(), gr.clone()
            versions = [q._version for q in pb]
            oa.step()
            ob.step()
            assert all(q._version > v for q, v in zip(pb, versions))      # an in-place update autograd (and the packed-weight caches) can see
            for k, (p, q) in enumerate(zip(pa, pb)):
                tol = 4e-7 * float(p.detach().abs().max()) + 1e-9
                assert float((p.detach() - q.detach()).abs().max()) <= tol, (wd, step, k)
                for key in ("exp_avg", "exp_avg_sq"):
                    a, b = oa.state[p][key], ob.state[q][key]
                    assert float((a - b).abs().max()) <= 4e-7 * float(a.abs().max()) + 1e-12, (wd, step, k, key)
                assert float(oa.state[p]["step"]) == float(ob.state[q]["step"]) == step
            assert not any(bad for _, bad in ob.nan_report())
        # state dicts are interchangeable: torch's state into the fused optimizer and back, then one more identical step
        sa, sb = oa.state_dict(), ob.state_dict()
        assert sa["param_groups"][0].keys() >= {"lr", "betas", "eps", "weight_decay"} and set(sb["state"][0]) == set(sa["state"][0])
        ob2 = s2l.FusedAdam(pb, lr=1e-3, weight_decay=wd)
        ob2.load_state_dict(sa)
        oa2 = torch.optim.Adam(pa, lr=1e-3, weight_decay=wd)
        oa2.load_state_dict(sb)
        for p, q in zip(pa, pb):
            q.data.copy_(p.data)
            p.grad = q.grad = torch.ones_like(p)
        oa2.step()
        ob2.step()
        for p, q in zip(pa, pb):
            assert float((p.detach() - q.detach()).abs().max()) <= 4e-7 * float(p.detach().abs().max()) + 1e-9
    # the NaN report: the value BEFORE the update is what counts (the reference checks before optimizer.step(), training.py:572)
    pb[3].data[17] = float("nan")
    for q in pb:
        q.grad = torch.zeros_like(q)
    ob2.step()
    report = ob2.nan_report()
    assert [bad for _, bad in report] == [k == 3 for k in range(len(pb))] and report[3][0] is pb[3]
    with pytest.raises(s2l._abi.S2LError):
        cpu = torch.nn.Parameter(torch.zeros(3))
        cpu.grad = torch.zeros(3)
        s2l.FusedAdam([cpu]).step()
    with pytest.raises(NotImplementedError):
        s2l.FusedAdam(pb, amsgrad=True)


def test_pending_step_fused_adam_and_device_prefetch_equal_the_synchronous_route(golden, syncnet, dev, tmp_path):
    """The three things `train_steps` gained for the loop that is not allowed to idle the device (train.py:173-199 with K frames per step):
    `wait=False` (a PendingStep whose result() is read after the next step is queued), `FusedAdam` (with its NaN flags merged into the
    check_weights warnings) and `FramePrefetcher(device=...)` (uploads on a side stream) -- same losses bit for bit; the parameters after two
    real optimisation steps move the way the torch.optim.Adam route moves them (Adam's first steps are lr * sign(gradient): an element whose
    gradient is at the level of the composite adjoint's float atomics may take the other sign from run to run, so the UPDATES are compared as
    vectors; `test_fused_adam_is_torch_adam_in_one_launch` holds the optimizer itself to a few ulps on identical gradients)."""
    import logging
    import random
    _, data, _, _, _ = _g11_device(golden, dev)
    frames = [dict(data, index=int(data["index"]) + k) for k in range(2)]

    def run(opt_cls, wait, on_device):
        m = _late_model(dev)
        opt = opt_cls([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        tr = s2l.Trainer(m, opt, cfg=_cfg(m), syncnet=syncnet, use_syncloss=True, precision="bf16", hole_noise="device")
        torch.manual_seed(3)
        random.seed(3)
        batch = frames
        if on_device:
            batch = [{k: (v.to(dev) if isinstance(v, torch.Tensor) and v.is_floating_point() and v.numel() > 16 else v) for k, v in f.items()} for f in frames]
        outs, pend = [], None
        for it in (100001, 100002):
            h = tr.train_steps(batch, it=it, wait=wait)
            if wait:
                outs.append(h)
            else:
                assert isinstance(h, s2l.training.PendingStep)
                if pend is not None:
                    outs.append(pend.result())
                pend = h
        if pend is not None:
            outs.append(pend.result())
            assert pend.result() is outs[-1]                      # idempotent
        return outs, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, tr, m

    (ref, pref, _, _) = run(torch.optim.Adam, True, False)
    p0 = {n: p.detach().clone() for n, p in _late_model(dev).named_parameters() if p.requires_grad}      # (every run starts from these seeded weights)
    for opt_cls, wait, on_device in ((torch.optim.Adam, False, False), (s2l.FusedAdam, True, False), (s2l.FusedAdam, False, True)):
        outs, params, tr, m = run(opt_cls, wait, on_device)
        for (la, da), (lb, db) in zip(ref, outs):
            # host tensors like the reference's loss dict (training.py:562-569); `loss` itself stays where the step computed it, as in train_stage1
            assert isinstance(lb, torch.Tensor) and not lb.is_cuda and all(not v.is_cuda for k, v in db.items() if isinstance(v, torch.Tensor) and k != "loss")
            if opt_cls is torch.optim.Adam:
                assert float(la) == float(lb) and all(float(da[k]) == float(db[k]) for k in da if k != "rgb_window")
        assert abs(float(ref[0][1]["loss"]) - float(outs[0][1]["loss"])) == 0.0        # step 1 starts from the same weights whatever the optimizer
        # ... and step 2 sees the weights step 1 left (the packed-weight caches follow an update made through raw pointers)
        assert abs(float(ref[1][1]["loss"]) - float(outs[1][1]["loss"])) <= 2e-3 * abs(float(ref[1][1]["loss"]))
        assert float(outs[1][1]["loss"]) != float(outs[0][1]["loss"])
        up_ref = torch.cat([(pref[n] - p0[n]).reshape(-1) for n in pref])
        up = torch.cat([(params[n] - p0[n]).reshape(-1) for n in pref])
        assert float(up.abs().max()) <= 2.1e-3 and float(up_ref.abs().max()) <= 2.1e-3      # two Adam steps move nothing much further than 2 lr
        cos = float((up * up_ref).sum() / (up.norm() * up_ref.norm()))
        assert cos >= 0.98 and float(((up - up_ref).abs() > 1e-5).float().mean()) <= 0.05, (opt_cls.__name__, cos)
    # a NaN in a stepped parameter is reported through the optimizer's flags, with the state-dict name the reference prints
    m.fc_time.bias.data[3] = float("nan")
    records = []

    class Catch(logging.Handler):
        def emit(self, rec):
            records.append(rec.getMessage())
    log = logging.getLogger("speech2lip_amd.training")
    h = Catch()
    log.addHandler(h)
    try:
        tr.train_steps(frames, it=100003)
    finally:
        log.removeHandler(h)
    assert any("fc_time.bias" in r for r in records), records

    # FramePrefetcher(device=...): the same dictionaries, floating-point tensors already on the device
    from tools.benchlib import write_synthetic_dataset
    root = str(tmp_path / "may_face_crop_lip")
    write_synthetic_dataset(root, 12, FH=40, FW=56, lh=8, lw=12, x0=20, y0=22, train=False, workers=4)
    ds = s2l.SomeonesLipClip(root, "train", s2l.may_config(8, 12, data_path=root))
    order = [3, 0, 7, 5]
    with s2l.FramePrefetcher(ds, order, workers=2, depth=2, per_step=2, collate=False, mode="thread", device=dev) as pf:
        groups = list(pf)
    assert [len(g_) for g_ in groups] == [2, 2]
    for got, i in zip([f for g_ in groups for f in g_], order):
        ref_f = ds.load_one_frame(i)
        assert set(got) == set(ref_f)
        for k, v in ref_f.items():
            if isinstance(v, torch.Tensor) and v.is_floating_point() and v.numel() > 16:
                assert got[k].is_cuda and torch.equal(got[k].cpu(), v), k
            elif isinstance(v, torch.Tensor):
                assert not got[k].is_cuda and torch.equal(got[k], v), k
            else:
                assert got[k] == v, k
