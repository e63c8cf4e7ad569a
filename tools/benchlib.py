"""Measurement functions shared by bench.py (`extra` object, configs 3 and 5) and the per-row CLI tools under tools/.
Every function takes a device, runs on synthetic inputs resident in HBM (SURVEY.md §8d recipe, generators in
speech2lip_amd/weights.py), times with HIP events on the current stream and returns a JSON-able dict."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import speech2lip_amd as s2l  # noqa: E402
from speech2lip_amd import weights as W  # noqa: E402

HBM_PEAK = 8.0e12
FP32_MFMA_PEAK = 157.3e12
BF16_MFMA_PEAK = 2.5e15


def lip_flops_per_frame(hw: int) -> int:
    """SURVEY.md §8d official (factored) figure."""
    return 2 * 459_520 * hw + 2 * (67_328 + 43_008 + 131_072)


def make_model(dev, h, w, unet=False, train=False):
    m = s2l.TalkingFace(dev, s2l.may_config(h, w))
    m = m.train() if train else m.eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
    if unet:
        m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
    return m


def device_warp_coords(dev, n, FH=500, FW=500, seed=2):
    """Device-side twin of weights.synthetic_warp_coords (same recipe, torch generator) for sizes where 2 MB/frame of
    host-generated grids would dominate the set-up time."""
    g = torch.Generator(device=dev).manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(FH, device=dev), torch.arange(FW, device=dev), indexing="ij")
    ident = torch.stack([(2 * xs + 1) / FW - 1, (2 * ys + 1) / FH - 1], -1).float()
    ang = (torch.rand(n, device=dev, generator=g) - 0.5) * (6 * np.pi / 180)          # rotation <= 3 deg
    rot = torch.stack([torch.stack([ang.cos(), -ang.sin()], -1), torch.stack([ang.sin(), ang.cos()], -1)], -2)
    shift = (torch.rand(n, 1, 1, 2, device=dev, generator=g) - 0.5) * 0.04
    jit = torch.randn(n, FH, FW, 2, device=dev, generator=g) * 1e-3
    return (torch.einsum("hwk,fjk->fhwj", ident, rot) + shift + jit).clamp(-1, 1).contiguous(), g


def _median_ms(fn, reps=6, inner=10):
    """Median over `reps` event pairs, each around `inner` back-to-back calls (host launch latency stays out)."""
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) / inner


def bench_render_split(dev, frames=1000, h=96, w=96):
    """The opt-in split-half speed mode of the renderer (csrc/render16.hip) on BASELINE config 2's workload, next to the exact
    fp32 kernel on the same inputs: frames/s of both (HIP events around whole render_clip calls: encoder + frame vectors +
    render), the ratio, and the accuracy of BOTH against the CPU oracle on sampled frames.  Reported in `extra` only: the
    headline value and dtype stay the exact kernel's."""
    from oracle import s2l_oracle as O
    m = make_model(dev, h, w)
    win = torch.from_numpy(W.synthetic_audio(frames, seed=1).astype(np.float32)).to(dev)
    idx = torch.arange(frames, device=dev)
    out = torch.empty(frames, h, w, 3, device=dev)
    res = {"workload": f"{h}x{w} lip crop, {frames} frames per launch (BASELINE config 2's)", "kernel": "s2l::render16_tiles_kernel (s2l_render_lip_split)",
           "arithmetic": "operands as hi + lo IEEE halves, W_lo a_hi + W_hi a_lo + W_hi a_hi on v_mfma_f32_16x16x32_f16, fp32 accumulation"}
    ms = {}
    for prec in ("fp32", "split"):
        m.render_clip(win, idx, h, w, out=out, precision=prec)
        torch.cuda.synchronize()
        ms[prec] = _median_ms(lambda: m.render_clip(win, idx, h, w, out=out, precision=prec), reps=5, inner=4)
    sd = O.to_sd(W.make_state_dict(0, "he"))
    ks = [0, frames // 2, frames - 1]
    with torch.no_grad():
        ref = torch.stack([O.render_clip(sd, win[k:k + 1].cpu(), [k], h, w)[0] for k in ks])
    acc = {}
    for prec in ("fp32", "split"):
        got = m.render_clip(win, idx, h, w, precision=prec)[ks].cpu()
        acc[prec] = {"rmse_vs_cpu": float(f"{O.rmse(got, ref):.3e}"), "psnr_db_vs_cpu": round(O.psnr(got, ref), 1),
                     "max_abs_err": float(f"{float((got - ref).abs().max()):.3e}")}
    flop = lip_flops_per_frame(h * w) * frames
    res.update(ms_per_clip=round(ms["split"], 3), frames_per_s=round(frames / ms["split"] * 1e3, 1),
               exact_kernel_ms_per_clip=round(ms["fp32"], 3), exact_kernel_frames_per_s=round(frames / ms["fp32"] * 1e3, 1),
               speedup_vs_exact=round(ms["fp32"] / ms["split"], 3),
               algorithmic_tflops=round(flop / (ms["split"] * 1e-3) / 1e12, 1),
               frac_of_f16_mfma_peak_at_3_mfma_per_product=round(3 * flop / (ms["split"] * 1e-3) / BF16_MFMA_PEAK, 4),
               parity=acc["split"], exact_kernel_parity=acc["fp32"], frames_checked=ks)
    return res


def bench_config4_rank_block(dev, per_rank=5000, n_total=40_000, world=8):
    """BASELINE config 4's per-rank workload on ONE GPU from its wire format: a 40 000-window float64 `audio.npy` written to disk
    (deepspeech_features.py:65-75), read back and cast (someones_lip_dataset.py:246), the 5 000-frame block of the LAST rank of an
    8-GPU job (frame indices 35 000 ... 39 999) uploaded and rendered at 96x96 in one launch.  `frames_per_s` is the one-GPU rate
    at config 4's step size (the headline's steps are config 2's 1 000 frames): the denominator of a weak-scaling efficiency."""
    import shutil
    import tempfile
    from speech2lip_amd import sharded
    h = w = 96
    tmp = tempfile.mkdtemp(prefix="s2l_c4_")
    try:
        path = os.path.join(tmp, "audio.npy")
        t0 = time.perf_counter()
        np.save(path, W.synthetic_audio(n_total, seed=1))
        t_write = time.perf_counter() - t0
        t0 = time.perf_counter()
        wire = np.load(path)
        first, count, _ = sharded.shard_range(n_total, world - 1, world)
        audio = torch.from_numpy(wire[first:first + count].astype(np.float32)).pin_memory().to(dev, non_blocking=True)
        torch.cuda.synchronize()
        t_read = time.perf_counter() - t0
        nbytes = os.path.getsize(path)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    m = make_model(dev, h, w)
    idx = torch.arange(first, first + count, device=dev)
    out = torch.empty(count, h, w, 3, device=dev)
    m.render_clip(audio, idx, h, w, out=out)
    ms = _median_ms(lambda: m.render_clip(audio, idx, h, w, out=out), reps=5, inner=1)
    u8 = s2l.to8b(out)
    ms_q = _median_ms(lambda: s2l.to8b(out), reps=5, inner=1)
    assert torch.equal(m.render_clip(audio[-8:], idx[-8:], h, w), out[-8:])       # the tail tile: the same bits as its own call
    return {"frames_per_step": count, "first_frame_index": first, "ms_per_step": round(ms, 3), "frames_per_s": round(count / ms * 1e3, 1),
            "tflops": round(2 * 459_520 * h * w * count / ms / 1e9, 1), "to8b_ms": round(ms_q, 3), "audio_npy_mb": round(nbytes / 1e6, 1),
            "audio_npy_write_s": round(t_write, 2), "audio_npy_read_cast_upload_s": round(t_read, 3),
            "gather_mb_per_rank": {"f32": round(out.numel() * 4 / 1e6, 1), "u8": round(u8.numel() / 1e6, 1)}}


def bench_small_clips(dev):
    """The reference's own operating point: ONE frame per call (inference.py:129 DataLoader batch 1, :140-159), and BASELINE
    config 1's 16-frame clip.  Per size: `render_clip` at F = 1 and F = 16 (per-call latency through the host wrapper as a caller
    sees it, and stream time per call from HIP events around back-to-back calls), the long-clip per-frame cost for comparison,
    and the drop-in per-frame sequence audio_merge_forward(H*W tiled windows) + rgb_forward(H*W rows) (inference.py:144-159)."""
    res = {}
    for h in (64, 96):
        w, hw = h, h * h
        m = make_model(dev, h, w)
        long_f = 480
        audio = torch.from_numpy(W.synthetic_audio(long_f, 1).astype(np.float32)).to(dev)
        idx = torch.arange(long_f, device=dev)
        out_long = torch.empty(long_f, h, w, 3, device=dev)
        m.render_clip(audio, idx, h, w, out=out_long)
        ms_long = _median_ms(lambda: m.render_clip(audio, idx, h, w, out=out_long), reps=5, inner=2)
        entry = {"long_clip_us_per_frame": round(ms_long / long_f * 1e3, 2)}
        for F in (1, 16):
            a, i, o = audio[:F].contiguous(), idx[:F].contiguous(), torch.empty(F, h, w, 3, device=dev)
            call = lambda: m.render_clip(a, i, h, w, out=o)
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            ms_stream = _median_ms(call, reps=7, inner=20)
            lat = []
            for _ in range(30):               # latency of ONE call seen by a synchronous caller
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                call()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            assert torch.equal(o, out_long[:F])          # any tile shape: the same bits as the long clip's frames
            entry[f"render_clip_F{F}"] = {"stream_us_per_call": round(ms_stream * 1e3, 1), "frames_per_s": round(F / ms_stream * 1e3, 1),
                                          "sync_latency_us_per_call": round(float(np.median(lat)) * 1e6, 1),
                                          "per_frame_cost_over_long_clip": round(ms_stream / F / (ms_long / long_f), 2)}
        # ONE frame per call through a captured HIP graph (speech2lip_amd.FrameGraph): input copies + one replay, as a synchronous
        # per-frame caller (the reference's batch_size-1 loop) would see it
        fg = s2l.FrameGraph(m, 1, h, w)
        a1, i1 = audio[:1].contiguous(), idx[:1].contiguous()
        gcall = lambda: fg(a1, i1)
        for _ in range(5):
            gcall()
        torch.cuda.synchronize()
        ms_g = _median_ms(gcall, reps=7, inner=20)
        lat = []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gcall()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        assert torch.equal(fg(a1, i1), out_long[:1])
        entry["frame_graph_F1"] = {"stream_us_per_call": round(ms_g * 1e3, 1), "frames_per_s": round(1e3 / ms_g, 1),
                                   "sync_latency_us_per_call": round(float(np.median(lat)) * 1e6, 1)}
        del fg
        # the drop-in per-frame sequence, as inference.py drives the module
        from speech2lip_amd import get_coords
        coords = get_coords(w, h, dev)
        win = audio[:1]

        def dropin():
            feat = m.audio_merge_forward(win.tile(hw, 1, 1))                         # inference.py:144, 151
            rows = torch.cat([coords[:, None, :], feat[:, None, :]], -1).view(-1, 66)   # :152
            return m.rgb_forward(rows, time_pts=torch.tensor([7]))[:, :3]           # :158-159
        with torch.no_grad():
            for _ in range(3):
                r = dropin()
            torch.cuda.synchronize()
            ms_drop = _median_ms(dropin, reps=5, inner=10)
        entry["dropin_per_frame_sequence"] = {"us_per_frame": round(ms_drop * 1e3, 1), "frames_per_s": round(1e3 / ms_drop, 1),
                                              "max_abs_diff_vs_render_clip": float((r.reshape(h, w, 3) - m.render_clip(win, [7], h, w)[0]).abs().max())}
        res[f"{h}x{w}"] = entry
    return res


def bench_composite(dev, frames=256, check=True):
    """A7 alone at BASELINE config 3 geometry: 128x128 lip in a 500x500 face.  HBM-bound: 8,000,000 + 12 h w B/frame."""
    h = w = 128
    FH = FW = 500
    x0, y0 = 186, 300
    m = make_model(dev, h, w)
    coord, g = device_warp_coords(dev, frames)
    lip = torch.rand(frames, h, w, 3, device=dev, generator=g)
    face = torch.rand(1, FH, FW, 3, device=dev, generator=g)
    gt = torch.rand(frames, FH, FW, 3, device=dev, generator=g)
    mask = torch.zeros(1, FH, FW, 3, device=dev)
    mask[:, y0:y0 + h, x0:x0 + w] = 1
    out = torch.empty(frames, FH, FW, 3, device=dev)
    run = lambda: m.composite_clip(lip, face, gt, mask, x0, y0, coord, out=out)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ms = _median_ms(run)
    bpf = 8_000_000 + 12 * h * w
    gbs = bpf * frames / (ms * 1e-3) / 1e9
    res = {"kernel": "s2l::span_kernel + s2l::lip_merge_kernel (clip path; per-clip s2l::composite_tables_kernel)", "frames": frames, "ms": round(ms, 4), "frames_per_s": round(frames / ms * 1e3, 1),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": round(gbs * 1e9 / HBM_PEAK, 4), "algorithmic_bytes_per_frame": bpf}}
    if check:
        from oracle import s2l_oracle as O
        rn, _ = O.composite(lip[:1].cpu(), face.cpu(), gt[:1].cpu(), mask.cpu(), x0, y0, coord[:1].cpu())
        d = (out[:1].cpu() - rn).abs()
        res["parity_frame0"] = {"max_abs_err": float(d.max()), "pixels_off_by_more_than_1e-5": int((d > 1e-5).sum())}
    return res


def bench_config3(dev, n=5000, batch=500, unet=False, check=True, unet_precision="fp32", lip_precision="fp32"):
    """BASELINE config 3 end to end: 128x128 lip render for n frames + composite into 500x500 (+ optionally the U-Net;
    unet_precision "fp32" = the exact parity kernels, "split" = split-bf16 operands, SimpleUnetLight.forward_nhwc)."""
    h = w = 128
    FH = FW = 500
    x0, y0 = 186, 300
    m = make_model(dev, h, w, unet=True)
    audio = torch.from_numpy(W.synthetic_audio(n, 1).astype(np.float32)).to(dev)
    coord, g = device_warp_coords(dev, batch)      # one batch of pose grids / observed frames, reused (content-independent timing)
    face = torch.rand(1, FH, FW, 3, device=dev, generator=g)
    gt = torch.rand(batch, FH, FW, 3, device=dev, generator=g)
    mask = torch.zeros(1, FH, FW, 3, device=dev)
    mask[:, y0:y0 + h, x0:x0 + w] = 1
    lip = torch.empty(batch, h, w, 3, device=dev)
    out = torch.empty(batch, FH, FW, 3, device=dev)
    recon = torch.empty(batch, FH, FW, 3, device=dev) if unet else None

    def run():
        for s in range(0, n, batch):
            k = min(batch, n - s)
            m.render_clip(audio[s:s + k], torch.arange(s, s + k, device=dev), h, w, out=lip[:k], precision=lip_precision)
            m.composite_clip(lip[:k], face, gt[:k], mask, x0, y0, coord[:k], out=out[:k])
            if unet:
                m.post_fusion_unet.forward_nhwc(out[:k], out=recon[:k], precision=unet_precision)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = lip_flops_per_frame(h * w)
    res = {"config": f"config 3: {n} frames, 128x128 lip + composite into 500x500"
           + (f" + post-fusion U-Net ({'exact fp32 MFMA' if unet_precision == 'fp32' else 'split operands hi+lo (IEEE halves), fp32 accumulation' if unet_precision == 'split' else unet_precision})" if unet else "")
           + (", lip renderer in its split-half speed mode" if lip_precision == "split" else "")
           + f", batches of {batch}", "seconds": round(dt, 3), "frames_per_s": round(n / dt, 1),
           "lip_gflop_per_frame": round(fl / 1e9, 3), "lip_tflops": round(fl * n / dt / 1e12, 1)}
    if check:
        from oracle import s2l_oracle as O
        s = (n - 1) // batch * batch
        sd = O.to_sd(W.make_state_dict(0, "he"))
        with torch.no_grad():
            ref_lip = O.render_clip(sd, audio[s:s + 1].cpu(), [s], h, w)
            ref_new, _ = O.composite(ref_lip, face.cpu(), gt[:1].cpu(), mask.cpu(), x0, y0, coord[:1].cpu())
        par = {"lip_rmse": float(f"{O.rmse(lip[0].cpu(), ref_lip[0]):.3e}"),
               "composite_rmse": float(f"{O.rmse(out[0].cpu(), ref_new[0]):.3e}"),
               "composite_psnr_db": round(O.psnr(out[0].cpu(), ref_new[0]), 1)}
        if unet:
            with torch.no_grad():
                ref_recon = O.unet_forward(O.to_sd(W.make_unet_state_dict(0)), ref_new)
            par["unet_rmse"] = float(f"{O.rmse(recon[0].cpu(), ref_recon[0]):.3e}")
            par["unet_psnr_db"] = round(O.psnr(recon[0].cpu(), ref_recon[0]), 1)
        res["parity"] = par
    return res


def bench_unet(dev, F=16, H=500, Wd=500):
    """Post-fusion U-Net at the reference's face frame: eval forward (what inference runs), and saved forward + input gradient
    (the frozen net inside loss.backward()); 2 * 78.8 GMAC per frame and pass."""
    import speech2lip_amd as s2l
    u = s2l.SimpleUnetLight().to(dev).eval()
    u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
    x = torch.rand(F, H, Wd, 3, device=dev)
    d = torch.randn(F, H, Wd, 3, device=dev)
    out = torch.empty_like(x)
    macs = 64 * 3 * H * Wd
    for (name, cin, cout), s in zip(W.UNET_CONVS, (1, 1, 2, 2, 4, 4, 2, 2, 1, 1)):
        macs += cin * cout * 9 * (H // s) * (Wd // s)

    def fb():
        o, ctx = u.forward_saved_nhwc(x)
        u.backward_input(ctx, d)
    res = {"config": f"SimpleUnetLight {H}x{Wd}, fp32 MFMA, {F} frames per call", "gflop_per_frame": round(2 * macs / 1e9, 2)}
    for key, fn, passes in (("forward", lambda: u.forward_nhwc(x, out=out), 1), ("forward_plus_input_gradient", fb, 2)):
        for _ in range(2):
            fn()
        ms = _median_ms(fn, reps=5, inner=1)
        tf = passes * 2 * macs * F / (ms * 1e-3) / 1e12
        res[key] = {"ms_per_frame": round(ms / F, 4), "frames_per_s": round(F / ms * 1e3, 1), "tflops": round(tf, 1),
                    "frac_of_mfma_peak": round(tf / 157.3, 4)}
    # the inference speed mode: split-bf16 operands (three bf16 MFMAs per product), against the exact fp32 output above
    ref = u.forward_nhwc(x).clone()
    for _ in range(2):
        u.forward_nhwc(x, out=out, precision="split")
    ms = _median_ms(lambda: u.forward_nhwc(x, out=out, precision="split"), reps=5, inner=1)
    mse = float(((out.double() - ref.double()) ** 2).mean())
    res["forward_split_bf16"] = {"ms_per_frame": round(ms / F, 4), "frames_per_s": round(F / ms * 1e3, 1),
                                 "speedup_vs_fp32": round(res["forward"]["ms_per_frame"] / (ms / F), 2),
                                 "psnr_db_vs_fp32": round(10 * np.log10(1.0 / max(mse, 1e-30)), 1), "rmse_vs_fp32": float(f"{np.sqrt(mse):.3e}"),
                                 "operands": "hi + lo IEEE-half parts of every fp32 operand (hi toward zero, lo to nearest); a_lo b_hi + a_hi b_lo + a_hi b_hi on v_mfma_f32_32x32x16_f16, fp32 accumulation"}
    return res


def warm_until_allocator_settles(fn, max_steps=8, min_steps=4, tol=1.25):
    """Run `fn` until a step is in steady state: at least `min_steps` steps, no new device allocation by torch's caching
    allocator during the step (a step that still calls hipMalloc for tens of GB is 5-10x slower), and a wall time within `tol`
    of the previous step's (the first few steps of a process also pay one-off costs -- lazily loaded code objects of the
    optimiser's kernels, clock ramp -- of ~70 ms in total, which a 5-step timed window would otherwise report as 2x).
    Returns fn's last result."""
    r, last = None, None
    for i in range(max_steps):
        before = torch.cuda.memory_stats().get("num_device_alloc", 0)
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        settled = torch.cuda.memory_stats().get("num_device_alloc", 0) == before and last is not None and dt <= tol * last
        last = dt
        if settled and i + 1 >= min_steps:
            break
    return r


def median_window_seconds(fn, steps, windows=3):
    """Seconds per step: the median of `windows` timed windows of `steps` back-to-back steps each (no synchronisation inside a window,
    as a training loop runs).  One window that meets a device allocation or a clock ramp does not move the figure as it moved the
    single three-step mean (48.8 instead of 44.4 ms once).  Returns (seconds, fn's last result)."""
    means, r = [], None
    for _ in range(windows):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            r = fn()
        torch.cuda.synchronize()
        means.append((time.perf_counter() - t) / steps)
    means.sort()
    return means[len(means) // 2], r


def train_flops(B, hw=96 * 96):
    """as-written model, fwd + dgrad + wgrad, 4 ensemble taps."""
    return 3 * 4 * 2 * 644_864 * hw * B


def sync_batch(dev, S, T=5, FH=500, FW=500, h=96, w=96, x0=202, y0=316, seed=11):
    """Synthetic inputs of the sync-loss window for S samples (SURVEY.md §8d geometry: 96x96 lip at (202,316) in a 500x500
    face): the dict StageOneStep.loss_and_grads takes as `sync`."""
    g = torch.Generator(device=dev).manual_seed(seed)
    coord, _ = device_warp_coords(dev, S * T, FH, FW, seed=seed)
    mask = torch.zeros(1, FH, FW, 3, device=dev)
    mask[:, y0:y0 + h, x0:x0 + w] = 1
    mel, _, neg = (torch.from_numpy(x).to(dev) for x in W.synthetic_sync_batch(S, seed=seed))
    return dict(audio_window=torch.from_numpy(W.synthetic_audio(S * T, seed).astype(np.float32)).reshape(S, T, 16, 29).to(dev),
                u01=torch.rand(S, T, generator=torch.Generator().manual_seed(seed)).tolist(), total_frame=100000,
                rgb_face_canonical=torch.rand(1, FH, FW, 3, device=dev, generator=g), rgb_face_gt=torch.rand(S, FH, FW, 3, device=dev, generator=g),
                mask_lip_canonical=mask, lip_lefttop_x=x0, lip_lefttop_y=y0, coord_window=coord.reshape(S, T, FH, FW, 2),
                canonical_face_bbox=[110, 90, 390, 420, 1.0], mel=mel, rgb_window_neg=neg)


def bench_train_sync(dev, B=64, S=8, precision="bf16", steps=3, unet_train_mode=False, half_width_tensors=True, frames_per_group=None):
    """BASELINE config 5 as named -- MLP forward + backward WITH the lipsync_expert loss: B main frames (MSE) of which the
    first S carry a 5-frame sync window (5 S more renders -> composite -> frozen U-Net @500x500 -> crop/resize -> SyncNet x2 ->
    BCE, and all of it back to the MLP), bf16 MLP kernels, Adam.  FLOPs counted: the MLP's as-written 3 x 4 x 2 x 644,864 per
    pixel-frame over B + 5 S frames, plus 2 x 157.6 GFLOP per 500x500 window frame for the U-Net forward + input gradient, scaled
    by the share of the frame the chain actually evaluates (the face box dilated by the network's dependency radius) --
    reported separately."""
    H = Wd = 96
    m = make_model(dev, H, Wd, unet=True, train=True)
    for p in m.post_fusion_unet.parameters():
        p.requires_grad = False
    # eval: running statistics (train.py:195), crop window + bf16 operands; train: the mode the reference's loop really leaves the
    # frozen net in (Trainer.train_step's model.train(), training.py:150; golden G16): batch statistics per one-frame call,
    # whole 500x500 frames (bf16 operands too when the step's precision is bf16)
    if not unet_train_mode:
        m.post_fusion_unet.eval()
    m.post_fusion_unet.half_width_tensors = half_width_tensors      # (train mode + bf16 only: bf16 tensors between the U-Net's kernels)
    if frames_per_group:
        m.post_fusion_unet.train_frames_per_group = frames_per_group
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    opt = torch.optim.Adam([p for n_, p in m.named_parameters() if not n_.startswith(("coord_linears", "post_fusion_unet"))], lr=1e-4)
    audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
    target = torch.rand(B, H * Wd, 3, device=dev)
    step = s2l.StageOneStep(m, H, Wd, syncnet=net, precision=precision)
    _loss_conv_ab(step)
    sync = sync_batch(dev, S) if S else None
    u01 = [0.5] * B

    def one():
        loss, g, aux = step.loss_and_grads(audio, list(range(B)), target, u01, sync=sync)
        s2l.training.apply_grads(m, g)
        opt.step()
        return loss, aux
    l0 = float(one()[0])
    warm_until_allocator_settles(one)
    dt, (l, aux) = median_window_seconds(one, steps)
    mlp = train_flops(B + 5 * S, H * Wd)
    wx0, wy0, wx1, wy1 = step.chain.unet_window(sync["canonical_face_bbox"], 500, 500) if S else (0, 0, 500, 500)
    unet = 2 * 157.6e9 * 5 * S * ((wx1 - wx0) * (wy1 - wy0) / 250000.0)     # the U-Net runs on the face box + its dependency radius
    if unet_train_mode:
        wx0, wy0, wx1, wy1 = 0, 0, 500, 500
        unet = 2 * 157.6e9 * 5 * S
    return {"config": f"stage-1 step, {B} main frames 96x96 + sync loss on {S} samples ({5 * S} window frames through composite + U-Net "
                      f"@500x500 + SyncNet), {precision} MLP, Adam" + (", frozen U-Net in TRAIN-mode BatchNorm (the reference's loop, G16)" +
                                                                        (", bf16 tensors between its kernels (csrc/unet_half.inc)" if half_width_tensors and precision == "bf16" else ", fp32 tensors between its kernels")
                                                                        if unet_train_mode else ", frozen U-Net in eval-mode BatchNorm"), "ms_per_step": round(dt * 1e3, 2),
            "unet_window": [wx0, wy0, wx1, wy1], "mlp_frames_per_step": B + 5 * S, "mlp_tflop_per_step": round(mlp / 1e12, 3), "unet_tflop_per_step": round(unet / 1e12, 3),
            "tflops": round((mlp + unet) / dt / 1e12, 1), "loss_first": l0, "loss_last": float(l),
            "loss_sync_last": float(aux.get("loss_sync", 0.0)), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def _loss_conv_ab(step):
    """S2L_BENCH_LOSS_CONV=fp32: the bf16 step with the loss nets' EXACT convolutions (the A/B of the split-operand form)."""
    form = os.environ.get("S2L_BENCH_LOSS_CONV")
    if form:
        step.loss_conv_precision = form
        if step.chain is not None:
            step.chain.sync.precision = form


def bench_stage1_full(dev, B=8, precision="bf16", steps=3, unet_train_mode=False, early=False):
    """One FULL stage-1 iteration of the reference per sample (training.py:347-574 after it > 100000, May flags), B samples per
    step: MSE + LPIPS on the 96x96 lip, MSE + LPIPS on the fused face (composite with black holes -> frozen U-Net @500x500), the
    lipsync_expert loss over a 5-frame window (5 more renders -> composite -> U-Net crop -> SyncNet x2), all of it back to
    the MLP, Adam.  (The reference runs batch_size 1: `ms_per_sample` is the time of one of its iterations.)"""
    H = Wd = 96
    m = make_model(dev, H, Wd, unet=True, train=True)
    # early = the iterations before `it > 100000` (train.py:188-197 not reached yet): the post-fusion net TRAINS with the MLP (train-mode
    # BatchNorm, parameter gradients, golden G14), no sync loss.  Otherwise the net is frozen: eval-mode BatchNorm as train.py:195 words
    # it, or (unet_train_mode) train-mode BatchNorm as the loop really runs it (Trainer.train_step's model.train(), golden G16)
    if not early:
        for p in m.post_fusion_unet.parameters():
            p.requires_grad = False
        if not unet_train_mode:
            m.post_fusion_unet.eval()
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    lp = s2l.LPIPS(pretrained=False, net="alex", version="0.1").to(dev)      # seeded weights are loaded below
    lp.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_lpips_state_dict(0).items()})
    opt = torch.optim.Adam([p for n_, p in m.named_parameters() if not n_.startswith("coord_linears") and
                            (early or not n_.startswith("post_fusion_unet"))], lr=1e-4)
    audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
    target = torch.rand(B, H * Wd, 3, device=dev)
    step = s2l.StageOneStep(m, H, Wd, syncnet=None if early else net, precision=precision, face_loss=True, perceptual=lp)
    _loss_conv_ab(step)
    sync = sync_batch(dev, B)
    coord, g = device_warp_coords(dev, B, seed=5)
    face = dict(rgb_face_canonical=sync["rgb_face_canonical"], rgb_face_gt=sync["rgb_face_gt"], mask_lip_canonical=sync["mask_lip_canonical"],
                lip_lefttop_x=sync["lip_lefttop_x"], lip_lefttop_y=sync["lip_lefttop_y"], coord=coord,
                hole_noise=(torch.randn(B, 500, 500, device=dev, generator=g), torch.randn(B, 500, 500, device=dev, generator=g)))
    u01 = [0.5] * B

    def one():
        loss, gr, aux = step.loss_and_grads(audio, list(range(B)), target, u01, sync=None if early else sync, face=face)
        s2l.training.apply_grads(m, gr)
        opt.step()
        return loss, aux
    l0 = float(one()[0])
    warm_until_allocator_settles(one)
    dt, (l, aux) = median_window_seconds(one, steps)
    mode = ("U-Net TRAINING with the MLP (it <= 100000: train-mode BatchNorm, parameter gradients), no sync loss" if early else
            "frozen U-Net in TRAIN-mode BatchNorm (the reference's loop, G16), sync loss over a 5-frame window" if unet_train_mode else
            "frozen U-Net in eval-mode BatchNorm, sync loss over a 5-frame window")
    return {"config": f"full stage-1 iteration x {B} samples: MSE + LPIPS(alex) on the 96x96 lip and on the fused 500x500 face (composite "
                      f"with black holes + post-fusion U-Net), {mode}, {precision} MLP, Adam",
            "ms_per_step": round(dt * 1e3, 2), "ms_per_sample": round(dt * 1e3 / B, 2), "loss_first": l0, "loss_last": float(l),
            "loss_perceptual_last": float(aux["loss_perceptual"]), "loss_face_last": float(aux["loss_face"]),
            "loss_sync_last": float(aux.get("loss_sync", 0.0)), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def bench_sync_loss(dev, B=16, iters=10):
    """T3 alone: SyncNet on B generated + B negative windows, cosine/BCE, gradient w.r.t. the generated window -- with the exact fp32
    convolutions (default) and with the split-operand form the bf16-precision steps use (csrc/conv_gemm.h)."""
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    mel, pos, neg = (torch.from_numpy(x).to(dev) for x in W.synthetic_sync_batch(B, seed=1))
    out = {"config": f"sync loss alone: SyncNet_color on {B} generated + {B} negative windows, cosine/BCE, gradient w.r.t. the window", "batch": B}
    for prec in ("fp32", "split"):
        sl = s2l.SyncLoss(net, precision=prec)
        fn = lambda: sl.get_sync_contrastive_loss(mel, pos, neg, want_grad=True)
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record(); torch.cuda.synchronize()
        out[f"ms_loss_and_grad_{prec}"] = round(a.elapsed_time(b) / iters, 3)
    return out


def bench_train(dev, B=64, precision="bf16", steps=5):
    """BASELINE config 5: one step = 4-tap ensemble forward + MSE + full backward + Adam for B frames at 96x96."""
    H = Wd = 96
    m = make_model(dev, H, Wd, train=True)
    opt = torch.optim.Adam([p for n_, p in m.named_parameters() if not n_.startswith("coord_linears")], lr=1e-4)
    audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
    target = torch.rand(B, H * Wd, 3, device=dev)
    step = s2l.LipTrainStep(m, H, Wd, precision=precision)
    u01 = [0.5] * B

    def one():
        loss, g, _ = step.loss_and_grads(audio, list(range(B)), target, u01)
        s2l.training.apply_grads(m, g)
        opt.step()
        return loss
    l0 = float(one())
    warm_until_allocator_settles(one)
    t0 = time.perf_counter()
    for _ in range(steps):
        l = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = train_flops(B, H * Wd)
    peak = BF16_MFMA_PEAK if precision == "bf16" else FP32_MFMA_PEAK
    return {"config": f"training step, {B} frames 96x96, {precision}" + (" parity mode" if precision == "fp32" else
            " MFMA, fp32 accumulate + master weights") + ", Adam", "ms_per_step": round(dt * 1e3, 2),
            "frames_per_s": round(B / dt, 1), "as_written_tflop_per_step": round(fl / 1e12, 3),
            "tflops": round(fl / dt / 1e12, 1), "frac_of_mfma_peak": round(fl / dt / peak, 4), "loss_first": l0,
            "loss_last": float(l), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "_step": step,
            "_one": one}


# ---------------------------------------------------------------------------------------------------------------------
# The callers the reference actually has, fed from a dataset folder on disk (VERDICT r04 "missing" 2 and 4)
def write_synthetic_dataset(root, n, FH=500, FW=500, lh=96, lw=96, x0=202, y0=316, seed=3, train=True, workers=16):
    """A clip of `n` frames in the reference's on-disk layout (someones_lip_dataset.py:43-120; field list in
    tools/make_dataset_fixture.py) at the REAL sizes: 500x500 face JPEGs, 96x96 lip JPEGs, float32 [500,500,2] pose grids
    (identity + rigid perturbation + jitter), DeepSpeech windows, and -- `train` -- the sync-loss side inputs
    (mel.npy, face_bbox_dict.npy).  Smooth random images (JPEG-friendly); written by a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    rng = np.random.default_rng(seed)
    for sub in ("audio", "audio_test", "coords", "ori_images_face", "images", "landmarks"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    np.save(os.path.join(root, "audio", "audio.npy"), W.synthetic_audio(n, seed=1).astype(np.float64))
    np.save(os.path.join(root, "audio_test", "audio.npy"), W.synthetic_audio(8, seed=2).astype(np.float64))
    ys, xs = np.meshgrid(np.arange(FH), np.arange(FW), indexing="ij")
    ident = np.stack([(2 * xs + 1) / FW - 1, (2 * ys + 1) / FH - 1], -1).astype(np.float32)
    base = rng.random((FH // 10 + 1, FW // 10 + 1, 3)).astype(np.float32)
    face0 = np.asarray(Image.fromarray((base * 255).astype(np.uint8)).resize((FW, FH), Image.BILINEAR))

    def one(k):
        r = np.random.default_rng(seed * 100003 + k)
        ang = (r.random() - 0.5) * (6 * np.pi / 180)
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]], np.float32)
        grid = ident @ rot.T + ((r.random(2) - 0.5) * 0.04).astype(np.float32) + (r.standard_normal((FH, FW, 2)) * 1e-3).astype(np.float32)
        np.save(os.path.join(root, "coords", "%05d.npy" % (k + 1)), np.clip(grid, -1, 1).astype(np.float32))
        img = np.clip(face0.astype(np.int16) + r.integers(-20, 20, (1, 1, 3)), 1, 255).astype(np.uint8)      # > 0: inside the face mask
        Image.fromarray(img).save(os.path.join(root, "ori_images_face", "%05d.jpg" % (k + 1)), quality=92)
        Image.fromarray(img[y0:y0 + lh, x0:x0 + lw]).save(os.path.join(root, "images", "%05d.jpg" % (k + 1)), quality=92)
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(one, range(n)))
    mask = np.zeros((FH, FW, 3), np.uint8)
    mask[y0:y0 + lh, x0:x0 + lw] = 255
    Image.fromarray(mask).save(os.path.join(root, "canonical_lip_mask.jpg"), quality=100)
    # mouth landmarks whose bounding box centre gives the lip origin (x0, y0) through compute_mouth_bbox's 1.02 rule
    # (someones_lip_dataset.py:173-193: boundingRect's centre, y scaled by 1.02, minus half the crop, truncated)
    cx = int(x0 + lw / 2.0)
    cy = next(c for c in range(1, FH) if int(c * 1.02 - lh / 2.0) == y0)
    lms = np.full((68, 2), 1.0, np.float32)
    lms[48:, 0] = np.linspace(cx - 20, cx + 19, 20)
    lms[48:, 1] = np.linspace(cy - 10, cy + 9, 20)
    np.savetxt(os.path.join(root, "landmarks", "00001.lms"), lms)
    if train:
        np.save(os.path.join(root, "audio", "mel.npy"), rng.standard_normal((80, 4 * n + 64)).astype(np.float32))
        np.save(os.path.join(root, "face_bbox_dict.npy"), {"%05d.jpg" % (k + 1): np.array([110, 90, 390, 420, 0.99], np.float32) for k in range(n)})
    return root


def _dataset_tmp(name):
    import tempfile
    return os.path.join(tempfile.mkdtemp(prefix="s2l_bench_"), name)


def bench_infer_clip(dev, n_total=640, batch=50):
    """The inference driver END TO END with I/O, the loop `inference.py:140-178` replaced: dataset folder on disk (JPEG frames +
    2-MB pose grids per frame) -> load -> lip render 96x96 -> composite into 500x500 -> post-fusion U-Net -> 8-bit -> JPEG files.
    The folder's name contains `may`, so the validation split is the reference's: the LAST 598 frames (someones_lip_dataset.py
    :143-145).  Three forms of tools/infer_clip.py's loop, frames/s each: serial (load, render, write one after the other),
    pipelined (ClipStreamer + FrameWriter: JPEG decode in child processes -- PIL's decoder holds the interpreter lock: `_decode_threads`
    is the same pipeline with the decoders in threads --, encode on host threads, byte-wide H2D on a side stream, beside the GPU
    work), and pipelined with the split speed modes; plus the GPU-only rate of the same batches from resident inputs."""
    import shutil
    root = _dataset_tmp("may_face_crop_lip")
    t0 = time.perf_counter()
    write_synthetic_dataset(root, n_total, train=False)
    t_write = time.perf_counter() - t0
    cfg = s2l.may_config(96, 96, data_path=root)
    ds = s2l.SomeonesLipClip(root, "val", cfg)
    n = len(ds)
    m = make_model(dev, ds.lip_h, ds.lip_w, unet=True)
    m.data_path = root
    out_dir = os.path.join(os.path.dirname(root), "out")
    res = {"frames": n, "decode_workers": min(8, os.cpu_count() or 1), "encode_threads": min(16, os.cpu_count() or 1), "dataset_write_s": round(t_write, 2)}

    def serial():
        for first in range(0, n, batch):
            clip = ds.load(dev, first, min(batch, n - first))
            lip, recon, merged = s2l.render_clip_frames(m, clip)
            s2l.write_frames(recon, clip.names, out_dir)

    def piped(precision, mode="process"):
        with s2l.ClipStreamer(ds, dev, batch, mode=mode) as st, s2l.FrameWriter(out_dir) as wr:
            for clip in st:
                lip, recon, merged = s2l.render_clip_frames(m, clip, precision=precision)
                wr.submit(s2l.to8b(recon), clip.names)
    # warm-up on a short prefix (kernel code objects, pinned allocations), then one timed pass each
    t0 = time.perf_counter()
    with s2l.ClipStreamer(ds, dev, 8, 0, 16, mode="process") as warm:      # starts the decode worker processes (once per process; ~1-3 s on a cold box)
        for _ in warm:
            pass
    res["decode_workers_start_s"] = round(time.perf_counter() - t0, 2)
    clip0 = ds.load(dev, 0, min(batch, n))
    s2l.render_clip_frames(m, clip0)
    s2l.render_clip_frames(m, clip0, precision="split")
    torch.cuda.synchronize()
    for key, fn, reps in (("serial", serial, 1), ("pipelined_decode_threads", lambda: piped("fp32", "thread"), 3), ("pipelined", lambda: piped("fp32"), 3),
                          ("pipelined_split_modes", lambda: piped("split"), 3)):
        rates = []
        for _ in range(reps):      # (a 598-frame pass lasts under a second: the median of three)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            rates.append(n / (time.perf_counter() - t0))
        res[key + "_frames_per_s"] = round(sorted(rates)[len(rates) // 2], 1)
    for key, prec in (("gpu_only", "fp32"), ("gpu_only_split_modes", "split")):
        ms = _median_ms(lambda: s2l.to8b(s2l.render_clip_frames(m, clip0, precision=prec)[1]), reps=3, inner=1)
        res[key + "_frames_per_s"] = round(clip0.audio.shape[0] / ms * 1e3, 1)
    # the files of the pipelined pass hold the frames of the serial pass (same kernels, same quantisation): compare one decoded file
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(out_dir, "%05d.jpg" % n)))
    ref = s2l.to8b(s2l.render_clip_frames(m, ds.load(dev, n - 1, 1), precision="split")[1])[0].cpu().numpy()
    res["last_file_mean_abs_diff_vs_frame_u8"] = round(float(np.abs(a.astype(np.int16) - ref.astype(np.int16)).mean()), 3)     # JPEG loss only
    shutil.rmtree(os.path.dirname(root), ignore_errors=True)
    return res


def bench_dropin_trainer(dev, n_frames=48, iters=200):
    """The training caller the reference has: train.py:173-199's loop body -- batch = next(DataLoader over SomeonesLipDataset),
    `Trainer.train_step(batch, it=it)` -- through the drop-in `speech2lip_amd.Trainer`, one frame per iteration, bf16 precision,
    the frozen post-fusion U-Net in train-mode BatchNorm (what `model.train()` in train_step leaves it in, training.py:150),
    fed by `SomeonesLipClip.load_one_frame` on a 96x96-lip / 500x500-face folder on disk.  Reported per phase (`it` <= 100000:
    the U-Net still trains, no sync loss; `it` > 100000: frozen U-Net + the 5-frame sync window): ms per iteration of
      * `train_step` with the reference's CPU black-hole noise stream, data loaded in the loop (the loop as written),
      * the same with a prefetching loader (FramePrefetcher = DataLoader workers) and `hole_noise="device"`,
      * the same with `Trainer(fused_step=True)` (train_step's one frame goes through the fused engine),
      * `train_steps` with K = 8 frames per optimisation step (ms per FRAME),
    each with the GPU-busy share (sum of HIP-event spans around the step's GPU work / wall time), and the data loader alone."""
    import shutil
    root = _dataset_tmp("may_face_crop_lip")
    write_synthetic_dataset(root, n_frames, train=True)
    cfg = s2l.may_config(96, 96, data_path=root, train_flags=True)
    cfg["model"]["use_canonical_depth"] = False
    cfg["training"].update(use_sync_contrastive_loss=True, use_perceptual_loss=True, use_canonical_depth_loss_photo_v2=False,
                           use_syncloss=True)
    ds = s2l.SomeonesLipClip(root, "train", cfg=cfg)
    n = len(ds)
    lp = s2l.LPIPS(pretrained=False, net="alex", version="0.1").to(dev)      # seeded weights (the real ones are not in the reference repo)
    lp.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_lpips_state_dict(0).items()})
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    for p in net.parameters():
        p.requires_grad = False
    res = {"frames_in_split": n, "iterations": iters, "loader_threads": min(16, os.cpu_count() or 1)}
    t0 = time.perf_counter()
    for i in range(8):
        ds.load_one_frame(i % n)
    res["load_one_frame_ms"] = round((time.perf_counter() - t0) / 8 * 1e3, 2)

    def build(late, hole_noise, fused_step=False, adam=torch.optim.Adam):
        m = make_model(dev, 96, 96, unet=True, train=True)
        m.data_path = root
        if late:      # train.py:188-197
            for p in m.post_fusion_unet.parameters():
                p.requires_grad = False
            m.post_fusion_unet.eval()
        opt = adam([p for nm, p in m.named_parameters() if p.requires_grad and not nm.startswith("coord_linears")], lr=1e-4)
        return s2l.Trainer(m, opt, dev, None, cfg=cfg, syncnet=net, perceptual_loss_fn=lp, precision="bf16", hole_noise=hole_noise,
                           fused_step=fused_step)

    def run(tr, it0, batches, per_step, pipelined=False):
        torch.cuda.synchronize()
        spans, k, pending = [], 0, None
        t0 = time.perf_counter()
        for batch in batches:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if per_step == 1:
                tr.train_step(batch, it=it0 + k)
            elif pipelined:      # step k's losses / NaN report are read after step k + 1 has been queued
                h = tr.train_steps(batch, it=it0 + k, wait=False)
                if pending is not None:
                    pending.result()
                pending = h
            else:
                tr.train_steps(batch, it=it0 + k)
            e1.record()
            spans.append((e0, e1))
            k += 1
        if pending is not None:
            pending.result()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        busy = sum(a.elapsed_time(b) for a, b in spans) * 1e-3
        return {"ms_per_frame": round(wall / (k * per_step) * 1e3, 3), "gpu_span_share": round(busy / wall, 3), "frames_per_step": per_step}

    for phase, it0 in (("it_le_100000", 1000), ("it_gt_100000", 100001)):
        late = it0 > 100000
        entry = {}
        order = [i % n for i in range(iters)]
        # (1) the loop as written: load in the loop, host noise stream
        tr = build(late, "host")
        few = order[:max(8, iters // 8)]
        gen = (s2l.data.collate_batch([ds.load_one_frame(i)]) for i in few)
        run(tr, it0, (s2l.data.collate_batch([ds.load_one_frame(i)]) for i in few[:3]), 1)      # warm-up
        entry["train_step_as_written"] = run(tr, it0, gen, 1)
        # (2) prefetching loader + device noise
        tr = build(late, "device")
        with s2l.FramePrefetcher(ds, order[:3]) as pf:
            run(tr, it0, pf, 1)
        with s2l.FramePrefetcher(ds, order) as pf:
            entry["train_step_prefetch_device_noise"] = run(tr, it0, pf, 1)
        # (2b) the same loop, `Trainer(fused_step=True)`: train_step sends its one frame through the fused engine
        tr = build(late, "device", fused_step=True)
        with s2l.FramePrefetcher(ds, order[:3]) as pf:
            run(tr, it0, pf, 1)
        with s2l.FramePrefetcher(ds, order) as pf:
            entry["train_step_fused_prefetch_device_noise"] = run(tr, it0, pf, 1)
        # (3) K frames per optimisation step through the fused engine: (a) round 5's loop -- torch.optim.Adam, host tensors from the loader,
        #     every step waited for; (b) the loop that keeps the device fed: FusedAdam (one launch, NaN flags folded in), the loader uploading on
        #     a side stream, step k's results read after step k + 1 is queued
        K = 8
        with s2l.FramePrefetcher(ds, order[:2 * K], per_step=K, depth=2 * K, collate=False) as pf:
            run(tr, it0, pf, K)
        with s2l.FramePrefetcher(ds, order, per_step=K, depth=3 * K, collate=False) as pf:
            entry["train_steps_K8_sync_torch_adam"] = run(tr, it0, pf, K)
        #     ... and a loader that leaves the sync-loss side inputs out of the frames while `it` <= 100000 (the reference reads them for every
        #     frame once use_syncloss is configured -- most of a frame's load time -- and only looks at them after it > 100000, training.py:491)
        tr = build(late, "device", fused_step=True, adam=s2l.FusedAdam)
        kw = dict(per_step=K, collate=False, device=dev, sync_fields=late, workers=int(os.environ.get("S2L_BENCH_K8_WORKERS", "0")) or None)
        with s2l.FramePrefetcher(ds, order[:2 * K], depth=2 * K, **kw) as pf:
            run(tr, it0, pf, K, pipelined=True)
        # (the interpreter hands the lock over every 5 ms by default: with eight loader threads taking turns the training thread's launches
        #  arrive in bursts: 200 us in this leg -- same-call A/B 1.90 -> 1.74 ms per frame early, 5.74 -> 5.41 late; S2L_BENCH_SWITCH_US=0 keeps
        #  the default)
        sw_old = sys.getswitchinterval()
        sw_us = float(os.environ.get("S2L_BENCH_SWITCH_US", "200"))
        if sw_us > 0:
            sys.setswitchinterval(sw_us * 1e-6)
        try:
            with s2l.FramePrefetcher(ds, order, depth=3 * K, **kw) as pf:
                entry["train_steps_K8"] = run(tr, it0, pf, K, pipelined=True)
        finally:
            sys.setswitchinterval(sw_old)
        res[phase] = entry
        del tr
        torch.cuda.empty_cache()
    shutil.rmtree(os.path.dirname(root), ignore_errors=True)
    return res
