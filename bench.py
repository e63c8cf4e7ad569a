#!/usr/bin/env python3
"""bench.py -- rendered lip frames/s on MI355X (BASELINE.json metric), one rank per GPU.

A "step" renders one clip of FRAMES synthetic audio windows at 96x96 (BASELINE config 2:
8-layer x 256 MLP, 1000 frames) from inputs already resident in HBM: audio encoder ->
per-frame vectors -> fused MLP render -> [FRAMES,96,96,3] fp32 in HBM.  With N > 1 every rank
renders its own FRAMES (weak scaling) and the clip is reassembled on every rank with an RCCL all-gather
(one per step by default; --chunks > 1 splits it into per-chunk gathers issued while the next chunk renders).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without torchrun: spawns its own N ranks
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N = 1 is BASELINE config 2 (1000 frames per step).  N > 1 is config 4's workload: 5 000 frames per GPU per step (40 000 at 8
GPUs), reassembled on every rank; after the timed steps rank 0 re-renders another rank's block locally and
requires the gathered frames to be bit-identical (frames are pure functions of (audio window, frame index)).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the fused
MLP kernel (MFMA-bound, fp32 MFMA peak 157.3 TFLOP/s; algorithmic FLOPs per SURVEY.md §8d) and
`cpu_baseline` (the oracle -- our PyTorch-CPU port of the reference path -- on the host cores).
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W_ = 96
HW = H * W_
FLOPS_PER_FRAME = 2 * 459_520 * HW + 2 * (67_328 + 43_008 + 131_072)   # SURVEY.md §8d (official, factored)
FP32_MFMA_PEAK = 157.3e12


def render_kernel_digest():
    """sha256 (first 16 hex digits) of the generated assembly text of the long-shape render kernel of THIS build
    (speech2lip_amd/build/render_body.inc): the identity of the kernel a PMC figure belongs to."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "speech2lip_amd", "build", "render_body.inc"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def measured_traffic(frames_per_launch):
    """HBM bytes per frame of the render kernel from the latest PMC passes (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3
    runs): tools/summarize_profiles.py writes profiles/render_traffic.json next to the profile summary it was computed from,
    with the digest of the kernel text and the launch size it was measured on (and the figures of earlier profiles of the same kernel text:
    the third return value is their [min, max] per launch -- the counter's spread, +- 20 % between two profiles).  The PMC figure is NOT collected in this run
    (counters need rocprofv3 around the process); it is reported only when it describes this build's kernel at this launch
    size -- otherwise traffic is null and `traffic_source` says why."""
    try:
        with open(os.path.join(ROOT, "profiles", "render_traffic.json")) as f:
            t = json.load(f)
        per_frame = t["hbm_bytes_per_dispatch"] / t["frames_per_dispatch"]
    except (OSError, KeyError, ValueError):
        return None, "no profiles/render_traffic.json", None
    now = render_kernel_digest()
    if t.get("kernel_text_sha256_16") not in (None, now):
        return None, f"stale: {t['profile']} was taken on kernel text {t.get('kernel_text_sha256_16')}, this build is {now}", None
    if int(t["frames_per_dispatch"]) != int(round(frames_per_launch)):
        return None, f"stale: {t['profile']} measured {t['frames_per_dispatch']} frames per launch, this run launches {frames_per_launch:g}", None
    # the counter is noisy (the same kernel text measured 2.03e8 and 2.44e8 bytes per launch in two profiles): quote the range of the figures on record
    seen = [t["hbm_bytes_per_dispatch"]] + [p["hbm_bytes_per_dispatch"] for p in t.get("previous", [])]
    return per_frame, t["profile"], [int(min(seen)), int(max(seen))]


def cpu_baseline(frames_budget_s: float = 12.0):
    """Oracle (kind 'port'): the as-shipped per-frame path (encoder on H*W tiled copies +
    unfactored MLP, inference.py:144-159) on the host cores, bounded sample.  The thread count is CALIBRATED: every
    candidate in {8, 16, 32, 64, 128} (capped at the host's cores) renders one untimed frame and then >= 5 timed ones (a
    candidate that needs more than 4 s for them is cut short and says so); the table is reported and the best candidate
    runs the sample -- the baseline is the best this host does with this code, not the first guess."""
    from oracle import s2l_oracle as O
    from speech2lip_amd import weights as W
    sd = O.to_sd(W.make_state_dict(0, "he"))
    win = torch.from_numpy(W.synthetic_audio(8, seed=1).astype(np.float32))
    ncpu = os.cpu_count() or 1
    with torch.no_grad():
        table = {}
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
            torch.set_num_threads(nt)
            O.render_frame_as_shipped(sd, win[0], 0, H, W_)
            k, t0 = 0, time.perf_counter()
            while k < 5 or (k < 8 and time.perf_counter() - t0 < 1.0):
                O.render_frame_as_shipped(sd, win[k % 8], k, H, W_)
                k += 1
                if time.perf_counter() - t0 > 4.0:
                    break
            table[nt] = round(k / (time.perf_counter() - t0), 2)
        best_nt = max(table, key=table.get)
        torch.set_num_threads(best_nt)
        n, t0 = 0, time.perf_counter()
        while n < 600 and (time.perf_counter() - t0) < frames_budget_s:
            O.render_frame_as_shipped(sd, win[n % 8], n, H, W_)
            n += 1
        dt = time.perf_counter() - t0
        # for fairness (SURVEY.md §8d ii): the same oracle with the encoder run once per frame and 8 frames per call
        O.render_clip(sd, win, list(range(8)), H, W_)
        nb, tb0 = 0, time.perf_counter()
        while nb < 64 and (time.perf_counter() - tb0) < 6.0:
            O.render_clip(sd, win, list(range(nb, nb + 8)), H, W_)
            nb += 8
        dtb = time.perf_counter() - tb0
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "host_cores": ncpu,
            "cpu_model": _cpu_model(), "kind": "port",
            "sample": f"{n} frames 96x96, as-shipped per-frame path (oracle.render_frame_as_shipped), fp32, {dt:.1f} s",
            "threads_calibration_frames_per_s": {str(k): v for k, v in table.items()},
            "batched_value": round(nb / dtb, 3),
            "batched_sample": f"{nb} frames 96x96 in clips of 8 (oracle.render_clip: encoder once per frame), {dtb:.1f} s"}


def eager_torch_gpu(dev, budget_s: float = 6.0):
    """CONTEXT, never `vs_baseline`: the oracle -- plain torch ops (F.conv1d / F.linear through MIOpen / rocBLAS) -- on cuda:0
    of the same MI355X.  The reference as shipped runs on the GPU (inference.py:70 forces CUDA tensors): its per-frame
    sequence (inference.py:144-159: encoder on H*W tiled windows + the unfactored MLP, one frame per call, the `.cpu()` of
    :172 included) is what a maintainer would compare the drop-in with; `render_clip` is the same oracle with the encoder run
    once per frame.  Test-side code: nothing in speech2lip_amd/ imports the oracle."""
    from oracle import s2l_oracle as O
    from speech2lip_amd import weights as W
    sd = {k: v.to(dev) for k, v in O.to_sd(W.make_state_dict(0, "he")).items()}
    win = torch.from_numpy(W.synthetic_audio(8, seed=1).astype(np.float32)).to(dev)
    res = {}
    with torch.no_grad():
        for key, fn, per in (("as_shipped_per_frame", lambda k: O.render_frame_as_shipped(sd, win[k % 8], k, H, W_).cpu(), 1),
                             ("render_clip_8_frames_per_call", lambda k: O.render_clip(sd, win, list(range(k, k + 8)), H, W_).cpu(), 8)):
            for k in range(3):
                fn(k)
            torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while n < 400 * per and time.perf_counter() - t0 < budget_s / 2:
                fn(n)
                n += per
            torch.cuda.synchronize()
            res[key] = round(n / (time.perf_counter() - t0), 1)
        ref = O.render_frame_as_shipped(O.to_sd(W.make_state_dict(0, "he")), win[0].cpu(), 0, H, W_)
        res["max_abs_diff_vs_cpu_oracle"] = float((O.render_frame_as_shipped(sd, win[0], 0, H, W_).cpu() - ref).abs().max())
    res.update(unit="frames/s", what="oracle (eager PyTorch ops) on cuda:0, 96x96, fp32, device->host copy of every frame included; context only")
    return res


def eager_torch_gpu_train(dev, budget_s: float = 10.0):
    """CONTEXT, never `vs_baseline`: the reference's TRAINING step as eager PyTorch on cuda:0 of the same MI355X -- what the reference
    runs (train.py:199 -> Trainer.train_step -> train_stage1, training.py:347-574: autograd over ATen / MIOpen / rocBLAS kernels), here
    through the oracle's restatement of that step (oracle.stage_one_losses: one frame per iteration at 96x96 lip / 500x500 face, MSE on
    lip and face, frozen or training post-fusion U-Net, the 5-frame sync window after it > 100000; no LPIPS -- the AlexNet weights are
    not in the image) + loss.backward() + torch.optim.Adam.  Beside it the MLP-only step of config 5 (4-tap ensemble + MSE + backward
    per frame).  NB the oracle's composite is a gather-based restatement of F.grid_sample, not the ATen kernel: its share is reported.
    Compare with extra.train_bf16* (per sample = ms_per_step / frames) and extra.dropin_trainer (ms per frame)."""
    from oracle import s2l_oracle as O
    from speech2lip_amd import weights as W
    from tools import benchlib
    h = w = 96
    sb = benchlib.sync_batch(dev, 1)
    T = sb["audio_window"].shape[1]
    leaf = lambda v: torch.from_numpy(v).to(dev).requires_grad_(True)
    sd = {k: leaf(v) for k, v in W.make_state_dict(0, "he").items()}
    usd_frozen = {k: v.to(dev) for k, v in O.to_sd(W.make_unet_state_dict(0)).items()}
    usd_train = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "num_batches" not in k else v.clone())
                 for k, v in usd_frozen.items()}
    ssd = {k: v.to(dev) for k, v in O.to_sd(W.make_syncnet_state_dict(0)).items()}
    g = torch.Generator(device=dev).manual_seed(3)
    data = {"audio": torch.from_numpy(W.synthetic_audio(1, 1).astype(np.float32)).to(dev), "rgb": torch.rand(1, h, w, 3, device=dev, generator=g),
            "index": 7, "total_frame": 100000, "rgb_face_zero": sb["rgb_face_canonical"], "rgb_face_ori": sb["rgb_face_gt"][:1],
            "mask_lip_canonical": sb["mask_lip_canonical"], "lip_lefttop_x": sb["lip_lefttop_x"], "lip_lefttop_y": sb["lip_lefttop_y"],
            "coord": sb["coord_window"][0, :1], "audio_window": sb["audio_window"][:1], "coord_window": sb["coord_window"][:1],
            "canonical_face_bbox": [sb["canonical_face_bbox"]], "mel": sb["mel"][:1], "rgb_window_neg": sb["rgb_window_neg"][:1]}
    eps = [0.5] * (1 + T)
    res = {}

    def timed(fn, cap):
        for _ in range(2):
            fn()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while n < cap and time.perf_counter() - t0 < budget_s / 3:
            fn()
            n += 1
        if dev.type == "cuda":
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, n

    # (1) after it > 100000: frozen U-Net in train-mode BatchNorm (what train_step leaves it in, G16), sync loss on
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4)

    def late():
        opt.zero_grad(set_to_none=True)
        out = O.stage_one_losses(sd, usd_frozen, ssd, W.SYNCNET_FACE, W.SYNCNET_AUDIO, data, eps, None, h, w, unet_training=True, with_sync=True)
        out["loss"].backward()
        opt.step()
    ms, n = timed(late, 40)
    res["stage1_iteration_it_gt_100000_ms_per_frame"] = round(ms, 2)
    # (2) before: the U-Net trains too, no sync term
    opt2 = torch.optim.Adam(list(sd.values()) + [v for v in usd_train.values() if v.requires_grad], lr=1e-4)

    def early():
        opt2.zero_grad(set_to_none=True)
        out = O.stage_one_losses(sd, usd_train, None, None, None, data, eps[:1], None, h, w, unet_training=True, with_sync=False)
        out["loss"].backward()
        opt2.step()
    ms, n = timed(early, 40)
    res["stage1_iteration_it_le_100000_ms_per_frame"] = round(ms, 2)
    # (3) config 5's MLP-only step, per frame: 4-tap ensemble + MSE + backward + Adam
    coords = O.get_coords(w, h, device=dev)
    target = data["rgb"].reshape(-1, 3)

    def mlp():
        opt.zero_grad(set_to_none=True)
        pred = O.predict_lip_image(sd, coords, data["audio"][0], 7, h, w, 0.5)
        O.mse_loss(pred, target).backward()
        opt.step()
    ms, n = timed(mlp, 100)
    res["mlp_only_step_ms_per_frame"] = round(ms, 3)
    # the composite's share of (1): the oracle restates grid_sample with gathers (the reference calls the ATen kernel)
    with torch.no_grad():
        lip = torch.rand(1, h, w, 3, device=dev)
        ms_c, _ = timed(lambda: O.composite(lip, data["rgb_face_zero"], data["rgb_face_ori"], data["mask_lip_canonical"], int(data["lip_lefttop_x"]),
                                            int(data["lip_lefttop_y"]), data["coord"]), 100)
    res["composite_forward_ms_per_frame"] = round(ms_c, 3)
    res.update(what="oracle's stage-1 step (eager PyTorch autograd) on cuda:0, one 96x96 / 500x500 frame per iteration, fp32, Adam; context only")
    return res


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _flush_c_stdout():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def spawn_ranks(args) -> int:
    """`--gpus N` (N > 1) outside torchrun: start one rank per GPU ourselves (same env contract as
    torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Rank 0 owns stdout (the JSON
    line); the other ranks' stdout goes to stderr.  Never prints a line whose n_gpus differs from --gpus: with fewer
    visible GPUs than requested this is an error, not a silent 1-GPU run."""
    import socket
    import subprocess
    n_vis = torch.cuda.device_count()
    if n_vis < args.gpus and not args.debug_one_device:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {n_vis} GPU(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    finally:
        for p in procs:          # a failed rank must not leave the others waiting in a collective forever
            if p.poll() is None:
                p.kill()
    return rc


def extra_measurements(dev):
    """Configs 3 and 5 (and the composite kernel alone), measured after and outside the headline's timed region so that
    they are in the driver-timed record too.  Each entry is what the matching tools/bench_*.py prints."""
    from tools import benchlib
    out = {}
    for name, fn in (("composite", lambda: benchlib.bench_composite(dev, 256)),
                     ("render_split", lambda: benchlib.bench_render_split(dev)),
                     ("small_clips", lambda: benchlib.bench_small_clips(dev)),
                     ("config4_rank_block", lambda: benchlib.bench_config4_rank_block(dev)),
                     ("config3", lambda: benchlib.bench_config3(dev, 5000, 500)),
                     ("config3_with_unet", lambda: benchlib.bench_config3(dev, 1000, 100, unet=True)),
                     ("config3_with_unet_split_bf16", lambda: benchlib.bench_config3(dev, 1000, 100, unet=True, unet_precision="split")),
                     ("config3_all_split", lambda: benchlib.bench_config3(dev, 1000, 100, unet=True, unet_precision="split", lip_precision="split")),
                     ("config3_lip_split", lambda: benchlib.bench_config3(dev, 5000, 500, lip_precision="split")),
                     ("unet_fp32", lambda: benchlib.bench_unet(dev, 16)),
                     ("train_bf16", lambda: benchlib.bench_train(dev, 64, "bf16")),
                     ("train_bf16_with_sync_loss", lambda: benchlib.bench_train_sync(dev, 64, 8, "bf16")),
                     ("train_bf16_with_sync_loss_trainmode_bn", lambda: benchlib.bench_train_sync(dev, 64, 8, "bf16", unet_train_mode=True)),
                     ("train_bf16_with_sync_loss_trainmode_bn_fp32_tensors",
                      lambda: benchlib.bench_train_sync(dev, 64, 8, "bf16", unet_train_mode=True, half_width_tensors=False)),
                     ("stage1_full_iteration_bf16", lambda: benchlib.bench_stage1_full(dev, 8, "bf16")),
                     ("stage1_full_iteration_bf16_trainmode_bn", lambda: benchlib.bench_stage1_full(dev, 8, "bf16", unet_train_mode=True)),
                     ("stage1_early_iteration_bf16", lambda: benchlib.bench_stage1_full(dev, 8, "bf16", early=True)),
                     ("sync_loss", lambda: benchlib.bench_sync_loss(dev, 16)),
                     ("train_fp32", lambda: benchlib.bench_train(dev, 64, "fp32", steps=3)),
                     ("dropin_trainer", lambda: benchlib.bench_dropin_trainer(dev)),
                     ("infer_clip_end_to_end", lambda: benchlib.bench_infer_clip(dev)),
                     ("eager_torch_gpu", lambda: eager_torch_gpu(dev)),
                     ("eager_torch_gpu_train", lambda: eager_torch_gpu_train(dev))):
        try:
            torch.cuda.reset_peak_memory_stats()      # every leg's peak_mem_gb is its own
            r = fn()
            out[name] = {k: v for k, v in r.items() if not k.startswith("_")}
        except Exception as e:      # an extra must never cost the headline line
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


_DROP_KEYS = {"loss_first", "loss_last", "loss_sync_last", "loss_perceptual_last", "loss_face_last", "unet_window", "frames_checked",
              "lip_gflop_per_frame", "gflop_per_frame", "algorithmic_bytes_per_frame", "peak", "unit", "bound", "mlp_frames_per_step",
              "as_written_tflop_per_step", "mlp_tflop_per_step", "unet_tflop_per_step", "sync_latency_us_per_call", "seconds",
              "exact_kernel_parity", "exact_kernel_ms_per_clip", "max_abs_err", "per_frame_cost_over_long_clip"}


def compact(v, digits=4):
    """The `extra` object as it goes into the ONE JSON line: numbers only.  Every descriptive string (workload, kernel,
    arithmetic: docs/BENCH_LEGEND.md holds them, leg by leg) and the secondary fields are dropped, floats keep `digits`
    significant digits -- the driver keeps ~8 KB of stdout tail and the whole line has to fit in it.  The uncut object is
    written to gpurun_out/bench_extra_full.json."""
    if isinstance(v, dict):
        return {k: compact(x, digits) for k, x in v.items()
                if k not in _DROP_KEYS and not (isinstance(x, str) and len(x) > 28 and k != "error")}
    if isinstance(v, (list, tuple)):
        return [compact(x, digits) for x in v]
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=None,
                    help="frames per GPU per step; default 1000 at N = 1 (BASELINE config 2), 5000 at N > 1 (config 4: 40k / 8)")
    ap.add_argument("--chunks", type=int, default=None,
                    help="all-gather chunks per step (N > 1).  Default: AUTO -- during warm-up one step of each implemented schedule "
                         "({1 chunk, no CUs reserved} and {4 chunks, 8 CUs left to RCCL}) is timed (max over ranks) and the timed steps "
                         "run the faster one; both times are printed in multi_gpu.schedules.  1 = render the clip in one persistent launch, then one "
                         "all-gather: the renderer fills every CU (151 KiB LDS + all registers per workgroup), so an "
                         "RCCL kernel overlapped with it can only start on CUs a finished workgroup has released and "
                         "then delays the statically-striped workgroups of the next launch; >1 enables the overlap")
    ap.add_argument("--gather", choices=["f32", "u8"], default="f32",
                    help="dtype of the reassembled clip (N > 1): f32 = bit-identical to a 1-GPU render (default); u8 = the 8-bit "
                         "frames the reference writes (cv2.imwrite semantics), quantised per rank, 4x less all-gather traffic")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="leave this many CUs to RCCL (use with --chunks > 1): the renderer launches CUs - k persistent workgroups, "
                         "so the all-gather of one chunk can run while the next chunk renders.  Default: 0 with an explicit --chunks, AUTO otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3 / config-5 `extra` measurements (N = 1)")
    ap.add_argument("--force-chunks", action="store_true", help="debug: chunked launches at N=1 (measures chunking overhead)")
    ap.add_argument("--debug-one-device", action="store_true",
                    help="debug / tests: every rank uses cuda:0 and the process group runs on gloo (host-staged collectives): the whole "
                         "N > 1 control flow -- schedule selection, gathers, barriers, remote-block and ragged-clip checks -- on a "
                         "ONE-GPU box.  Timings of such a run mean nothing and the line says so")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: take the RCCL path (process group, all-gather, barriers) even with one rank")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:      # the line's n_gpus must be what --gpus asked for
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.debug_one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    dev_list = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.debug_one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        # RCCL sets up its rings lazily on the first collective of each kind: do that here, outside any step, so that a
        # run with --warmup 0 does not time communicator setup
        t = torch.zeros(1, device=dev)
        dist.all_reduce(t)
        g = torch.zeros(world * 4, device=dev)
        dist.all_gather_into_tensor(g, g[rank * 4:rank * 4 + 4].clone())
        torch.cuda.synchronize()
        _flush_c_stdout()      # RCCL's version banner (NCCL_DEBUG=VERSION) leaves every rank's C stdio buffer now, not at exit
        # which device every rank sits on (index + uuid), collected THROUGH the process group: N distinct entries in
        # multi_gpu.devices = the backend really connected N GPUs
        dev_list = [None] * world
        dist.all_gather_object(dev_list, f"cuda:{local_rank} {getattr(torch.cuda.get_device_properties(dev), 'uuid', '')}")

    import speech2lip_amd as s2l
    from speech2lip_amd import sharded, weights as W

    F = args.frames if args.frames is not None else (1000 if world == 1 else 5000)
    from speech2lip_amd import _abi
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    model = s2l.TalkingFace(dev, s2l.may_config(H, W_), mode="eval").eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
    QUANTUM = 48   # frames: keeps each chunk launch a whole number of 256-tile waves at 96x96
    chunked = world > 1 or args.force_chunks
    quant = s2l.to8b if args.gather == "u8" else None
    clip = torch.empty((F * world, H, W_, 3), dtype=torch.uint8 if quant else torch.float32, device=dev) if use_dist else None
    kernel_events = []

    class Schedule:
        """One way of running a step: how many chunks (= launches + all-gathers) and how many CUs the renderer leaves to RCCL."""

        def __init__(self, n_chunks, reserve):
            self.n_chunks, self.reserve = (n_chunks if chunked else 1), reserve
            self.audio, self.gids = self.rank_inputs(rank)                                      # resident in HBM
            self.name = f"{self.n_chunks}-chunk" + (f"+{reserve}cu-reserved" if reserve else "")

        def rank_inputs(self, r):      # audio windows and global frame ids of rank r (any rank can rebuild any other rank's)
            ids = sharded.global_frame_ids(F, r, world, self.n_chunks, QUANTUM).to(dev)
            return torch.from_numpy(W.synthetic_audio(F, seed=1 + r).astype(np.float32)).to(dev), ids

        def activate(self):            # s2l_set_render_cus is process-global: set it for the steps that follow
            _abi.check(_abi.load().s2l_set_render_cus(max(1, n_cu - self.reserve) if self.reserve else 0), "s2l_set_render_cus")

        def render(self, off, cnt, out):
            model.render_clip(self.audio[off:off + cnt], self.gids[off:off + cnt], H, W_, out=out, _events=kernel_events)

        def step(self, gather=True):
            return sharded.render_sharded(self.render, F, (H, W_, 3), dev, n_chunks=self.n_chunks, clip=clip, gather=gather,
                                          quantum=QUANTUM, force_collective=args.force_dist, quantize=quant if use_dist else None)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n, fn):        # barrier + sync on both sides, MAX over ranks
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, r

    # ---- schedule: explicit flags, or (N > 1) the faster of the implemented ones, measured here during warm-up -------------
    auto = world > 1 and args.chunks is None and args.reserve_cus is None
    schedule_times = None
    if auto:
        candidates = [Schedule(1, 0), Schedule(4, 8)]
        schedule_times = {}
        for c in candidates:
            c.activate()
            c.step()                                  # first launch of this form (allocations, RCCL channel set-up): untimed
            dt_c, _ = timed(1, c.step)                # MAX over ranks, so every rank picks the same schedule
            schedule_times[c.name] = round(dt_c * 1e3, 3)
        sched = min(candidates, key=lambda c: schedule_times[c.name])
        del candidates
    else:
        sched = Schedule(args.chunks or 1, args.reserve_cus or 0)
    sched.activate()
    n_chunks, audio, gids, step = sched.n_chunks, sched.audio, sched.gids, sched.step

    for _ in range(args.warmup):
        step()
    fence()
    kernel_events.clear()
    dt, (out, local) = timed(args.steps, step)

    # dominant-kernel time from HIP events recorded on the launch stream around s2l_render_lip
    torch.cuda.synchronize()
    k_ms = [s.elapsed_time(e) for s, e in kernel_events]
    # several launches per step when chunked: total algorithmic FLOPs of a step / total kernel time of a step
    launches_per_step = len(k_ms) / args.steps
    frames_per_launch = F / launches_per_step
    k_avg_s = (sum(k_ms) / len(k_ms)) * 1e-3
    achieved = FLOPS_PER_FRAME * frames_per_launch / k_avg_s

    multi = None
    if use_dist:
        # where the step goes: the same steps without the gather, and the gather alone (both barrier-bracketed, MAX over ranks)
        n_aux = max(1, min(args.steps, 5))
        dt_render, _ = timed(n_aux, lambda: step(gather=False))
        src = quant(local) if quant else local

        def gather_only():
            dist.all_gather_into_tensor(clip, src)
        dt_gather, _ = timed(n_aux, gather_only)
        # the ONE-GPU reference at THIS step size: every rank renders its F frames in one full-chip launch, no collective anywhere in
        # the timed region (ranks are independent; barrier-bracketed, MAX over ranks).  The driver's N = 1 run is config 2 (1000-frame
        # steps); weak-scaling efficiency at config 4's 5 000 frames per GPU is value / (N * this rate).
        one = Schedule(1, 0)
        one.activate()
        one.step(gather=False)
        dt_one, _ = timed(n_aux, lambda: one.step(gather=False))
        del one
        sched.activate()
        # cross-rank verification: rank 0 re-renders the LAST rank's block itself and compares with what the gather delivered
        step()
        torch.cuda.synchronize()
        verified = None
        if rank == 0:
            r = world - 1
            a_r, ids_r = sched.rank_inputs(r)
            mine = model.render_clip(a_r, ids_r, H, W_)
            theirs = clip[ids_r]                      # rank r's frames sit at their global frame ids, whatever the chunking
            verified = bool(torch.equal(quant(mine) if quant else mine, theirs))
            if not verified:
                raise SystemExit(f"bench.py: gathered frames of rank {r} differ from a local re-render of the same frame ids")
        # the product entry on a RAGGED clip (N % G != 0): sharded.render_clip_sharded pads the short blocks, gathers, trims;
        # rank 0 compares the whole clip with its own one-GPU render (bit for bit)
        n_rag = max(1, 48 * world - 5)
        a_rag = torch.from_numpy(W.synthetic_audio(n_rag, seed=99).astype(np.float32)).to(dev)
        i_rag = torch.arange(39_000, 39_000 + n_rag, device=dev)
        got_rag = sharded.render_clip_sharded(model, a_rag, i_rag, H, W_, force_collective=args.force_dist)
        torch.cuda.synchronize()
        ragged_ok = bool(torch.equal(got_rag, model.render_clip(a_rag, i_rag, H, W_))) if rank == 0 else None
        if rank == 0 and not ragged_ok:
            raise SystemExit("bench.py: render_clip_sharded on a ragged clip differs from the one-GPU render")
        multi = {"render_only_ms": round(dt_render / n_aux * 1e3, 3), "gather_only_ms": round(dt_gather / n_aux * 1e3, 3),
                 "gather_bytes_per_rank": int(src.numel() * src.element_size()),
                 "per_gpu_rate_with_gather_over_without": round((dt_render / n_aux) / (dt / args.steps), 4),
                 "one_gpu_same_frames_per_step": {"frames_per_step": F, "ms_per_step": round(dt_one / n_aux * 1e3, 3),
                                                  "frames_per_s": round(F * n_aux / dt_one, 1)},
                 "remote_block_bit_identical_to_local_render": verified,
                 "ragged_clip": {"frames": n_rag, "bit_identical_to_one_gpu_render": ragged_ok},
                 "schedule": sched.name, "schedule_selection": "auto (timed during warm-up, max over ranks)" if auto else "flags",
                 "schedules_ms_per_step": schedule_times,
                 "backend": dist.get_backend(), "ranks_in_process_group": dist.get_world_size(),
                 "rccl_version": ".".join(map(str, torch.cuda.nccl.version())) if dist.get_backend() == "nccl" else None,
                 "devices": sorted(set(dev_list)) if dev_list else None,
                 "note": "weak-scaling efficiency = value_N / (N * value_1) is computed by the driver from its own runs"}

    if rank == 0:
        # parity spot-check outside the timed region: frame 0 against the CPU oracle
        from oracle import s2l_oracle as O
        with torch.no_grad():
            ref = O.render_clip(O.to_sd(W.make_state_dict(0, "he")), audio[:1].cpu(), [int(gids[0])], H, W_)[0]
        got = local[0].cpu()
        traffic_per_frame, traffic_src, traffic_range = measured_traffic(frames_per_launch)
        line = {
            "metric": "rendered lip frames/sec (96x96)", "value": round(F * world * args.steps / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"INVALID_debug_one_device": "all ranks shared cuda:0 over gloo: control-flow test, not a measurement"} if args.debug_one_device else {}),
            "config": {"workload": f"May face_simple 96x96 lip crop, 8-layer x256 v2 MLP, {F} synthetic audio frames per GPU per step"
                       + (" (BASELINE config 2)" if world == 1 and F == 1000 else "")
                       + (f" (BASELINE config 4 workload: {F * world} frames over {world} GPUs)" if world > 1 else ""),
                       "frames_per_gpu": F, "height": H, "width": W_, "parallelism": f"frame-shard x{world}" +
                       (f" + {n_chunks}-chunk {args.gather} all-gather" if world > 1 else "") +
                       (f", {sched.reserve} CUs left to RCCL" if sched.reserve else "")},
            "roofline": {"bound": "mfma", "achieved": round(achieved / 1e12, 3), "peak": FP32_MFMA_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK, 4),
                         # HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, separate passes), see profiles/
                         "traffic": round(traffic_per_frame * frames_per_launch) if traffic_per_frame else None,
                         "traffic_source": traffic_src, "traffic_range_of_profiles": traffic_range,
                         "kernel": "s2l::render_tiles_kernel (s2l_render_lip)", "kernel_ms": round(k_avg_s * 1e3, 4),
                         "frames_per_launch": frames_per_launch, "algorithmic_gflop_per_frame": round(FLOPS_PER_FRAME / 1e9, 4)},
            "parity": {"rmse_vs_cpu": float(f"{O.rmse(got, ref):.3e}"), "psnr_db_vs_cpu": round(O.psnr(got, ref), 1)},
        }
        if multi:
            line["multi_gpu"] = multi
    if use_dist:
        _flush_c_stdout()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1:
            del out, local, clip
            torch.cuda.empty_cache()
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline()
            if not args.no_extra:
                full = extra_measurements(dev)
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    with open(os.path.join(ROOT, "gpurun_out", "bench_extra_full.json"), "w") as f:
                        json.dump(full, f, indent=1)
                except OSError:
                    pass
                line["extra"] = compact(full)
                line["extra"]["legend"] = "docs/BENCH_LEGEND.md"
        _flush_c_stdout()      # the JSON line must be the LAST line on stdout
        print(json.dumps(line, separators=(",", ":")), flush=True)


if __name__ == "__main__":
    main()
