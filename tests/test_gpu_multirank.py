"""BASELINE config 4's collective path on real GPUs: process group on RCCL ("nccl"), frame-sharded renders, the
`all_gather_into_tensor` that reassembles the clip, the uint8 gather, and the remote-block bit check.

The reference has no inference sharding (its only `init_process_group` is DDP for training, train.py:58-60), so there
is nothing of its own to mirror here: the contract is SURVEY.md §8e -- the assembled N-GPU clip equals the 1-GPU clip
bit for bit, because every frame is a pure function of (weights, audio window, frame index).

Leg (a) runs on ONE GPU: `bench.py --force-dist` takes the RCCL path with a single rank (communicator set-up, the IPC
environment default `HSA_ENABLE_IPC_MODE_LEGACY=0` of bench.py, barriers, the gather into the preallocated clip,
rank 0's re-render of the last rank's block), and `sharded.render_sharded(force_collective=True)` is driven in-process
with the real renderer.  Leg (b) needs >= 2 visible GPUs and spawns min(device_count, 8) ranks; it is skipped on the
driver's 1-GPU box and validates itself on the first multi-GPU node it meets.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = W_ = 96


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_bench(*flags, timeout=900, env_extra=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):      # never inherit a launcher's rank variables
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, f"bench.py {' '.join(flags)} -> rc {p.returncode}\n{p.stderr[-3000:]}"
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert lines, p.stderr[-2000:]
    return json.loads(lines[-1])         # the JSON line is the LAST line on stdout (bench.py contract)


COMMON = ("--steps", "1", "--warmup", "1", "--no-extra", "--no-cpu-baseline")


@pytest.mark.parametrize("flags", [
    (),                                              # one fp32 all-gather per step
    ("--force-chunks", "--chunks", "2"),             # per-chunk async gathers (the overlap form)
    ("--gather", "u8"),                              # the 8-bit frames the reference writes, gathered as uint8
], ids=["f32-1chunk", "f32-2chunks", "u8"])
def test_bench_force_dist_single_rank(flags):
    line = _run_bench("--force-dist", "--frames", "96", *COMMON, *flags)
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["unit"] == "frames/s" and line["value"] > 0
    mg = line["multi_gpu"]
    assert mg["gather_only_ms"] > 0 and mg["render_only_ms"] > 0
    bytes_per_frame = H * W_ * 3 * (1 if "u8" in flags else 4)
    assert mg["gather_bytes_per_rank"] == 96 * bytes_per_frame
    assert mg["remote_block_bit_identical_to_local_render"] is True           # any chunking: frames sit at their global ids
    assert mg["ragged_clip"]["bit_identical_to_one_gpu_render"] is True and mg["ragged_clip"]["frames"] == 43
    assert mg["schedule_selection"] == "flags" and mg["schedule"] == ("2-chunk" if "--chunks" in flags else "1-chunk")
    if "--chunks" in flags:
        assert "2-chunk" not in line["config"]["parallelism"]                 # world 1: the label only names real gathers
    assert line["parity"]["psnr_db_vs_cpu"] >= 90.0
    assert line["roofline"]["frames_per_launch"] == (48 if "--chunks" in flags else 96)


def test_bench_sets_the_dmabuf_ipc_default():
    """RCCL on this pool needs HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver only supports dmabuf IPC); bench.py must
    supply it when the caller's environment does not."""
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    code = ("import os, runpy, sys; sys.argv=['bench.py','--help']\n"
            "try:\n    runpy.run_path('bench.py', run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "print('IPC=' + os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset'))")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert "IPC=0" in p.stdout, p.stdout[-500:] + p.stderr[-500:]


@pytest.fixture(scope="module")
def single_rank_group():
    import torch.distributed as dist
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    yield dev
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("frames,n_chunks,quantum", [(96, 1, 48), (100, 2, 48), (37, 3, 1)])
def test_render_sharded_collective_equals_direct_render(single_rank_group, frames, n_chunks, quantum):
    """sharded.render_sharded with the REAL renderer through the RCCL all-gather (one rank) == model.render_clip."""
    import speech2lip_amd as s2l
    from speech2lip_amd import sharded, weights as W
    from tests.test_gpu_parity import make_model
    dev = single_rank_group
    m = make_model(dev, H, W_)
    audio = torch.from_numpy(W.synthetic_audio(frames, seed=11).astype(np.float32)).to(dev)
    gids = sharded.global_frame_ids(frames, 0, 1, n_chunks, quantum).to(dev)
    assert gids.tolist() == list(range(frames))
    direct = m.render_clip(audio, gids, H, W_)

    def render(off, cnt, out):
        m.render_clip(audio[off:off + cnt], gids[off:off + cnt], H, W_, out=out)

    pre = torch.full((frames, H, W_, 3), float("nan"), device=dev)            # gathered INTO a caller-owned clip
    clip, local = sharded.render_sharded(render, frames, (H, W_, 3), dev, n_chunks=n_chunks, clip=pre, quantum=quantum,
                                         force_collective=True)
    torch.cuda.synchronize()
    assert clip.data_ptr() == pre.data_ptr() and clip.data_ptr() != local.data_ptr()
    assert torch.equal(clip, direct) and torch.equal(local, direct)
    clip8, _ = sharded.render_sharded(render, frames, (H, W_, 3), dev, n_chunks=n_chunks, quantum=quantum,
                                      force_collective=True, quantize=s2l.to8b)
    torch.cuda.synchronize()
    assert clip8.dtype == torch.uint8 and torch.equal(clip8, s2l.to8b(direct))


@pytest.mark.parametrize("n,n_chunks,gather", [(37, 1, "f32"), (50, 3, "f32"), (37, 1, "u8"), (1, 1, "f32")])
def test_render_clip_sharded_product_entry_on_rccl(single_rank_group, n, n_chunks, gather):
    """The product entry (sharded.render_clip_sharded: whole-clip inputs in, whole clip out, blocks of ceil(N / G)) through the
    RCCL all-gather with one rank == model.render_clip; host-side inputs are accepted (audio.npy comes from disk)."""
    import speech2lip_amd as s2l
    from speech2lip_amd import sharded, weights as W
    from tests.test_gpu_parity import make_model
    dev = single_rank_group
    m = make_model(dev, H, W_)
    audio = torch.from_numpy(W.synthetic_audio(n, seed=5).astype(np.float32))          # on the HOST
    idx = torch.arange(39_990, 39_990 + n)
    direct = m.render_clip(audio.to(dev), idx.to(dev), H, W_)
    clip, (first, count) = sharded.render_clip_sharded(m, audio, idx, H, W_, gather=gather, n_chunks=n_chunks,
                                                       force_collective=True, return_local=True)
    torch.cuda.synchronize()
    assert (first, count) == (0, n) and clip.shape == (n, H, W_, 3)
    assert torch.equal(clip, s2l.to8b(direct) if gather == "u8" else direct)
    whole = sharded.gather_clip(s2l.to8b(direct), n)                                     # the final-frames form (tools/infer_clip.py --gather)
    assert torch.equal(whole, s2l.to8b(direct))


def test_allreduce_grads_on_rccl(single_rank_group):
    """The data-parallel gradient bucket of config 5 on the RCCL backend (world 1: identity, but through the same calls)."""
    from speech2lip_amd import sharded
    dev = single_rank_group
    g = {"b": torch.arange(6, dtype=torch.float32, device=dev).reshape(2, 3), "a": torch.ones(5, device=dev)}
    ref = {k: v.clone() for k, v in g.items()}
    out = sharded.allreduce_grads(g)
    for k in ref:
        assert torch.equal(out[k], ref[k]) and out[k].shape == ref[k].shape


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs (the driver's test box has one)")
@pytest.mark.parametrize("gather", ["f32", "u8"])
def test_bench_n_ranks_bit_check(gather):
    n = min(torch.cuda.device_count(), 8)
    line = _run_bench("--gpus", str(n), "--frames", "480", *COMMON, "--gather", gather, timeout=1800)
    assert line["n_gpus"] == n and line["scaling"] == "weak"
    assert line["config"]["frames_per_gpu"] == 480
    mg = line["multi_gpu"]
    assert mg["remote_block_bit_identical_to_local_render"] is True
    assert mg["ragged_clip"]["bit_identical_to_one_gpu_render"] is True and mg["ragged_clip"]["frames"] == 48 * n - 5     # N % G != 0
    assert mg["schedule_selection"].startswith("auto") and set(mg["schedules_ms_per_step"]) == {"1-chunk", "4-chunk+8cu-reserved"}
    assert mg["schedule"] in mg["schedules_ms_per_step"]
    assert mg["gather_bytes_per_rank"] == 480 * H * W_ * 3 * (4 if gather == "f32" else 1)
    # whole-job frames: every rank's 480 frames per step
    assert abs(line["value"] - 480 * n * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) <= 0.01 * line["value"]


@pytest.mark.parametrize("n,extra", [(2, ()), (3, ("--gather", "u8")), (2, ("--chunks", "3", "--reserve-cus", "8")), (8, ())],
                         ids=["2-auto", "3-u8-auto", "2-flags", "8-auto"])
def test_bench_multi_rank_control_flow_on_one_gpu(n, extra):
    """The WHOLE N > 1 path of bench.py with N real ranks -- schedule selection timed during warm-up (max over ranks), per-chunk
    gathers, barriers, rank 0's re-render of the last rank's block, the ragged clip through sharded.render_clip_sharded (N % G != 0)
    -- on the one GPU this box has: `--debug-one-device` puts every rank on cuda:0 and the process group on gloo.  The first
    multi-GPU node then only adds RCCL itself, which the one-rank legs above already exercise."""
    line = _run_bench("--gpus", str(n), "--frames", "96", *COMMON, "--debug-one-device", *extra, timeout=1800)
    assert line["n_gpus"] == n and "INVALID_debug_one_device" in line
    mg = line["multi_gpu"]
    assert mg["remote_block_bit_identical_to_local_render"] is True
    assert mg["ragged_clip"] == {"frames": 48 * n - 5, "bit_identical_to_one_gpu_render": True}
    if "--chunks" in extra:
        assert mg["schedule_selection"] == "flags" and mg["schedule"] == "3-chunk+8cu-reserved"
    else:
        assert mg["schedule_selection"].startswith("auto") and set(mg["schedules_ms_per_step"]) == {"1-chunk", "4-chunk+8cu-reserved"}
        assert mg["schedule"] == min(mg["schedules_ms_per_step"], key=mg["schedules_ms_per_step"].get)
    assert mg["gather_bytes_per_rank"] == 96 * H * W_ * 3 * (1 if "u8" in extra else 4)
    assert line["config"]["frames_per_gpu"] == 96 and line["parity"]["psnr_db_vs_cpu"] >= 90.0
    one = mg["one_gpu_same_frames_per_step"]      # the 1-GPU reference at THIS step size (the driver's N = 1 run uses 1000-frame steps)
    assert one["frames_per_step"] == 96 and one["frames_per_s"] > 0 and one["ms_per_step"] > 0
