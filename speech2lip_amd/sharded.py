"""Frame-sharded clip rendering across the GPUs of one node.

Every frame is a pure function of (weights, audio window, frame index): ranks render disjoint
frames with NO data-path dependency, and the only collective is the all-gather that reassembles
the clip on every rank (RCCL over xGMI when the backend is "nccl").  The clip is cut into
`n_chunks` chunks; inside a chunk rank r owns the contiguous block
    [chunk_start + r*per, chunk_start + (r+1)*per),
so each chunk's `all_gather_into_tensor` lands directly in global frame order, and chunk c's
gather (issued asynchronously, on the process group's own stream) overlaps chunk c+1's render.
Because frames are independent the assembled clip is bit-identical to a 1-GPU render.

The reference has no inference sharding (inference.py is single-process); this is the build's
multi-GPU counterpart of its per-frame loop (inference.py:140).
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def chunk_plan(frames_per_rank: int, world: int, n_chunks: int, quantum: int = 1) -> List[Tuple[int, int]]:
    """[(local_offset, count)] per chunk: `frames_per_rank` split into <= n_chunks chunks whose
    sizes are multiples of `quantum` frames (the last chunk takes the remainder).  Identical on
    every rank.  quantum=48 keeps every chunk launch of the 96x96 renderer a whole number of
    tile waves (576 pixel groups x 4 frame groups = 9 x 256 tiles)."""
    if frames_per_rank < 0 or world < 1 or n_chunks < 1 or quantum < 1:
        raise ValueError("bad shard plan arguments")
    n_chunks = max(1, min(n_chunks, frames_per_rank)) if frames_per_rank else 1
    units, tail = divmod(frames_per_rank, quantum)
    if units < n_chunks:          # too short to quantise: plain near-equal split
        quantum, units, tail = 1, frames_per_rank, 0
    base, rem = divmod(units, n_chunks)
    plan, off = [], 0
    for c in range(n_chunks):
        cnt = (base + (1 if c < rem else 0)) * quantum + (tail if c == n_chunks - 1 else 0)
        plan.append((off, cnt))
        off += cnt
    return plan


def global_frame_ids(frames_per_rank: int, rank: int, world: int, n_chunks: int, quantum: int = 1) -> torch.Tensor:
    """Global frame index of each of this rank's local frames, in local order."""
    ids = []
    start = 0
    for off, cnt in chunk_plan(frames_per_rank, world, n_chunks, quantum):
        ids.append(torch.arange(start + rank * cnt, start + (rank + 1) * cnt, dtype=torch.int64))
        start += cnt * world
    return torch.cat(ids) if ids else torch.zeros(0, dtype=torch.int64)


def render_sharded(render_fn: Callable[[int, int, torch.Tensor], None], frames_per_rank: int, frame_shape,
                   device, n_chunks: int = 4, group=None, clip: torch.Tensor = None, gather: bool = True,
                   quantum: int = 1, force_collective: bool = False, quantize: Callable = None):
    """Render this rank's frames chunk by chunk and all-gather each chunk into `clip`.

    render_fn(local_offset, count, out) renders local frames [local_offset, local_offset+count)
    into `out` ([count, *frame_shape], a view of the local fp32 buffer).
    quantize: None -> the clip is gathered as fp32 (bit-identical to a 1-GPU render); or a function
    fp32 tensor -> uint8 tensor (speech2lip_amd.to8b: the 8-bit frames the reference writes to disk,
    inference.py:177), in which case each chunk is quantised on its rank and gathered as uint8 (4x less xGMI traffic).
    Returns (clip [world*frames_per_rank, *frame_shape] in global frame order, local fp32 buffer).
    With world == 1 (or gather=False) no collective is issued and clip aliases the local buffer
    (quantised when `quantize` is given).
    """
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    local = torch.empty((frames_per_rank, *frame_shape), dtype=torch.float32, device=device)
    plan = chunk_plan(frames_per_rank, world, n_chunks, quantum)
    if (world == 1 and not force_collective) or not gather:
        for off, cnt in plan:
            if cnt:
                render_fn(off, cnt, local[off:off + cnt])
        return (quantize(local) if quantize is not None else local), local
    out_dtype = torch.uint8 if quantize is not None else torch.float32
    if clip is None or clip.dtype != out_dtype:
        clip = torch.empty((frames_per_rank * world, *frame_shape), dtype=out_dtype, device=device)
    works, start, keep = [], 0, []
    for off, cnt in plan:
        if cnt:
            render_fn(off, cnt, local[off:off + cnt])
            src = local[off:off + cnt]
            if quantize is not None:
                src = quantize(src)
                keep.append(src)                  # stays alive until its gather has completed
            works.append(dist.all_gather_into_tensor(clip[start:start + cnt * world], src, group=group, async_op=True))
        start += cnt * world
    for w in works:
        w.wait()
    return clip, local


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int, int]:
    """SURVEY.md §8e partitioning of a clip of N frames over G ranks: contiguous blocks of per = ceil(N / G) frames; rank r owns
    [r * per, min((r + 1) * per, N)) -- the last ranks' blocks may be short or empty.  Returns (first, count, per)."""
    if n_frames < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    per = -(-n_frames // world) if n_frames else 0
    first = min(rank * per, n_frames)
    return first, min(per, n_frames - first), per


def gather_clip(local: torch.Tensor, n_frames: int, group=None) -> torch.Tensor:
    """Reassemble a clip from the ranks' contiguous blocks: `local` is this rank's [count, ...] block under `shard_range`
    (any dtype: fp32 lip frames, uint8 final frames); returns [n_frames, ...] in global frame order on every rank.  Short blocks
    are zero-padded to ceil(N / G) for the one `all_gather_into_tensor`, the result is trimmed to N."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.shape[0] != n_frames:
            raise ValueError("single process: the local block must be the whole clip")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count, per = shard_range(n_frames, rank, world)
    if local.shape[0] != count:
        raise ValueError(f"rank {rank} owns {count} frames of {n_frames}, got a block of {local.shape[0]}")
    src = local.contiguous()
    if count < per:
        src = torch.zeros((per, *local.shape[1:]), dtype=local.dtype, device=local.device)
        src[:count].copy_(local)
    clip = torch.empty((world * per, *local.shape[1:]), dtype=local.dtype, device=local.device)
    if per:
        dist.all_gather_into_tensor(clip, src, group=group)
    return clip[:n_frames]


def render_clip_sharded(model, audio, frame_idx, height: int, width: int, group=None, gather: str = "f32", n_chunks: int = 1,
                        quantum: int = 1, render_fn: Callable = None, device=None, force_collective: bool = False,
                        return_local: bool = False, precision: str = "fp32"):
    """The multi-GPU product entry for BASELINE config 4: every rank calls this with the WHOLE clip's inputs
    (audio [N,16,29] -- 1.9 KB per frame -- and the N frame indices, host or device tensors) and gets the whole rendered clip
    [N,H,W,3] back in global frame order; each rank renders only its own contiguous block of ceil(N / G) frames
    and ONE `all_gather_into_tensor` per chunk (RCCL over xGMI with backend "nccl") reassembles it.  N need not divide by G: the
    short ranks' blocks are padded to the common length for the collective (the padding frames are never rendered: they stay
    zero) and the gathered clip is trimmed to N.  The loop being sharded is inference.py:128-140 (one process, batch_size 1,
    frame after frame); the reference's only process-group code is DDP for training (train.py:58-60).

    gather: "f32" -> fp32 frames, bit-identical to `model.render_clip` of the whole clip on one GPU (frames are pure functions
            of (weights, audio window, frame index)); "u8" -> the 8-bit frames the reference writes (inference.py:177),
            quantised on the rank that rendered them, a quarter of the xGMI traffic.
    n_chunks > 1 cuts every rank's block into chunks whose gathers run (async, on the process group's stream) while the next
            chunk renders; the clip is then assembled chunk by chunk (each chunk lands as G contiguous pieces), still in global
            frame order because chunk c of rank r covers frames [r * per + off_c, r * per + off_c + cnt_c).
    precision: "fp32" (default, the exact kernel) or "split" (the opt-in speed mode of TalkingFace.render_clip).
    render_fn(audio_block, idx_block, out_block) replaces `model.render_clip` (tests run a CPU stand-in over gloo).
    Without a process group (or world 1) it is a plain `render_clip` of the whole clip.
    Returns the clip (and this rank's (first, count) with return_local=True)."""
    if gather not in ("f32", "u8"):
        raise ValueError("gather must be 'f32' or 'u8'")
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    H, W = int(height), int(width)
    if device is None:
        device = model.packed_weights().device if render_fn is None else torch.device("cpu")
    audio = torch.as_tensor(audio)
    idx = torch.as_tensor(frame_idx).to(torch.int64).reshape(-1)
    N = int(audio.shape[0])
    if idx.numel() != N:
        raise ValueError("frame_idx must have one entry per audio window")
    if render_fn is None:
        def render_fn(a, i, out):
            model.render_clip(a.to(device), i.to(device), H, W, out=out, precision=precision)
    if gather == "u8":
        from .data import to8b as quantize
    else:
        quantize = None
    first, count, per = shard_range(N, rank, world)

    def finish(clip):
        return (clip, (first, count)) if return_local else clip

    if world == 1 and not force_collective:
        out = torch.empty((N, H, W, 3), dtype=torch.float32, device=device)
        if N:
            render_fn(audio, idx, out)
        return finish(quantize(out) if quantize is not None else out)
    if N == 0:
        return finish(torch.empty((0, H, W, 3), dtype=torch.uint8 if quantize else torch.float32, device=device))
    # this rank's block, padded to `per` frames; padding frames are not rendered
    local = torch.zeros((per, H, W, 3), dtype=torch.float32, device=device) if count < per else \
        torch.empty((per, H, W, 3), dtype=torch.float32, device=device)
    out_dtype = torch.uint8 if quantize is not None else torch.float32
    plan = chunk_plan(per, world, n_chunks, quantum)
    # chunk c is gathered as [G, cnt_c, H, W, 3]; with one chunk that IS the padded clip in frame order
    pieces, works, keep = [], [], []
    for off, cnt in plan:
        if not cnt:
            continue
        valid = max(0, min(cnt, count - off))
        if valid:
            a = audio[first + off:first + off + valid]
            render_fn(a, idx[first + off:first + off + valid], local[off:off + valid])
        src = local[off:off + cnt]
        if quantize is not None:
            src = quantize(src)
            keep.append(src)
        piece = torch.empty((world * cnt, H, W, 3), dtype=out_dtype, device=device)
        works.append(dist.all_gather_into_tensor(piece, src, group=group, async_op=True))
        pieces.append((off, cnt, piece))
    for w_ in works:
        w_.wait()
    if len(pieces) == 1:
        clip = pieces[0][2]
    else:       # re-interleave: rank r's chunk c goes to frames [r * per + off_c, +cnt_c)
        clip = torch.empty((world * per, H, W, 3), dtype=out_dtype, device=device)
        view = clip.view(world, per, H, W, 3)
        for off, cnt, piece in pieces:
            view[:, off:off + cnt].copy_(piece.view(world, cnt, H, W, 3))
    return finish(clip[:N])


def allreduce_grads(grads: dict, group=None, average: bool = True) -> dict:
    """Data-parallel training: the gradients of `LipTrainStep.loss_and_grads` (42 tensors, 691 k
    floats = 2.8 MB) are flattened into ONE bucket, all-reduced once (latency-bound on xGMI, so one collective, not 42) and
    scattered back in place.  Keys are visited in sorted order so that every rank builds the same bucket.  With no process
    group (or world 1) the gradients are returned unchanged."""
    if not (dist.is_available() and dist.is_initialized()):
        return grads
    world = dist.get_world_size(group)
    if world == 1:
        return grads
    keys = sorted(grads)
    flat = torch.cat([grads[k].reshape(-1).to(torch.float32) for k in keys])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= world
    off = 0
    for k in keys:
        n = grads[k].numel()
        grads[k] = flat[off:off + n].reshape(grads[k].shape)
        off += n
    return grads


def broadcast_module_state(module, src: int = 0, group=None) -> None:
    """What DistributedDataParallel does at construction (and, for buffers, before every forward) for the reference's
    `Trainer(multi_gpu=True)` (training.py:41): every rank takes rank `src`'s parameters AND buffers -- the U-Net's BatchNorm
    running statistics included -- so that ranks built or seeded differently start (and keep evaluating) as ONE replica.  One flattened
    bucket per dtype; no-op without an initialised process group or with one rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        if t.numel():
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for (dtype, dev), tensors in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in tensors])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].reshape(t.shape))
                off += t.numel()

