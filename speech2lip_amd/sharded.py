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
