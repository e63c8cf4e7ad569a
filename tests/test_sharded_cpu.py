"""CPU, world_size 2 over gloo: the frame-shard plan and the chunked all-gather assemble the clip
in global frame order, identical to the 1-process result.  The renderer is a
stand-in that stamps each frame with a function of its global index (the real one needs a GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speech2lip_amd import sharded


def test_chunk_plan_covers_everything_once():
    for fpr in (0, 1, 5, 7, 1000):
        for world in (1, 2, 8):
            for nc in (1, 3, 4, 16):
                for quantum in (1, 48):
                    plan = sharded.chunk_plan(fpr, world, nc, quantum)
                    assert sum(c for _, c in plan) == fpr
                    assert all(c % quantum == 0 for _, c in plan[:-1]) or fpr < quantum * len(plan)
                    ids = torch.cat([sharded.global_frame_ids(fpr, r, world, nc, quantum) for r in range(world)])
                    assert sorted(ids.tolist()) == list(range(fpr * world))
    assert [c for _, c in sharded.chunk_plan(1000, 8, 4, 48)] == [240, 240, 240, 280]


def _stamp(gid):
    return (gid.double() * 0.37 + 1.0).float()


def _worker(rank, world, port, fpr, nc, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gids = sharded.global_frame_ids(fpr, rank, world, nc)

        def render(off, cnt, out):
            out.copy_(_stamp(gids[off:off + cnt]).view(-1, 1, 1).expand(cnt, 2, 3))

        clip, local = sharded.render_sharded(render, fpr, (2, 3), torch.device("cpu"), n_chunks=nc)
        expect = _stamp(torch.arange(fpr * world)).view(-1, 1, 1).expand(-1, 2, 3)
        ok = torch.equal(clip, expect) and local.shape[0] == fpr
        # uint8 gather: every rank quantises its own chunk (cv2.imwrite rounding), the clip arrives as uint8
        from speech2lip_amd.data import to8b
        clip8, _ = sharded.render_sharded(lambda off, cnt, out: out.copy_((_stamp(gids[off:off + cnt]) / 16).view(-1, 1, 1).expand(cnt, 2, 3)),
                                          fpr, (2, 3), torch.device("cpu"), n_chunks=nc, quantize=to8b)
        ok = ok and clip8.dtype == torch.uint8 and torch.equal(clip8, to8b((expect / 16).contiguous()))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gather_equals_single_process():
    for fpr, nc in [(10, 4), (7, 3), (1, 4)]:
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), fpr, nc, ret), nprocs=2, join=True)
        assert ret[0] and ret[1], (fpr, nc)


def test_single_process_needs_no_collective():
    gids = sharded.global_frame_ids(9, 0, 1, 4)
    assert gids.tolist() == list(range(9))

    def render(off, cnt, out):
        out.copy_(_stamp(gids[off:off + cnt]).view(-1, 1))

    clip, local = sharded.render_sharded(render, 9, (1,), torch.device("cpu"), n_chunks=4)
    assert clip.data_ptr() == local.data_ptr()
    assert torch.equal(clip.view(-1), _stamp(torch.arange(9)))


def _grad_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = {"b.weight": torch.full((3, 4), float(rank + 1)), "a.bias": torch.arange(5.0) * (rank + 1), "c": torch.tensor([2.0 * rank])}
        out = sharded.allreduce_grads(dict(g))
        ok = torch.equal(out["b.weight"], torch.full((3, 4), 1.5)) and torch.equal(out["a.bias"], torch.arange(5.0) * 1.5) \
            and torch.equal(out["c"], torch.tensor([1.0])) and out["b.weight"].shape == (3, 4)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_bucket_allreduce():
    """Config 5 data parallelism: one flattened bucket, averaged, same result on both ranks."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grad_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]
    g = {"x": torch.ones(2)}
    assert sharded.allreduce_grads(g) is g            # no process group: untouched


# ---- the product entry: render_clip_sharded (ragged N, trimmed clip, global frame order) ---------------------------------
def test_shard_range_is_ceil_blocks():
    """contiguous blocks of ceil(N / G) frames; the tail ranks may be short or empty."""
    assert [sharded.shard_range(10, r, 3) for r in range(3)] == [(0, 4, 4), (4, 4, 4), (8, 2, 4)]
    assert [sharded.shard_range(5, r, 4) for r in range(4)] == [(0, 2, 2), (2, 2, 2), (4, 1, 2), (5, 0, 2)]
    assert [sharded.shard_range(2, r, 3) for r in range(3)] == [(0, 1, 1), (1, 1, 1), (2, 0, 1)]
    assert sharded.shard_range(0, 1, 2) == (0, 0, 0) and sharded.shard_range(40000, 7, 8) == (35000, 5000, 5000)
    for n in (0, 1, 7, 16, 1000, 1001):
        for g in (1, 2, 3, 8):
            blocks = [sharded.shard_range(n, r, g) for r in range(g)]
            assert sum(c for _, c, _ in blocks) == n
            assert all(b[0] == sum(c for _, c, _ in blocks[:r]) for r, b in enumerate(blocks))


def _stub_frame(audio_block, idx_block, out):
    """CPU stand-in for the renderer: like the real one, every frame is a pure function of (its audio window, its frame index)."""
    v = audio_block.double().sum(dim=(1, 2)) * 0.001 + idx_block.double() * 0.37
    out.copy_(v.float().view(-1, 1, 1, 1).expand_as(out))


def _clip_inputs(n):
    g = torch.Generator().manual_seed(n)
    return torch.randn(n, 16, 29, generator=g), torch.arange(100, 100 + n)


def _clip_worker(rank, world, port, cases, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from speech2lip_amd.data import to8b
        ok = True
        for n, nc in cases:
            audio, idx = _clip_inputs(n)
            want = torch.empty(n, 2, 3, 3)
            if n:
                _stub_frame(audio, idx, want)
            calls = []

            def render(a, i, out):
                calls.append(int(a.shape[0]))
                _stub_frame(a, i, out)

            clip, (first, count) = sharded.render_clip_sharded(None, audio, idx, 2, 3, n_chunks=nc, render_fn=render,
                                                               return_local=True)
            ok = ok and clip.shape == (n, 2, 3, 3) and torch.equal(clip, want)
            ok = ok and (first, count) == sharded.shard_range(n, rank, world)[:2] and sum(calls) == count     # only its own frames
            clip8 = sharded.render_clip_sharded(None, audio, idx, 2, 3, n_chunks=nc, gather="u8",
                                                render_fn=lambda a, i, o: (_stub_frame(a, i, o), o.div_(64.0))[0])
            ok = ok and clip8.dtype == torch.uint8 and torch.equal(clip8, to8b((want / 64.0).contiguous()))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_ragged_clip_on_two_and_three_ranks_equals_one_process():
    """N % G != 0 (and N < G): padded blocks through the all-gather, trimmed to N, global frame order; fp32 and uint8;
    one chunk and several."""
    cases = [(10, 1), (7, 1), (7, 3), (11, 2), (2, 1), (1, 1), (0, 1), (12, 4)]
    for world in (2, 3):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_clip_worker, args=(world, _free_port(), cases, ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world)), (world, dict(ret))


def test_render_clip_sharded_without_a_process_group_is_a_plain_render():
    audio, idx = _clip_inputs(5)
    want = torch.empty(5, 2, 3, 3)
    _stub_frame(audio, idx, want)
    got = sharded.render_clip_sharded(None, audio, idx, 2, 3, render_fn=_stub_frame)
    assert torch.equal(got, want)


def _gather_clip_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for n in (7, 9, 2, 0):
            whole = (torch.arange(n * 4, dtype=torch.int32).reshape(n, 2, 2) % 251).to(torch.uint8)
            first, count, _ = sharded.shard_range(n, rank, world)
            got = sharded.gather_clip(whole[first:first + count].clone(), n)
            ok = ok and got.dtype == torch.uint8 and torch.equal(got, whole)
        # the DDP-equivalent gradient averaging of Trainer(multi_gpu=True) (reference: training.py:41)
        from speech2lip_amd.training import Trainer
        tr = Trainer.__new__(Trainer)
        tr.multi_gpu, tr.model = True, torch.nn.Linear(3, 2)
        for p_ in tr.model.parameters():
            p_.grad = torch.full_like(p_, float(rank + 1))
        tr._average_gradients_over_ranks()
        mean = sum(range(1, world + 1)) / world
        ok = ok and all(torch.allclose(p_.grad, torch.full_like(p_, mean)) for p_ in tr.model.parameters())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_gather_clip_and_gradient_averaging_three_ranks():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_clip_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
    assert all(ret[r] for r in range(3))


def _bcast_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import speech2lip_amd as s2l
        torch.manual_seed(100 + rank)                       # ranks that build their replica DIFFERENTLY
        net = s2l.SimpleUnetLight()
        with torch.no_grad():
            net.inc.double_conv[1].running_mean.add_(float(rank))      # ... and whose BatchNorm statistics have drifted apart
            net.inc.double_conv[1].num_batches_tracked.add_(rank)
        before = net.inc.double_conv[0].weight.clone()
        sharded.broadcast_module_state(net, src=0)
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v.clone() for k, v in net.state_dict().items()})
        same = all(torch.equal(gathered[0][k], g[k]) for g in gathered[1:] for k in gathered[0])
        moved = rank == 0 or not torch.equal(before, net.inc.double_conv[0].weight)
        ret[rank] = bool(same and moved and int(net.inc.double_conv[1].num_batches_tracked) == 0)
    finally:
        dist.destroy_process_group()


def test_multi_gpu_trainer_state_starts_as_one_replica():
    """Trainer(multi_gpu=True) stands in for DistributedDataParallel (training.py:41): parameters and buffers (BatchNorm running
    statistics, batch counters) of every rank become rank 0's -- 2 gloo ranks seeded differently end with identical state dicts."""
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_bcast_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
