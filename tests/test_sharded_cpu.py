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
