#!/usr/bin/env python3
"""Clip-level replacement of the reference's inference.py loop (inference.py:78-178) on the MI355X path: same YAML config
surface, same checkpoint keys, same dataset folder layout, same output naming; one render + composite + U-Net call per
batch of frames instead of >100 launches per frame.

    python tools/infer_clip.py --config configs/face_simple_configs/may/may.yaml --default configs/default.yaml \
        --checkpoint out/may/model.pt [--mode val|test] [--batch 100] [--out DIR]

Multi-GPU (BASELINE config 4): launch it under torchrun and the frames are sharded across the ranks in contiguous blocks of
ceil(N / G) (speech2lip_amd.sharded.shard_range); every rank renders + composites + runs the U-Net on its own
block and writes its own files -- the data path needs no collective.  With --gather the 8-bit clip is also reassembled on every
rank by one RCCL all-gather (sharded.gather_clip; N need not divide by G) and rank 0 writes all files; with --lip-only the lip
frames alone are rendered through sharded.render_clip_sharded (fp32, bit-identical to one GPU).

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/infer_clip.py --config ... [--gather]

Not a CLI product: a short example of the drop-in calls."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import sharded


def _render_batches(args, s2l, model, batches, writer, blocks, post, rank, world, n, out_dir):
    for clip in batches:
        lip, recon, merged = s2l.render_clip_frames(model, clip, use_post_fusion=post, precision="split" if args.fast else "fp32")
        frames = recon if recon is not None else (merged if merged is not None else lip)
        if args.gather and world > 1:
            blocks.append(s2l.to8b(frames))
        elif writer is not None:
            writer.submit(s2l.to8b(frames), clip.names)
        else:
            s2l.write_frames(frames, clip.names, out_dir)
        print(f"[rank {rank}] frames {clip.names[0]}..{clip.names[-1]} of {n} -> {out_dir}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--default", required=True)
    ap.add_argument("--checkpoint", help="reference checkpoint (dict with key 'model'); random init when absent")
    ap.add_argument("--mode", default="val", choices=["val", "test"])        # test = --use_new_audio (audio_test/audio.npy)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--out")
    ap.add_argument("--gather", action="store_true", help="multi-GPU: all-gather the 8-bit clip, rank 0 writes every file")
    ap.add_argument("--fast", action="store_true", help="the opt-in split speed modes of the lip renderer and the U-Net (fp32-grade output)")
    ap.add_argument("--lip-only", action="store_true", help="render only the lip crops (sharded.render_clip_sharded)")
    ap.add_argument("--workers", type=int, default=None, help="host threads that decode the inputs and encode the outputs (default: min(32, cores))")
    ap.add_argument("--serial", action="store_true", help="the un-pipelined loop (load -> render -> write, one after the other): the baseline "
                    "bench.py's extra.infer_clip_end_to_end compares the pipeline with")
    args = ap.parse_args()
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL on this driver)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    cfg = s2l.load_config(args.config, args.default)
    ds = s2l.SomeonesLipClip(cfg["data"]["path"], args.mode, cfg)
    cfg["data"]["height"], cfg["data"]["width"] = ds.lip_h, ds.lip_w
    model = s2l.TalkingFace(dev, cfg, mode="eval").eval()
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location="cpu")["model"], strict=False)
    out_dir = args.out or os.path.join(cfg["training"]["out_dir"], "test_post" if args.mode == "val" else "test_new_audio")
    n = len(ds)
    names = ["%05d" % (i + 1) for i in range(n)]                              # inference.py:177
    if args.lip_only:
        audio = torch.from_numpy(ds.aud_features.astype("float32"))
        lips = sharded.render_clip_sharded(model, audio, torch.arange(n), ds.lip_h, ds.lip_w, gather="u8",
                                           precision="split" if args.fast else "fp32")
        if rank == 0:
            s2l.write_frames(lips, names, out_dir)
    else:
        first0, count, _ = sharded.shard_range(n, rank, world)               # this rank's contiguous block of the clip
        post = bool(cfg["model"].get("use_post_fusion", True))
        blocks, writer, streamer = [], None, None
        if args.serial:
            batches = (ds.load(dev, first, min(args.batch, first0 + count - first)) for first in range(first0, first0 + count, args.batch))
        else:     # decode + H2D of batch k+1 and encode + write of batch k-1 run beside the GPU work of batch k
            batches = streamer = s2l.ClipStreamer(ds, dev, args.batch, first0, count, workers=args.workers)
            writer = s2l.FrameWriter(out_dir, workers=args.workers)
        import contextlib
        with contextlib.ExitStack() as owned:      # the streamer's shared-memory blocks / the writer's threads go whatever happens below
            for res in (streamer, writer):
                if res is not None:
                    owned.enter_context(res)
            _render_batches(args, s2l, model, batches, writer, blocks, post, rank, world, n, out_dir)
        if args.gather and world > 1:
            # an empty shard still takes part in the collective: its block has the frame size the OTHER ranks gather (the face frame
            # when the composite / U-Net ran, the lip crop otherwise)
            has_face = post and ds.coord_files is not None and ds.mode != "test"
            fh, fw = (ds.face_h, ds.face_w) if has_face else (ds.lip_h, ds.lip_w)
            mine = torch.cat(blocks, 0) if blocks else torch.empty((0, fh, fw, 3), dtype=torch.uint8, device=dev)
            whole = sharded.gather_clip(mine, n)
            if rank == 0:
                s2l.write_frames(whole, names, out_dir)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
