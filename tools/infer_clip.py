#!/usr/bin/env python3
"""Clip-level replacement of the reference's inference.py loop (inference.py:78-178) on the MI355X path: same YAML config
surface, same checkpoint keys, same dataset folder layout, same output naming; one render + composite + U-Net call per
batch of frames instead of >100 launches per frame.

    python tools/infer_clip.py --config configs/face_simple_configs/may/may.yaml --default configs/default.yaml \
        --checkpoint out/may/model.pt [--mode val|test] [--batch 100] [--out DIR]

Not a CLI product: a 40-line example of the drop-in calls."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--default", required=True)
    ap.add_argument("--checkpoint", help="reference checkpoint (dict with key 'model'); random init when absent")
    ap.add_argument("--mode", default="val", choices=["val", "test"])        # test = --use_new_audio (audio_test/audio.npy)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--out")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = s2l.load_config(args.config, args.default)
    ds = s2l.SomeonesLipClip(cfg["data"]["path"], args.mode, cfg)
    cfg["data"]["height"], cfg["data"]["width"] = ds.lip_h, ds.lip_w
    model = s2l.TalkingFace(dev, cfg, mode="eval").eval()
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location="cpu")["model"], strict=False)
    out_dir = args.out or os.path.join(cfg["training"]["out_dir"], "test_post" if args.mode == "val" else "test_new_audio")
    n = len(ds)
    for first in range(0, n, args.batch):
        clip = ds.load(dev, first, args.batch)
        lip, recon, merged = s2l.render_clip_frames(model, clip, use_post_fusion=bool(cfg["model"].get("use_post_fusion", True)))
        frames = recon if recon is not None else (merged if merged is not None else lip)
        s2l.write_frames(frames, clip.names, out_dir)
        print(f"frames {first + 1}..{first + len(clip.names)} of {n} -> {out_dir}", flush=True)


if __name__ == "__main__":
    main()
