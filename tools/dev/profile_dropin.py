#!/usr/bin/env python3
"""Host-side profile of the drop-in trainer (cProfile around Trainer.train_step / train_steps on a synthetic folder)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from tools import benchlib

dev = torch.device("cuda:0")
late = "--early" not in sys.argv
K = 8 if "--k8" in sys.argv else 1
FUSED1 = "--fused1" in sys.argv      # one frame per step through train_steps (the StageOneStep route)
FADAM = "--fusedadam" in sys.argv    # speech2lip_amd.FusedAdam instead of torch.optim.Adam
PIPE = "--pipelined" in sys.argv     # train_steps(wait=False): step k's result() after step k + 1 is queued
ONDEV = "--device-frames" in sys.argv   # the frames' floating-point tensors already on the device (what FramePrefetcher(device=) hands over)
root = benchlib._dataset_tmp("may_face_crop_lip")
benchlib.write_synthetic_dataset(root, 24, train=True)
cfg = s2l.may_config(96, 96, data_path=root, train_flags=True)
cfg["model"]["use_canonical_depth"] = False
cfg["training"].update(use_sync_contrastive_loss=True, use_perceptual_loss=False, use_canonical_depth_loss_photo_v2=False, use_syncloss=True)
ds = s2l.SomeonesLipClip(root, "train", cfg=cfg)
net = s2l.SyncNet_color().to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
m = benchlib.make_model(dev, 96, 96, unet=True, train=True)
m.data_path = root
if late:
    for p in m.post_fusion_unet.parameters():
        p.requires_grad = False
    m.post_fusion_unet.eval()
opt = (s2l.FusedAdam if FADAM else torch.optim.Adam)([p for nm, p in m.named_parameters() if p.requires_grad and not nm.startswith("coord_linears")], lr=1e-4)
tr = s2l.Trainer(m, opt, dev, None, cfg=cfg, syncnet=net, precision="bf16", hole_noise="device")
it0 = 100001 if late else 1000
frames = [ds.load_one_frame(i) for i in range(8)]
if ONDEV:
    frames = [{k: (v.to(dev) if isinstance(v, torch.Tensor) and v.is_floating_point() and v.numel() > 16 else v) for k, v in f.items()} for f in frames]
batches = [s2l.data.collate_batch([f]) for f in frames]
pending = [None]


def step(k):
    if FUSED1:
        tr.train_steps([frames[k % 8]], it=it0 + k)
    elif K == 1:
        tr.train_step(batches[k % 8], it=it0 + k)
    elif PIPE:
        h = tr.train_steps(frames, it=it0 + k, wait=False)
        if pending[0] is not None:
            pending[0].result()
        pending[0] = h
    else:
        tr.train_steps(frames, it=it0 + k)


for k in range(4):
    step(k)
torch.cuda.synchronize()
if "--syncs" in sys.argv:      # where does the host wait for the device inside one step?
    import warnings
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    step(4)
    torch.cuda.set_sync_debug_mode("default")
    sys.exit(0)
t0 = time.perf_counter()
for k in range(10):
    step(k)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
print(f"K={K} late={late}: host {t_host / 10 * 1e3:.2f} ms per step, wall {t_wall / 10 * 1e3:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for k in range(10):
    step(k)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
st.print_callers("_dev_f32")
st.print_callers("method 'to' of")
st.print_callers("named_parameters")
st.print_callers("_named_members")
st.print_callers("module.py.*parameters")
