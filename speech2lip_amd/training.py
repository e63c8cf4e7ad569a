"""Training-time forward of the lip path: the 4-tap local ensemble.

`Trainer.predict_lip_image` keeps the reference's signature
(`src/face_simple/training.py:158`) for the May flag set, so a training loop that calls it
needs no other change; the work happens in `s2l_predict_lip_image` (csrc/ensemble.hip).
Forward only: the backward kernels (BASELINE config 5) are the next row of the build.
"""
from __future__ import annotations

import ctypes

import torch

from . import _abi
from .talking_face import TalkingFace, _dev_f32, _ptr, _stream


def predict_lip_image(model: TalkingFace, coords, audio, index, height: int, width: int, u01: float):
    """coords [HW,2], audio [1,16,29], frame index, the U(0,1) draw of training.py:200 -> [HW,3]."""
    lib = _abi.load()
    packed = model.packed_weights()
    dev = packed.device
    c = _dev_f32(coords, dev, "coords")
    if c.dim() != 2 or c.shape[1] != 2:
        raise ValueError(f"coords must be [N,2], got {tuple(c.shape)}")
    feat = model.audio_merge_forward(audio)               # encoder once, then shared by all pixels (:165/:171)
    if feat.shape[0] != 1:
        raise ValueError("predict_lip_image renders one frame: audio must be [1,16,29]")
    n = c.shape[0]
    idx = int(index.reshape(-1)[0].item()) if isinstance(index, torch.Tensor) else int(index)
    work = torch.empty(max(int(lib.s2l_predict_lip_image_work_floats(n)), 4), dtype=torch.float32, device=dev)
    out = torch.empty(n, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _abi.check(lib.s2l_predict_lip_image(_ptr(packed), _ptr(c), _ptr(feat), idx, int(width), int(height),
                                             ctypes.c_float(float(u01)), _ptr(work), _ptr(out), n, _stream()),
                   "s2l_predict_lip_image")
    return out


class Trainer:
    """The slice of the reference `Trainer` (training.py:21-156) that sits on the hot path."""

    def __init__(self, model, optimizer=None, cfg=None, device=None, **kwargs):
        self.model = model
        self.optimizer = optimizer
        self.cfg = cfg if cfg is not None else model.cfg
        self.device = device if device is not None else model.device
        self.height = int(self.cfg["data"]["height"])
        self.width = int(self.cfg["data"]["width"])
        self.batch_rays = int(self.cfg["training"].get("batch_rays", self.height * self.width))
        self.multi_gpu = False
        self.use_audio = self.use_audio_net = self.use_time = True
        self.use_delta_uv = self.add_noise_audio = False
        self.audio_dims = model.audio_dims

    def predict_lip_image(self, i, coords, audio, pose, data, rgb_zero, lms, seed):
        """Same arguments as the reference method; `pose`, `rgb_zero`, `lms` are unused under the
        May flags exactly as there.  One chunk = the whole lip image (batch_rays = H*W)."""
        chunk = coords[i:i + self.batch_rays, :]
        time_pts = data["index"] if seed is None else data["index"] + seed
        u01 = float(torch.rand(1, device=self.device))          # eps_shift draw (training.py:200)
        return predict_lip_image(self.model, chunk, audio, time_pts, self.height, self.width, u01)[:, :3]
