"""Wire-format reader for a Speech2Lip dataset folder and the clip-level driver
that replaces the reference's per-frame inference loop on it.

On-disk layout and conventions follow `src/data/someones_lip_dataset.py` of the reference:

    audio/audio.npy            float64 [N,16,29] DeepSpeech windows (deepspeech_features.py:65-75)
    audio_test/audio.npy       same, for mode 'test' (--use_new_audio)
    coords/%05d.npy            float32 [FH,FW,2] warp grid per frame, in [-1,1]   (face_tracker.py:297-303)
    ori_images_face/%05d.jpg   observed face frames (rgb_face_ori); frame canonical_idx+1 is the canonical face
    images/%05d.jpg            lip crops; their size is the lip box size
    canonical_lip_mask.jpg     soft lip mask in canonical space, read with cv2 in the reference (BGR order)
    landmarks/%05d.lms         68 x 2(+) landmark text file; points 48.. are the mouth

`SomeonesLipClip` does what `SomeonesLipDataset.__init__` + `load_one_frame` do for the inference
modes ('val' / 'test'), but once per CLIP: every per-frame array is stacked and moved to the GPU in
one go (pinned staging, non-blocking copies), because the renderer consumes whole clips.  JPEG
decoding uses PIL (the reference uses imageio -- PIL underneath -- and cv2; decoders agree to
within 1/255 on the same libjpeg family, which is why parity of this row is pinned on synthetic
folders written by the tests rather than on golden pixels).  Training-only fields (mel windows,
SyncNet crops, head masks) are not read.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch


def _read_rgb(path: str) -> np.ndarray:
    """uint8 RGB image -> float32 [H,W,3] in [0,1]  (get_color, someones_lip_dataset.py:196-217)."""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"))
    return (arr / 255.0).astype(np.float32)


def _read_bgr01(path: str) -> np.ndarray:
    """cv2.imread(path) / 255 (someones_lip_dataset.py:72): channels in BGR order, float64 -> float32."""
    return _read_rgb(path)[:, :, ::-1].copy()


def list_dir(folder: str, ext: str):
    """sorted file names containing `ext` (someones_lip_dataset.py:166-171)."""
    return sorted(f for f in os.listdir(folder) if ext in f)


def bounding_rect(points: np.ndarray):
    """cv2.boundingRect of a float point set: x = floor(min x), y = floor(min y),
    w = floor(max x) - x + 1, h = floor(max y) - y + 1."""
    xs, ys = points[:, 0], points[:, 1]
    x, y = int(np.floor(xs.min())), int(np.floor(ys.min()))
    return x, y, int(np.floor(xs.max())) - x + 1, int(np.floor(ys.max())) - y + 1


def compute_mouth_bbox(lms: np.ndarray, lip_w: int, lip_h: int, dataset_folder: str, center_point_y_ratio: float = 1.0):
    """Fixed-size lip box centred on the mouth landmarks (someones_lip_dataset.py:173-193): the y centre
    is scaled by 1.02 (1.0 for 'adnerf' folders, cfg.data.center_point_y_ratio for 'macron')."""
    x, y, w, h = bounding_rect(np.asarray(lms)[48:, :2])
    cx = x + w / 2.0
    if "adnerf" in dataset_folder:
        cy = y + h / 2.0
    elif "macron" in dataset_folder:
        cy = (y + h / 2.0) * center_point_y_ratio
    else:
        cy = (y + h / 2.0) * 1.02
    return int(cx - int(lip_w) / 2.0), int(cy - int(lip_h) / 2.0), int(lip_w), int(lip_h)


def split_slice(n_audio: int, mode: str, dataset_folder: str) -> slice:
    """The slice the reference applies to the file list, the audio windows and the pose grids alike
    (someones_lip_dataset.py:122-155): train = [:length] with length = 90 % of the audio windows (all
    of them for 'lip_train' folders); val = [length:] where `length` is overridden to -598 / -650 /
    -800 for the May / obama2_face_crop / obama_adnerf folders."""
    length = n_audio if "lip_train" in dataset_folder else int(n_audio * 0.9)
    if mode == "train":
        return slice(None, length)
    if mode == "val":
        if "may" in dataset_folder:
            length = -598
        elif "obama2_face_crop" in dataset_folder:
            length = -650
        elif "obama_adnerf" in dataset_folder:
            length = -800
        return slice(length, None)
    raise ValueError(f"unknown split {mode!r}")


@dataclass
class ClipTensors:
    """One clip, resident on `device` (what `inference.py:140-172` pulls frame by frame from its DataLoader)."""
    audio: torch.Tensor                 # [F,16,29] fp32
    index: torch.Tensor                 # [F] int64: data['index'], relative to the split
    coord: Optional[torch.Tensor]       # [F,FH,FW,2]
    rgb_face_ori: Optional[torch.Tensor]    # [F,FH,FW,3]
    rgb_face_zero: torch.Tensor         # [1,FH,FW,3]
    mask_lip_canonical: torch.Tensor    # [1,FH,FW,3]
    lip_lefttop_x: int
    lip_lefttop_y: int
    height: int
    width: int
    names: list                         # output file stems ("%05d" of index+1, inference.py:177)


class SomeonesLipClip:
    def __init__(self, dataset_folder: str, mode: str, cfg=None, img_ext: str = ".jpg"):
        if mode not in ("val", "test", "train"):
            raise ValueError(f"unknown mode {mode!r}")
        self.dataset_folder, self.mode, self.cfg, self.img_ext = dataset_folder, mode, cfg, img_ext
        self.canonical_idx = 12 if "obama2" in dataset_folder else 0                      # :37-41
        j = lambda *p: os.path.join(dataset_folder, *p)
        canon = "{:05d}.jpg".format(self.canonical_idx + 1)
        self.rgb_face_zero = _read_rgb(j("ori_images_face", canon))                        # :57-59
        self.face_h, self.face_w = self.rgb_face_zero.shape[:2]
        rgb_zero = _read_rgb(j("images", canon))                                           # :69
        self.lip_h, self.lip_w = rgb_zero.shape[:2]
        self.mask_lip_canonical = _read_bgr01(j("canonical_lip_mask.jpg"))                 # :72
        lms = np.loadtxt(j("landmarks", "{:05d}.lms".format(self.canonical_idx + 1)), dtype=np.float32)
        ratio = float((cfg or {}).get("data", {}).get("center_point_y_ratio", 1.0)) if cfg else 1.0
        self.lefttop_x, self.lefttop_y, _, _ = compute_mouth_bbox(lms, self.lip_w, self.lip_h, dataset_folder, ratio)
        self.image_files = list_dir(j("images"), img_ext)
        self.coord_files = list_dir(j("coords"), ".npy") if os.path.isdir(j("coords")) else None
        aud = np.load(j("audio", "audio.npy"))                                             # :104
        if mode == "test":                                                                 # :156-161
            self.aud_features = np.load(j("audio_test", "audio.npy"))
        else:
            sl = split_slice(aud.shape[0], mode, dataset_folder)
            self.aud_features = aud[sl]
            self.image_files = self.image_files[sl]
            if self.coord_files is not None:
                self.coord_files = self.coord_files[sl]

    def __len__(self):
        return int(self.aud_features.shape[0]) if self.mode == "test" else len(self.image_files)

    def load(self, device, first: int = 0, count: Optional[int] = None) -> ClipTensors:
        """Frames [first, first+count) of the split as device tensors (one H2D batch)."""
        n = len(self)
        count = n - first if count is None else min(count, n - first)
        idx = np.arange(first, first + count)
        dev = torch.device(device)

        def up(arr):
            t = torch.from_numpy(np.ascontiguousarray(arr))
            if dev.type == "cuda":
                t = t.pin_memory()
            return t.to(dev, non_blocking=True)

        audio = up(self.aud_features[idx].astype(np.float32))      # torch.Tensor(float64 array) casts the same way (:246)
        coord = None
        if self.coord_files is not None and self.mode != "test":
            coord = up(np.stack([np.load(os.path.join(self.dataset_folder, "coords", self.coord_files[i])).astype(np.float32)
                                 for i in idx]))
        ori = None
        if self.mode != "test":
            ori = up(np.stack([_read_rgb(os.path.join(self.dataset_folder, "ori_images_face", self.image_files[i]))
                               for i in idx]))
        return ClipTensors(audio=audio, index=torch.from_numpy(idx.astype(np.int64)).to(dev), coord=coord, rgb_face_ori=ori,
                           rgb_face_zero=up(self.rgb_face_zero[None]), mask_lip_canonical=up(self.mask_lip_canonical[None]),
                           lip_lefttop_x=self.lefttop_x, lip_lefttop_y=self.lefttop_y, height=self.lip_h, width=self.lip_w,
                           names=["{:05d}".format(int(i) + 1) for i in idx])


def render_clip_frames(model, clip: ClipTensors, use_post_fusion: bool = True):
    """The body of `inference.py:140-172` for a whole clip: lip frames [F,h,w,3] and, when the clip has
    pose grids and observed frames, (rgb_face_recon, rgb_merged_new) [F,FH,FW,3]."""
    lip = model.render_clip(clip.audio, clip.index, clip.height, clip.width)
    if not (use_post_fusion and clip.coord is not None and clip.rgb_face_ori is not None):
        return lip, None, None
    new, _ = model.composite_clip(lip, clip.rgb_face_zero, clip.rgb_face_ori, clip.mask_lip_canonical, clip.lip_lefttop_x,
                                  clip.lip_lefttop_y, clip.coord)
    recon = None
    if getattr(model, "post_fusion_unet", None) is not None and not model.training:
        recon = model.post_fusion_unet.forward_nhwc(new)
    return lip, recon, new


def to8b(frames: torch.Tensor) -> torch.Tensor:
    """float frames -> uint8 with cv2.imwrite's conversion (round to nearest even, saturate): on the GPU through
    s2l_to8b (so that only a quarter of the bytes cross PCIe), with torch ops for host tensors."""
    if frames.device.type != "cuda":
        return (frames.detach().to(torch.float32) * 255.0).round().clamp(0, 255).to(torch.uint8)
    import ctypes
    from . import _abi
    x = frames.detach().to(torch.float32).contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _abi.check(_abi.load().s2l_to8b(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), x.numel(),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_to8b")
    return out


def write_frames(frames: torch.Tensor, names, out_dir: str, ext: str = ".jpg") -> None:
    """rgb*255 -> 8-bit image files named %05d (inference.py:172-178; the reference converts to BGR only
    because cv2.imwrite expects it: the files hold the same RGB picture).  cv2.imwrite rounds and saturates."""
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    arr = to8b(frames).cpu().numpy()
    for a, name in zip(arr, names):
        Image.fromarray(a, "RGB").save(os.path.join(out_dir, name + ext), quality=95)
