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
folders written by the tests rather than on golden pixels).

`SomeonesLipClip.load_one_frame(index)` is the per-frame mirror of the reference's `load_one_frame`
(:242-399): the same dictionary, key for key, including the training fields the sync loss consumes
(`mel`, `coord_window`, `audio_window`, `canonical_face_bbox`, `rgb_window_neg`, `total_frame`) and the
6-DoF pose / canonical masks of the depth loss.  It is checked field by field against what the
REFERENCE's own reader yields for the committed fixture folder (golden G15, tools/make_goldens.py;
pinned except JPEG decoding, cv2.resize's interpolation (restated from OpenCV's published code) and cv2.boundingRect's rounding -- those three
libraries are absent from the image).  The mel front-end (src/data/audio.py, librosa) is out of scope:
the spectrogram is read precomputed from `audio/mel.npy` ([80, T_mel], what `melspectrogram` returns).
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


def _resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h)) for uint8 images with the default INTER_LINEAR, restating the published algorithm of
    OpenCV 4.4 (`opencv-python==4.4.0.40`, requirement.txt:17; modules/imgproc/src/resize.cpp), pass by pass:
      * scale = 1 / (dst / src) in double; source position fx = float((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx;
        horizontally a tap left of the image gives (sx, fx) = (0, 0) and a tap pair that would leave it on the right gives
        (width - 1, 0) [resizeGeneric set-up]; vertically the two ROW indices are clamped instead and the fraction is kept;
      * coefficients are shorts: round-half-even(float(1 - f) * 2048) and round-half-even(f * 2048) (saturate_cast<short>);
      * horizontal pass (HResizeLinear, uchar -> int): D = S[sx] * a0 + S[sx + 1] * a1   (scale 2^11, exact);
      * vertical pass (VResizeLinear<uchar, int, short, FixedPtCast<.., 22>>): the TRUNCATING form
            dst = ( ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 ) >> 2
        -- not one rounding of the 22-bit product: the shifts drop bits between the passes, which moves the result by one
        level on some pixels;
      * an exact 2x reduction of both axes is switched to INTER_AREA by cv::resize (`is_area_fast && iscale == 2`):
        (a + b + c + d + 2) >> 2 over the 2x2 block.
    cv2 itself is not in the image, so this stays a restatement of published code; tests/test_data_reader.py pins it with a
    scalar re-derivation and hand-computed pixels on a non-flat image."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    if h == 2 * out_h and w == 2 * out_w:
        a = img.astype(np.int32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def positions(n_in, n_out):
        scale = 1.0 / (float(n_out) / float(n_in))
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        return i0, (f - i0.astype(np.float32)).astype(np.float32)

    def coefs(f):                       # saturate_cast<short>(float * 2048): cvRound = round half to even
        c0 = np.rint((np.float32(1.0) - f).astype(np.float32) * np.float32(2048.0)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
        return c0, c1

    sx, fx = positions(w, out_w)
    fx = np.where(sx < 0, np.float32(0), fx)
    sx = np.where(sx < 0, 0, sx)
    fx = np.where(sx >= w - 1, np.float32(0), fx).astype(np.float32)
    sx = np.where(sx >= w - 1, w - 1, sx)
    a0, a1 = coefs(fx)
    sx1 = np.minimum(sx + 1, w - 1)                                                    # its coefficient is 0 there
    sy, fy = positions(h, out_h)
    b0, b1 = coefs(fy)
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    a = img.astype(np.int64)
    cshape = (1, -1) + (1,) * (a.ndim - 2)
    rows = a[:, sx] * a0.reshape(cshape) + a[:, sx1] * a1.reshape(cshape)                # [h, out_w(, C)] * 2^11
    rshape = (-1, 1) + (1,) * (a.ndim - 2)
    out = (((b0.reshape(rshape) * (rows[y0] >> 4)) >> 16) + ((b1.reshape(rshape) * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_audio_window(spec: np.ndarray, start_frame: int, fps: int = 25, syncnet_mel_step_size: int = 16) -> np.ndarray:
    """someones_lip_dataset.py:401-414: 16 mel frames from int(80 * start_frame / fps), or the last 16."""
    start = int(80.0 * (start_frame / float(fps)))
    end = start + syncnet_mel_step_size
    if end > spec.shape[0]:
        start, end = spec.shape[0] - 16, spec.shape[0]
    return spec[start:end, :]


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
        tc, mc = ((cfg or {}).get("training", {}), (cfg or {}).get("model", {}))
        self.use_syncloss = bool(tc.get("use_syncloss", False))                            # :34
        # The reference reads the sync-loss side inputs of EVERY training frame once use_syncloss is configured (:328-385: five pose grids,
        # five more JPEG decodes + resizes, the mel crop -- most of a frame's ~10 ms and 11 of its 16 MB), although train_stage1 only looks
        # at them after it > 100000 (training.py:491).  False skips them (a loader for the early phase: FramePrefetcher(sync_fields=False)).
        self.load_sync_fields = True
        self.use_sync_contrastive_loss = bool(tc.get("use_sync_contrastive_loss", False))
        self.use_canonical_depth = bool(mc.get("use_canonical_depth", False))
        self.use_post_fusion = bool(mc.get("use_post_fusion", True))
        self.fmin = 95 if "may" in dataset_folder else 55                                  # :104-109 (mel front-end parameter)
        self.canonical_idx = 12 if "obama2" in dataset_folder else 0                      # :37-41
        j = lambda *p: os.path.join(dataset_folder, *p)
        canon = "{:05d}.jpg".format(self.canonical_idx + 1)
        self.rgb_face_zero = _read_rgb(j("ori_images_face", canon))                        # :57-59
        self.face_h, self.face_w = self.rgb_face_zero.shape[:2]
        rgb_zero = _read_rgb(j("images", canon))                                           # :69
        self.lip_h, self.lip_w = rgb_zero.shape[:2]
        self.rgb_zero = rgb_zero
        self.mask_lip_canonical = _read_bgr01(j("canonical_lip_mask.jpg"))                 # :72
        if self.use_canonical_depth:                                                       # :75-93
            if os.path.exists(j("track_params.pt")):
                params = torch.load(j("track_params.pt"))
                self.pose_features_euler, self.pose_features_trans = params["euler"], params["trans"]
                self.canonical_euler = self.pose_features_euler[self.canonical_idx]
                self.canonical_trans = self.pose_features_trans[self.canonical_idx]
            self.mask_head_canonical = _read_bgr01(j("canonical_head_mask.jpg"))[:, :, :1].copy()     # channel 0 of BGR
            self.mask_face_canonical = _read_bgr01(j("canonical_face_mask.jpg"))
        lms = np.loadtxt(j("landmarks", "{:05d}.lms".format(self.canonical_idx + 1)), dtype=np.float32)
        ratio = float((cfg or {}).get("data", {}).get("center_point_y_ratio", 1.0)) if cfg else 1.0
        self.lefttop_x, self.lefttop_y, _, _ = compute_mouth_bbox(lms, self.lip_w, self.lip_h, dataset_folder, ratio)
        self.image_files = list_dir(j("images"), img_ext)
        self.coord_files = list_dir(j("coords"), ".npy") if os.path.isdir(j("coords")) else None
        aud = np.load(j("audio", "audio.npy"))                                             # :102
        if self.use_syncloss and mode == "train":                                          # :113-120
            # orig_mel = melspectrogram(load_wav(audio/audio.wav), fmin).T -- the mel front-end is out of scope:
            # the spectrogram is taken precomputed, in melspectrogram's own [80, T_mel] orientation
            if not os.path.exists(j("audio", "mel.npy")):
                raise FileNotFoundError(
                    f"{j('audio', 'mel.npy')} not found: with training.use_syncloss the reader needs the mel spectrogram of "
                    f"audio/audio.wav precomputed as a float [80, T_mel] array -- what the reference computes on the fly with "
                    f"src/data/audio.py melspectrogram(load_wav(audio.wav, 16000), fmin={self.fmin}) (someones_lip_dataset.py:113-118; "
                    f"librosa is not part of this path).  Save that array with np.save.")
            self.orig_mel = np.load(j("audio", "mel.npy")).T
            self.face_bbox_dict = np.load(j("face_bbox_dict.npy"), allow_pickle=True).item()
        if mode == "test":                                                                 # :156-161
            self.aud_features = np.load(j("audio_test", "audio.npy"))
        else:
            sl = split_slice(aud.shape[0], mode, dataset_folder)
            self.aud_features = aud[sl]
            self.image_files = self.image_files[sl]
            if mode == "train":
                # :131 runs BEFORE the pose grids are sliced (:134-135) and while the 6-DoF pose still spans the whole clip
                self.data_zero = self.load_one_frame(self.canonical_idx)
                self.data_zero["rgb"] = self.data_zero["rgb"].unsqueeze(0)
                self.data_zero["audio"] = self.data_zero["audio"].unsqueeze(0)
            if self.coord_files is not None:
                self.coord_files = self.coord_files[sl]
            if self.use_canonical_depth and hasattr(self, "pose_features_euler"):          # :136-138, :153-155
                self.pose_features_euler = self.pose_features_euler[sl]
                self.pose_features_trans = self.pose_features_trans[sl]

    def __len__(self):
        return int(self.aud_features.shape[0]) if self.mode == "test" else len(self.image_files)

    def load_one_frame(self, index: int) -> dict:
        """The dictionary `SomeonesLipDataset.load_one_frame(index)` returns (someones_lip_dataset.py:242-399), key for key and
        type for type (host tensors / ints; the DataLoader's collate adds the batch axis).  Train mode with use_syncloss adds
        the sync-loss inputs (:328-385)."""
        j = lambda *p: os.path.join(self.dataset_folder, *p)
        n = len(self)
        inputs = {"audio": torch.from_numpy(np.asarray(self.aud_features[index]).astype(np.float32)),     # torch.Tensor(float64) casts
                  "index": index, "total_frame": n}
        if self.coord_files is not None:                                                   # :251-262
            inputs["coord"] = torch.from_numpy(np.load(j("coords", self.coord_files[index])).astype(np.float32))
        inputs["rgb_face_zero"] = torch.from_numpy(self.rgb_face_zero)
        inputs["mask_lip_canonical"] = torch.from_numpy(self.mask_lip_canonical)
        inputs["lip_lefttop_x"], inputs["lip_lefttop_y"] = self.lefttop_x, self.lefttop_y
        if self.use_post_fusion or self.mode in ("val", "test"):                           # :272-275
            inputs["rgb_face_ori"] = torch.from_numpy(_read_rgb(j("ori_images_face", self.image_files[index])))
        if self.use_canonical_depth:                                                       # :295-297
            inputs["mask_head_3DMM_canonical"] = torch.from_numpy(self.mask_head_canonical)
            inputs["mask_face_3DMM_canonical"] = torch.from_numpy(self.mask_face_canonical)

        def pose():                                                                        # :304-309 / :387-392
            if self.use_canonical_depth:
                inputs["canonical_euler"], inputs["canonical_trans"] = self.canonical_euler, self.canonical_trans
                inputs["euler"], inputs["trans"] = self.pose_features_euler[index], self.pose_features_trans[index]
        if self.mode == "test":                                                            # :299-314
            inputs["rgb_zero"] = torch.from_numpy(self.rgb_zero)
            pose()
            return inputs
        rgb = _read_rgb(j("images", self.image_files[index]))                              # :316-326
        inputs["rgb"] = torch.from_numpy(rgb)
        inputs["rgb_zero"] = torch.from_numpy(self.rgb_zero)
        inputs["height"], inputs["width"] = rgb.shape[0], rgb.shape[1]
        inputs["face_h"], inputs["face_w"] = self.face_h, self.face_w
        if self.use_syncloss and self.mode == "train" and self.load_sync_fields:           # :328-385
            mel = crop_audio_window(self.orig_mel.copy(), index + 2)
            inputs["mel"] = torch.from_numpy(np.ascontiguousarray(mel.T).astype(np.float32)).unsqueeze(0)      # [1,80,16]
            # five consecutive frames; past the end of the split the last one that existed is repeated (:333-362)
            last = lambda k, m: min(index + k, m - 1)
            inputs["coord_window"] = torch.from_numpy(np.stack(
                [np.load(j("coords", self.coord_files[last(k, len(self.coord_files))])).astype(np.float32) for k in range(5)]))
            inputs["audio_window"] = torch.from_numpy(np.stack(
                [np.asarray(self.aud_features[last(k, len(self.aud_features))]) for k in range(5)]).astype(np.float32))
            inputs["canonical_face_bbox"] = self.face_bbox_dict["{:05d}.jpg".format(self.canonical_idx + 1)]   # :363
            if self.use_sync_contrastive_loss:                                             # :365-385
                # a window 5 frames later, or 10 frames earlier near the end; `index - 10` can be negative, which python's
                # list indexing wraps to the end of the split -- reproduced, because that is what the reference trains on
                start = index + 5 if index + 5 + 5 < len(self.image_files) else index - 10
                frames = []
                for k in range(5):
                    from PIL import Image
                    with Image.open(j("ori_images_face", self.image_files[start + k])) as im:
                        u8 = np.asarray(im.convert("RGB"))
                    frames.append((_resize_bilinear_u8(u8, 96, 96) / 255.0).astype(np.float32))
                inputs["rgb_window_neg"] = torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(frames), (3, 0, 1, 2))))
        pose()
        return inputs

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


def collate_batch(items):
    """torch's default_collate for a list of `load_one_frame` dictionaries (what the reference's DataLoader hands to
    Trainer.train_step, someones_lip_dataset.py:422-431): tensors and numpy arrays are stacked, ints become int64 tensors."""
    out = {}
    for k in items[0]:
        vals = [it[k] for it in items]
        if isinstance(vals[0], torch.Tensor):
            out[k] = torch.stack(vals, 0)
        elif isinstance(vals[0], np.ndarray):
            out[k] = torch.stack([torch.from_numpy(np.ascontiguousarray(v)) for v in vals], 0)
        else:
            out[k] = torch.tensor(vals)
    return out


def render_clip_frames(model, clip: ClipTensors, use_post_fusion: bool = True, precision: str = "fp32"):
    """The body of `inference.py:140-172` for a whole clip: lip frames [F,h,w,3] and, when the clip has
    pose grids and observed frames, (rgb_face_recon, rgb_merged_new) [F,FH,FW,3].
    precision: "fp32" (default) the exact kernels; "split" the opt-in speed modes of the lip renderer (hi + lo halves) and of
    the U-Net's convolutions (hi + lo bf16): fp32-grade output (>= 99 dB against the exact path) at a multiple of its rate."""
    if precision not in ("fp32", "split"):
        raise ValueError("precision must be 'fp32' or 'split'")
    lip = model.render_clip(clip.audio, clip.index, clip.height, clip.width, precision=precision)
    if not (use_post_fusion and clip.coord is not None and clip.rgb_face_ori is not None):
        return lip, None, None
    new, _ = model.composite_clip(lip, clip.rgb_face_zero, clip.rgb_face_ori, clip.mask_lip_canonical, clip.lip_lefttop_x,
                                  clip.lip_lefttop_y, clip.coord)
    recon = None
    if getattr(model, "post_fusion_unet", None) is not None and not model.training:
        recon = model.post_fusion_unet.forward_nhwc(new, precision=precision)
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


def from8b(u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 device tensor -> float32, exactly the reader's `(array / 255.0).astype(float32)` (s2l_from8b): decoded frames cross
    PCIe as bytes and become on the device the floats `_read_rgb` yields on the host, bit for bit."""
    import ctypes
    from . import _abi
    if u8.device.type != "cuda" or u8.dtype != torch.uint8:
        raise _abi.S2LError("from8b: a uint8 tensor on the GPU is required (no CPU fallback)")
    x = u8.contiguous()
    o = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _abi.check(_abi.load().s2l_from8b(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(o.data_ptr()), x.numel(),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_from8b")
    return o


_DECODE_WORKERS = []      # child processes of ClipStreamer(mode="process"): started once per process, shared by every streamer
_DECODE_FREE = None       # queue of idle workers


def _spawn_decode_worker():
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_io_worker.py")
    return subprocess.Popen([sys.executable, "-u", script], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)


def _decode_workers(n: int):
    """At least `n` `_io_worker.py` children (started together on first use: each takes ~1 s to import numpy + PIL on a cold box;
    they stay for the life of this process -- daemons that end when their stdin closes) and the queue a task takes an idle one from."""
    import atexit
    import queue
    global _DECODE_FREE
    if _DECODE_FREE is None:
        _DECODE_FREE = queue.Queue()

        def _stop():
            for w in _DECODE_WORKERS:
                try:
                    w.stdin.close()
                except Exception:
                    pass
            for w in _DECODE_WORKERS:
                try:
                    w.wait(timeout=2)
                except Exception:
                    w.kill()
        atexit.register(_stop)
    while len(_DECODE_WORKERS) < n:
        w = _spawn_decode_worker()
        _DECODE_WORKERS.append(w)
        _DECODE_FREE.put(w)
    return _DECODE_FREE


def shutdown_decode_workers() -> None:
    """Ends the process-wide decode workers (they otherwise live until interpreter exit, shared by every `ClipStreamer`): a long-lived
    service that has finished rendering calls this to give the processes back.  Streamers created afterwards start new ones; streamers
    still open must be closed first."""
    import queue
    global _DECODE_FREE
    _drop_cached_blocks()                                         # (the blocks kept for the next streamer go with the workers that map them)
    workers, _DECODE_WORKERS[:] = list(_DECODE_WORKERS), []
    if _DECODE_FREE is not None:
        try:
            while True:
                _DECODE_FREE.get_nowait()
        except queue.Empty:
            pass
    for w in workers:
        try:
            w.stdin.close()
        except Exception:
            pass
    for w in workers:
        try:
            w.wait(timeout=5)
        except Exception:
            w.kill()
            w.wait()
        try:
            w.stdout.close()
        except Exception:
            pass


def _replace_decode_worker(dead):
    """A worker that stopped answering (killed, crashed in a decoder) is reaped and a fresh one takes its place in the pool."""
    try:
        dead.kill()
        dead.wait(timeout=2)
    except Exception:
        pass
    w = _spawn_decode_worker()
    try:
        _DECODE_WORKERS[_DECODE_WORKERS.index(dead)] = w
    except ValueError:
        _DECODE_WORKERS.append(w)
    return w


def _ask_decode_worker(free, line: str) -> str:
    """One request line to an idle worker, its one answer line back ("" = the worker died: it is replaced before the queue sees it)."""
    w = free.get()
    try:
        try:
            w.stdin.write(line + "\n")
            w.stdin.flush()
            ans = w.stdout.readline().strip()
        except (BrokenPipeError, OSError, ValueError):
            ans = ""
        if not ans:
            w = _replace_decode_worker(w)
    finally:
        free.put(w)
    return ans


_SHM_CACHE = {}                 # size -> [SharedMemory]: blocks of closed streamers, kept for the next one (bounded)
_SHM_CACHE_MAX = 1536 << 20     # bytes kept at most: three slots of a 100-frame 500 x 500 streamer are 0.8 GB


def _take_block(size: int):
    """A shared-memory block of `size` bytes: one a closed streamer gave back (already mapped in the decode workers, its pages already
    faulted in -- creating 0.4 - 0.8 GB afresh per clip cost ~8 % of a 600-frame clip's time in zero-fills, mmaps and munmaps), or a new one."""
    from multiprocessing import shared_memory
    pool = _SHM_CACHE.get(size)
    if pool:
        return pool.pop()
    return shared_memory.SharedMemory(create=True, size=size)


def _cached_bytes() -> int:
    return sum(size * len(v) for size, v in _SHM_CACHE.items())


def _drop_cached_blocks() -> None:
    for blocks in _SHM_CACHE.values():
        for b in blocks:
            try:
                b.close()
            except Exception:
                pass
            try:
                b.unlink()
            except Exception:
                pass
    _SHM_CACHE.clear()


def _release_shared(pool, blocks, free, readers, keep=False):
    """What a streamer / prefetcher owns outside the Python heap, given back exactly once (`close()`, `with`, garbage collection or
    interpreter exit -- `weakref.finalize`; it must not reference the owner): the pool's threads, the reader processes, the decode
    workers' mappings of the blocks, and the shared-memory blocks themselves."""
    import json
    import queue
    try:
        pool.shutdown(wait=True, cancel_futures=True)
    except Exception:
        pass
    for w in readers or []:
        try:
            w.stdin.close()
        except Exception:
            pass
    for w in readers or []:
        try:
            w.wait(timeout=5)
        except Exception:
            w.kill()
            try:
                w.wait(timeout=2)
            except Exception:
                pass
        for f in (w.stdout, w.stdin):
            try:
                f.close()
            except Exception:
                pass
    if keep and blocks:                  # a streamer's blocks go back to the cache while it has room (the workers keep their mappings of those)
        import atexit
        if not _SHM_CACHE:
            atexit.register(_drop_cached_blocks)
        rest = []
        for b in blocks:
            if _cached_bytes() + b.size <= _SHM_CACHE_MAX:
                _SHM_CACHE.setdefault(b.size, []).append(b)
            else:
                rest.append(b)
        blocks = rest
    if free is not None and blocks:      # the shared decode workers drop their mappings (else an unlinked block stays resident in each)
        line = json.dumps({"detach": [b.name for b in blocks]})
        held = []
        try:
            for _ in range(len(_DECODE_WORKERS)):
                held.append(free.get(timeout=5))      # (a worker busy for another streamer comes back within one decode)
        except queue.Empty:
            pass                                       # (the workers' own cap on cached mappings bounds what is left)
        sent = []
        for i, w in enumerate(held):              # ask every worker first, then collect: their munmaps (hundreds of MB each) run side by side
            try:
                w.stdin.write(line + "\n")
                w.stdin.flush()
                sent.append(i)
            except Exception:
                held[i] = _replace_decode_worker(w)
        for i in sent:
            try:
                if not held[i].stdout.readline().strip():
                    held[i] = _replace_decode_worker(held[i])
            except Exception:
                held[i] = _replace_decode_worker(held[i])
        for w in held:
            free.put(w)
    for b in blocks or []:
        try:
            b.close()
        except Exception:
            pass
        try:
            b.unlink()
        except Exception:
            pass


def _weak_call(ref, name, *args):
    """A pool task that does not keep its owner alive: queued / finished work items would otherwise hold the streamer through the bound
    method, and the LAST reference could then die in a pool thread -- the finalizer would run there, late, joining its own pool."""
    obj = ref()
    if obj is None:
        raise RuntimeError("the owner of this task was collected")
    return getattr(obj, name)(*args)


class _OwnsShared:
    """Context-manager / finalizer plumbing shared by `ClipStreamer` and `FramePrefetcher` (the reference's `DataLoader(num_workers=8)`,
    train.py:100-122, cleans up after itself: so do these)."""
    _finalizer = None

    def _own(self, pool, blocks=None, free=None, readers=None, keep=False):
        import weakref
        self._finalizer = weakref.finalize(self, _release_shared, pool, list(blocks or []), free, list(readers or []), keep)

    def _submit(self, name, *args):
        import weakref
        return self.pool.submit(_weak_call, weakref.ref(self), name, *args)

    def close(self):
        """Idempotent; also runs when the object is collected and at interpreter exit."""
        if self._finalizer is not None:
            self._finalizer()

    @property
    def closed(self) -> bool:
        return self._finalizer is None or not self._finalizer.alive

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class ClipStreamer(_OwnsShared):
    """`SomeonesLipClip.load` as a pipeline (the clip-level counterpart of the reference's DataLoader, inference.py:129-140):
    batches of `batch` frames are read by a pool of host threads (JPEG decode and `np.load` release the GIL) straight into
    PINNED staging buffers -- observed frames as the decoded bytes, a quarter of the fp32 size -- and copied to the device on a
    SIDE stream while the previous batch renders; `depth` batches are in flight.  Iterating yields `ClipTensors` whose tensors
    are the same values `load` returns (frames converted on the device by s2l_from8b, bit for bit the host conversion); a
    yielded batch stays valid until the iteration after the next one starts (work queued on it on the current stream up to that
    point is waited for before its device buffers are overwritten).  Owns threads, shared-memory blocks and (process mode) mappings
    in the decode workers: use it as a context manager or call `close()`; garbage collection and interpreter exit release them too."""

    def __init__(self, ds: "SomeonesLipClip", device, batch: int = 100, first: int = 0, count: Optional[int] = None,
                 workers: Optional[int] = None, depth: int = 2, mode: Optional[str] = None):
        """mode: "thread" -- the decoders run in this process's threads -- or "process": in `workers` spawned processes that write
        into shared-memory blocks (speech2lip_amd/_io_worker.py).  PIL's JPEG decoder holds the interpreter lock, so threads stop
        scaling at one core's ~600 images/s; processes scale with the cores.  Default: "process" for clips of >= 256 frames (the
        pool takes ~1 s to start), "thread" below."""
        from concurrent.futures import ThreadPoolExecutor
        self.ds, self.dev, self.batch = ds, torch.device(device), int(batch)
        n = len(ds)
        self.first = int(first)
        self.count = n - self.first if count is None else min(int(count), n - self.first)
        self.depth = max(1, int(depth))
        # (8, not "as many as there are cores": measured on a 256-core host the loader alone does 2 200 frames/s with 8 workers, 1 400
        #  with 32, 1 200 with 64 -- threads beyond what the interpreter lock can feed only fight over it)
        self.workers = int(workers) if workers else min(8, os.cpu_count() or 1)
        self.with_pose = ds.coord_files is not None and ds.mode != "test"
        self.with_frames = ds.mode != "test"
        self.mode = mode if mode is not None else ("process" if self.count >= 256 and (self.with_pose or self.with_frames) else "thread")
        if self.mode not in ("thread", "process"):
            raise ValueError("ClipStreamer mode must be 'thread' or 'process'")
        self.pool = ThreadPoolExecutor(self.workers)
        ns, B, FH, FW = self.depth + 1, self.batch, ds.face_h, ds.face_w
        self.procs = self.shm_frames = self.shm_coords = None
        blocks = []
        try:
            if self.mode == "process":
                from multiprocessing import shared_memory
                # plain `python _io_worker.py` children (they import numpy + PIL only and inherit nothing of this process's HIP state),
                # started once per process and shared by every streamer; a pool thread borrows an idle one per task
                self.procs = _decode_workers(self.workers)
                self.fshape, self.cshape = (B, FH, FW, 3), (B, FH, FW, 2)
                if self.with_frames:
                    self.shm_frames = []
                    for _ in range(ns):
                        self.shm_frames.append(_take_block(B * FH * FW * 3))
                        blocks.append(self.shm_frames[-1])
                if self.with_pose:
                    self.shm_coords = []
                    for _ in range(ns):
                        self.shm_coords.append(_take_block(B * FH * FW * 2 * 4))
                        blocks.append(self.shm_coords[-1])
        finally:
            self._own(self.pool, blocks, self.procs, keep=True)      # (from here on whatever exists is released, also when the rest of __init__ raises)
        pin = lambda *shape, dtype: [torch.empty(*shape, dtype=dtype).pin_memory() for _ in range(ns)]
        on = lambda *shape, dtype: [torch.empty(*shape, dtype=dtype, device=self.dev) for _ in range(ns)]
        self.h_coord = pin(B, FH, FW, 2, dtype=torch.float32) if self.with_pose else None
        self.d_coord = on(B, FH, FW, 2, dtype=torch.float32) if self.with_pose else None
        self.h_ori = pin(B, FH, FW, 3, dtype=torch.uint8) if self.with_frames else None
        self.d_ori8 = on(B, FH, FW, 3, dtype=torch.uint8) if self.with_frames else None
        self.d_ori = on(B, FH, FW, 3, dtype=torch.float32) if self.with_frames else None
        self.side = torch.cuda.Stream(self.dev)
        self.audio = torch.from_numpy(np.ascontiguousarray(ds.aud_features[self.first:self.first + self.count].astype(np.float32))).to(self.dev)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.face_zero, self.mask = up(ds.rgb_face_zero[None]), up(ds.mask_lip_canonical[None])

    def _decode(self, slot: int, j: int, i: int) -> None:
        ds = self.ds
        if self.procs is not None:      # a worker PROCESS decodes into shared memory; this thread only waits and copies to the pinned block
            jp = os.path.join(ds.dataset_folder, "ori_images_face", ds.image_files[i]) if self.with_frames else None
            cp = os.path.join(ds.dataset_folder, "coords", ds.coord_files[i]) if self.with_pose else None
            self._ask_worker([self.shm_frames[slot].name if self.with_frames else None, self.fshape,
                              self.shm_coords[slot].name if self.with_pose else None, self.cshape, j, jp, cp])
            if self.with_frames:
                np.copyto(self.h_ori[slot][j].numpy(), np.ndarray(self.fshape, np.uint8, buffer=self.shm_frames[slot].buf)[j])
            if self.with_pose:
                np.copyto(self.h_coord[slot][j].numpy(), np.ndarray(self.cshape, np.float32, buffer=self.shm_coords[slot].buf)[j])
            return
        if self.with_pose:
            self.h_coord[slot][j].numpy()[...] = np.load(os.path.join(ds.dataset_folder, "coords", ds.coord_files[i]))
        if self.with_frames:
            from PIL import Image
            with Image.open(os.path.join(ds.dataset_folder, "ori_images_face", ds.image_files[i])) as im:
                self.h_ori[slot][j].numpy()[...] = np.asarray(im.convert("RGB"))

    def _ask_worker(self, request) -> None:
        import json
        line = json.dumps(request)
        ans = _ask_decode_worker(self.procs, line)
        if not ans:                      # the worker died under this request: it has been replaced; one more try on a live one
            ans = _ask_decode_worker(self.procs, line)
        if ans != "ok":
            raise RuntimeError(f"decode worker: {ans or 'died'}")

    def __len__(self):
        return -(-self.count // self.batch)

    def __iter__(self):
        from collections import deque
        if self.closed:
            raise RuntimeError("ClipStreamer is closed")
        starts = list(range(self.first, self.first + self.count, self.batch))
        ns = self.depth + 1
        h2d_done = [None] * ns
        marks = deque(maxlen=max(1, ns - 1))      # marks[-1]: the consumer's stream at the start of this iteration, [-2]: of the previous ...
        pending = deque()

        def submit(k):
            slot = k % ns
            if h2d_done[slot] is not None:
                h2d_done[slot].synchronize()          # the staging buffers of this slot have left for the device
            s0 = starts[k]
            cnt = min(self.batch, self.first + self.count - s0)
            pending.append((k, slot, s0, cnt, [self._submit("_decode", slot, j, s0 + j) for j in range(cnt)]))
        try:
            for k in range(min(self.depth, len(starts))):
                submit(k)
            nxt = len(pending)
            while pending:
                k, slot, s0, cnt, futs = pending.popleft()
                for f in futs:
                    f.result()
                cur = torch.cuda.current_stream(self.dev)
                mark = torch.cuda.Event()
                mark.record(cur)                          # everything the consumer queued through iteration k-1
                marks.append(mark)
                with torch.cuda.stream(self.side):
                    # This slot's device buffers held batch k-ns, promised valid through iteration k-ns+1: wait for the consumer's
                    # stream as it stood at the start of iteration k-ns+2 (for the default depth 2: the previous iteration's start,
                    # so this copy still overlaps the work queued on batch k-1).
                    if k >= ns and len(marks) >= ns - 1:
                        self.side.wait_event(marks[0])
                    if self.with_pose:
                        self.d_coord[slot][:cnt].copy_(self.h_coord[slot][:cnt], non_blocking=True)
                    if self.with_frames:
                        self.d_ori8[slot][:cnt].copy_(self.h_ori[slot][:cnt], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                h2d_done[slot] = ev
                cur.wait_event(ev)
                if nxt < len(starts):
                    submit(nxt)
                    nxt += 1
                ori = from8b(self.d_ori8[slot][:cnt], out=self.d_ori[slot][:cnt]) if self.with_frames else None
                off = s0 - self.first
                idx = torch.arange(s0, s0 + cnt, dtype=torch.int64, device=self.dev)
                yield ClipTensors(audio=self.audio[off:off + cnt], index=idx, coord=self.d_coord[slot][:cnt] if self.with_pose else None,
                                  rgb_face_ori=ori, rgb_face_zero=self.face_zero, mask_lip_canonical=self.mask,
                                  lip_lefttop_x=self.ds.lefttop_x, lip_lefttop_y=self.ds.lefttop_y, height=self.ds.lip_h, width=self.ds.lip_w,
                                  names=["{:05d}".format(i + 1) for i in range(s0, s0 + cnt)])
        finally:
            # (an iteration abandoned half way -- an exception in the consumer, a `break` -- leaves decode tasks queued: drop those not
            #  started, wait for the running ones, so that `close()` / the next iteration find idle staging buffers)
            for _, _, _, _, futs in pending:
                for f in futs:
                    f.cancel()
            for _, _, _, _, futs in pending:
                for f in futs:
                    if not f.cancelled():
                        try:
                            f.result()
                        except Exception:
                            pass
            for ev in h2d_done:
                if ev is not None:
                    ev.synchronize()


class FramePrefetcher(_OwnsShared):
    """The reference's `DataLoader(num_workers > 0)` (train.py:136-140) for `load_one_frame`: the dictionaries of the frame
    indices in `order` are prepared up to `depth` frames ahead and come out in order, collated (`collate_batch`) `per_step` at a time.
    mode "thread" (default without a GPU; what the CPU tests use): a thread pool in this process; mode "process" (default with one):
    `workers` child processes (`_reader_worker.py`, each with its own `SomeonesLipClip`) that write a frame's tensors into shared
    memory -- the reader's PIL / numpy work (JPEG decode, the 8-bit resize of the negative window: 10 - 18 ms per frame) holds the
    interpreter lock, so loader threads top out near one core.  (torch's own DataLoader processes were tried first: their transport of a
    frame's ~16 MB measured 19 - 268 ms per iteration on the benchmark host.)  Owns threads, child processes and shared-memory slabs:
    a context manager; `close()` is idempotent and also runs on garbage collection / interpreter exit (as the reference's DataLoader
    reaps its workers, train.py:100-122)."""

    SLAB = 24 << 20      # bytes of shared memory per frame in flight (a May frame with its sync window: 16.3 MB)

    def __init__(self, ds: "SomeonesLipClip", order, workers: Optional[int] = None, depth: int = 8, per_step: int = 1, collate: bool = True,
                 pin_memory: Optional[bool] = None, mode: Optional[str] = None, device=None, sync_fields: Optional[bool] = None):
        """pin_memory (default: when a GPU is visible): every tensor arrives in page-locked memory, as `DataLoader(pin_memory=True)`
        does -- the ~16 MB a frame with its sync window carries then cross PCIe by DMA at ~50 GB/s instead of through a pageable
        staging copy (3.7 ms per frame measured).
        device (a CUDA device; default None = host tensors, what the reference's DataLoader yields): the floating-point tensors of every
        frame are ALSO copied to that device, by the loader thread on a side stream, as soon as the frame is read -- the trainer's
        `.to(device)` then finds them there and the step's 40 - 130 MB of PCIe traffic runs beside the previous step's kernels instead
        of at the head of its own.  The consumer's current stream waits for the copy when the frame is yielded.  Integer entries and
        scalars stay on the host (the trainer reads them as python numbers).
        sync_fields (default None = whatever `ds` is set to, i.e. the reference's behaviour): False leaves the sync-loss side inputs out of
        every frame (`SomeonesLipClip.load_sync_fields`) -- the loader of the iterations before `it > 100000`, which never read them."""
        from concurrent.futures import ThreadPoolExecutor
        if sync_fields is not None and bool(sync_fields) != ds.load_sync_fields:
            import copy
            ds = copy.copy(ds)              # (the caller's reader keeps its setting; the arrays are shared)
            ds.load_sync_fields = bool(sync_fields)
        self.ds, self.order, self.depth, self.per_step, self.collate = ds, list(order), max(1, int(depth)), max(1, int(per_step)), collate
        self.pin = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type != "cuda":
            raise ValueError("FramePrefetcher(device=...) takes a CUDA device")
        self.side = torch.cuda.Stream(self.device) if self.device is not None else None
        if self.device is not None:
            self.pin = True
        self.workers = int(workers) if workers else min(8, os.cpu_count() or 1)
        self.mode = mode if mode is not None else ("process" if torch.cuda.is_available() else "thread")
        if self.mode not in ("thread", "process"):
            raise ValueError("FramePrefetcher mode must be 'thread' or 'process'")
        self.free = self.procs = self.slabs = None
        head = None
        if self.mode == "process":
            import json

            def plain(o):      # (a value the worker could not rebuild must not silently become None)
                raise TypeError(f"FramePrefetcher(mode='process'): cfg value {o!r} of type {type(o).__name__} cannot be sent to the "
                                "reader processes (JSON); use mode='thread' or plain numbers / strings / lists / dicts in cfg")
            head = json.dumps({"folder": ds.dataset_folder, "mode": ds.mode, "cfg": ds.cfg, "img_ext": getattr(ds, "img_ext", ".jpg"),
                               "sync_fields": bool(ds.load_sync_fields)}, default=plain)
        self.pool = ThreadPoolExecutor(self.workers)
        self._blocks, procs = [], []
        try:
            if self.mode == "process":
                import queue
                import subprocess
                import sys
                from multiprocessing import shared_memory
                script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_reader_worker.py")
                self.free, self.slabs = queue.Queue(), queue.Queue()
                for _ in range(self.workers):      # started together; each builds its own reader (~2 s: python + torch import + the folder scan)
                    w = subprocess.Popen([sys.executable, "-u", script], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
                    procs.append(w)
                    w.stdin.write(head + "\n")
                    w.stdin.flush()
                for _ in range(self.depth + self.workers):
                    self._blocks.append(shared_memory.SharedMemory(create=True, size=self.SLAB))
                    self.slabs.put(self._blocks[-1])
                for w in procs:
                    if w.stdout.readline().strip() != "ready":
                        raise RuntimeError("FramePrefetcher: a reader process did not start")
                    self.free.put(w)
                self.procs = procs
        finally:
            self._own(self.pool, self._blocks, None, procs)

    def _load_in_process(self, i):
        import json
        w, slab = self.free.get(), self.slabs.get()
        try:      # (ONE finally for the slab: an empty or malformed reply line must not lose it)
            try:
                w.stdin.write(json.dumps({"index": int(i), "shm": slab.name}) + "\n")
                w.stdin.flush()
                line = w.stdout.readline()
            finally:
                self.free.put(w)
            if not line.strip():
                raise RuntimeError(f"reader process died (exit code {w.poll()}) while loading frame {int(i)}")
            man = json.loads(line)
            if "__error__" in man:
                raise RuntimeError(f"reader process: {man['__error__']}")
            d = {}
            for k, m in man.items():
                if "value" in m:
                    d[k] = m["value"]
                    continue
                a = np.ndarray(tuple(m["shape"]), np.dtype(m["dtype"]), buffer=slab.buf, offset=m["off"])
                if m["tensor"]:
                    t = torch.empty(tuple(m["shape"]), dtype=torch.from_numpy(np.empty(0, a.dtype)).dtype,
                                    pin_memory=self.pin and a.nbytes > 16384)
                    np.copyto(t.numpy(), a)      # (releases the interpreter lock for large arrays)
                    d[k] = t
                else:
                    d[k] = a.copy()
                del a
            return d
        finally:
            self.slabs.put(slab)

    def _load(self, i):
        d = self._load_in_process(i) if self.mode == "process" else self.ds.load_one_frame(i)
        if self.collate and self.per_step == 1:      # (collating stacks into fresh tensors: do it here, before pinning)
            d = collate_batch([d])
        if self.pin:
            d = {k: (v.pin_memory() if isinstance(v, torch.Tensor) and v.numel() > 4096 and not v.is_pinned() else v) for k, v in d.items()}
        if self.device is not None:
            with torch.cuda.stream(self.side):
                d = {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) and v.is_floating_point() and v.numel() > 16 else v)
                     for k, v in d.items()}
                ev = torch.cuda.Event()
                ev.record(self.side)
            d["__on_device__"] = ev
        return d

    def _hand_over(self, d):
        """a frame leaves the loader: the consumer's stream waits for its upload and becomes a user of its device blocks"""
        ev = d.pop("__on_device__", None)
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in d.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        return d

    def __iter__(self):
        from collections import deque
        if self.closed:
            raise RuntimeError("FramePrefetcher is closed")
        q, it = deque(), iter(self.order)
        try:
            for i in it:
                q.append(self._submit("_load", i))
                if len(q) >= self.depth:
                    break
            group = []
            while q:
                group.append(self._hand_over(q.popleft().result()))
                for i in it:
                    q.append(self._submit("_load", i))
                    break
                if len(group) == self.per_step or not q:
                    yield (group[0] if self.per_step == 1 else collate_batch(group)) if self.collate else group
                    group = []
        finally:      # an abandoned iteration: frames still in flight are dropped (not started) or waited for (running)
            for f in q:
                f.cancel()
            for f in q:
                if not f.cancelled():
                    try:
                        f.result()
                    except Exception:
                        pass


class FrameWriter:
    """`write_frames` as a pipeline: the 8-bit frames leave the device into a pinned buffer on a side stream, and a pool of host
    threads encodes and writes the files (PIL's encoder releases the GIL) while the GPU renders the next batch.
    `submit(frames, names)`; `close()` waits for every file (idempotent; `with FrameWriter(...) as w:` closes on the way out and
    re-raises the first encoder error)."""

    def __init__(self, out_dir: str, workers: Optional[int] = None, ext: str = ".jpg", depth: int = 3):
        import weakref
        from concurrent.futures import ThreadPoolExecutor
        os.makedirs(out_dir, exist_ok=True)
        self.out_dir, self.ext = out_dir, ext
        self.pool = ThreadPoolExecutor(int(workers) if workers else min(16, os.cpu_count() or 1))
        self.slots, self.depth, self.k, self.side = [], max(1, int(depth)), 0, None
        self.futs = []
        self._closed = False
        self._finalizer = weakref.finalize(self, self.pool.shutdown, wait=False)      # (threads only: nothing to unlink)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:                     # the caller is already unwinding: finish what was queued, do not mask its exception
            try:
                self.close()
            except Exception:
                pass
        return False

    def _save(self, arr, name):
        from PIL import Image
        Image.fromarray(arr, "RGB").save(os.path.join(self.out_dir, name + self.ext), quality=95)

    def _drain(self, pinned, ev, names):
        ev.synchronize()
        a = pinned.numpy()
        return [self.pool.submit(self._save, a[j], n) for j, n in enumerate(names)]

    def submit(self, frames: torch.Tensor, names) -> None:
        if self._closed:
            raise RuntimeError("FrameWriter is closed")
        u8 = frames if frames.dtype == torch.uint8 else to8b(frames)
        if u8.device.type != "cuda":
            self.futs.append(self.pool.submit(lambda: [self._save(a, n) for a, n in zip(u8.numpy(), names)]))
            return
        if self.side is None:
            self.side = torch.cuda.Stream(u8.device)
        slot = self.k % self.depth
        self.k += 1
        if len(self.slots) <= slot:
            self.slots.append([None, None])
        buf, busy = self.slots[slot]
        if busy is not None:
            for f in busy.result():          # the files of the batch that used this staging buffer are written
                f.result()
        if buf is None or buf.shape[0] < u8.shape[0] or buf.shape[1:] != u8.shape[1:]:
            buf = torch.empty(u8.shape, dtype=torch.uint8).pin_memory()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(u8.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            view = buf[:u8.shape[0]]
            view.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.side)
        u8.record_stream(self.side)
        job = self.pool.submit(self._drain, view, ev, list(names))
        self.slots[slot] = [buf, job]
        self.futs.append(job)

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        first = None
        try:
            for f in self.futs:
                try:
                    r = f.result()
                    for g in (r if isinstance(r, list) else []):
                        if hasattr(g, "result"):
                            g.result()
                except Exception as e:      # keep draining: every file that can be written is written
                    first = first or e
        finally:
            self.futs = []
            self.pool.shutdown(wait=True)
            self._finalizer.detach()
        if first is not None:
            raise first


def write_frames(frames: torch.Tensor, names, out_dir: str, ext: str = ".jpg") -> None:
    """rgb*255 -> 8-bit image files named %05d (inference.py:172-178; the reference converts to BGR only
    because cv2.imwrite expects it: the files hold the same RGB picture).  cv2.imwrite rounds and saturates."""
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    arr = (frames if frames.dtype == torch.uint8 else to8b(frames)).cpu().numpy()      # already-quantised frames pass through
    for a, name in zip(arr, names):
        Image.fromarray(a, "RGB").save(os.path.join(out_dir, name + ext), quality=95)
