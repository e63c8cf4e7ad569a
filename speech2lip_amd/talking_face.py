"""`TalkingFace` -- drop-in for the reference module on the lip-render hot path.

Mirrors `src/face_simple/models/tf_nerf.py` of CVMI-Lab/Speech2Lip for the May flag set
(v2 MLP, audio_net, audio_not_embed, use_time; no head pose / landmarks / text):

  * same constructor signature (tf_nerf.py:13-18), same attributes read by callers
    (`audio_dims`, `data_path`, ...), same state-dict keys (`encoder_conv.N.*`,
    `encoder_fc1.N.*`, `fc_*`, `pts_linears.N.*`, `output_linear.*`, and the dead
    `coord_linears.*`), so reference checkpoints load with `load_state_dict(strict=False)`;
  * same method signatures and return shapes: `audio_merge_forward` (:197), `rgb_forward`
    (:225-228), `post_fusion2_onlylip` (:287-304);
  * every method runs hand-written HIP kernels through the C-ABI of libs2l_hip.so.  There is
    no eager/CPU path: a missing library or a non-GPU tensor raises.

Beyond the reference surface, `render_clip` is the batched driver that replaces the per-frame
loop of inference.py:140-159 (audio encoder once per frame, one fused launch per clip).

The post-fusion U-Net (`post_fusion_unet`) is `speech2lip_amd.unet.SimpleUnetLight`
(eval mode); `canonical_depth_head` (tf_nerf.py:174-195) is the parameter the canonical-depth photometric loss trains
(speech2lip_amd.geometry.depth_photo_loss).
"""
from __future__ import annotations

import ctypes
import math
import random
from typing import Optional

import torch
import torch.nn as nn

from . import _abi, _modcache

_UNSUPPORTED_TRUE = ("use_attention", "use_audio_mel", "use_head_pose", "use_head_pose_net", "use_lms", "use_text")


class Embedder:
    """Sin/cos positional encoding descriptor (tf_nerf.py:391-425).  Only the metadata lives
    here; the encoding itself is evaluated inside the HIP kernels."""

    def __init__(self, multires, input_dims=29, include_input=True, log_sampling=True):
        self.multires = multires
        self.input_dims = input_dims
        self.include_input = include_input
        self.log_sampling = log_sampling
        self.max_freq_log2 = multires - 1
        self.num_freqs = multires
        self.out_dims = (input_dims if include_input else 0) + 2 * multires * input_dims


class PositionalEncodingTime:
    """Frame-index encoding descriptor (tf_nerf.py:427-442); holds the fp32 `div_term`."""

    def __init__(self, device, out_dims):
        self.out_dims = out_dims
        self.device = device
        self.div_term = torch.exp(torch.arange(0, out_dims, 2, dtype=torch.float) * -(math.log(10000.0) / out_dims))


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t: torch.Tensor, device, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a tensor")
    if t.device.type != "cuda":
        raise _abi.S2LError(f"{what}: tensor is on {t.device}; the lip-render path runs on the GPU only (no CPU fallback)")
    return t.detach().to(dtype=torch.float32).contiguous()


class TalkingFace(nn.Module):
    def __init__(self, device, cfg, mode="train",
                 use_viewdirs=False, coord_merge_audio=False,
                 W=256, D=8, coord_D=4, skips=[4],
                 uv_audio_dims=66, uv_dims=2, audio_dims=29, head_pose_dims=3,
                 time_multires=10, output_ch=3, **args):
        super().__init__()
        m = cfg["model"]
        for key in _UNSUPPORTED_TRUE:
            if m.get(key, False):
                raise NotImplementedError(f"model.{key}=True is outside the MI355X hot path (May flag set only)")
        if not (m.get("audio_net") and m.get("audio_not_embed") and m.get("use_audio", True) and m.get("use_time")
                and m.get("MLP_version") == "v2"):
            raise NotImplementedError("hot path supports audio_net + audio_not_embed + use_time + MLP_version 'v2'")
        if (W, D, list(skips), uv_dims, time_multires, output_ch) != (256, 8, [4], 2, 10, 3) or m.get("uv_embed", 10) != 10:
            raise NotImplementedError("kernels are specialised for W=256, D=8, skips=[4], uv_embed=10, time 20, rgb out")

        self.cfg = cfg
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.use_viewdirs = use_viewdirs
        self.coord_merge_audio = coord_merge_audio
        self.uv_audio_dims = uv_audio_dims
        self.use_attention = False
        self.use_audio_net = True
        self.use_uv_audio_sep = m.get("use_uv_audio_sep", True)
        self.audio_not_embed = True
        self.skips = list(skips)
        self.uv_dims = uv_dims
        self.audio_dims = 64  # with audio_net (tf_nerf.py:64-65)
        self.head_pose_dims = head_pose_dims
        self.use_audio = True
        self.N_sample = cfg["training"].get("n_sample_points", 16)
        self.use_head_pose = False
        self.use_head_pose_net = False
        self.use_time = True
        self.use_post_fusion = bool(m.get("use_post_fusion", False))
        self.use_lms = False
        self.use_text = False
        self.data_path = cfg["data"]["path"]
        self.use_light_unet = bool(m.get("use_light_unet", True))
        self.expand_lip_mask = bool(m.get("expand_lip_mask", False))
        self.MLP_version = "v2"

        self.uv_embedder = Embedder(10, input_dims=2)
        self.time_embedder_new = PositionalEncodingTime(self.device, 2 * time_multires)

        # Parameters: identical names/shapes to the reference so checkpoints interchange.
        self.encoder_conv = nn.Sequential(
            nn.Conv1d(29, 32, 3, stride=2, padding=1), nn.LeakyReLU(0.02, True),
            nn.Conv1d(32, 32, 3, stride=2, padding=1), nn.LeakyReLU(0.02, True),
            nn.Conv1d(32, 64, 3, stride=2, padding=1), nn.LeakyReLU(0.02, True),
            nn.Conv1d(64, 64, 3, stride=2, padding=1), nn.LeakyReLU(0.02, True))
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(0.02, True), nn.Linear(64, 64))
        self.coord_linears = nn.ModuleList(  # constructed but never used by any forward (tf_nerf.py:131-135)
            [nn.Linear(2, W)] + [nn.Linear(W, W) for _ in range(coord_D - 1)] + [nn.Linear(W, 64)])
        self.output_linear = nn.Linear(W, output_ch)
        self.fc_uv = nn.Linear(42, W)
        self.fc_uv_skip = nn.Linear(42, W)
        self.fc_audio = nn.Linear(64, W)
        self.fc_audio_skip = nn.Linear(64, W)
        self.fc_time = nn.Linear(20, W)
        self.fc_time_skip = nn.Linear(20, W)
        self.pts_linears = nn.ModuleList([nn.Linear(W, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(2 * W, W)
                                                             for i in range(D - 1)])
        if self.use_post_fusion:   # tf_nerf.py:55-59
            from .unet import SimpleUnetLight
            self.use_resnet = bool(m.get("use_resnet", False))
            self.post_fusion_channel = int(m.get("post_fusion_channel", 3))
            self.post_fusion_unet = SimpleUnetLight(cfg=cfg, n_channels=self.post_fusion_channel)
        if m.get("use_canonical_depth", False):   # tf_nerf.py:174-195
            self.canonical_depth_head = nn.Parameter(self._init_canonical_depth(cfg), requires_grad=True)
        self.to(self.device)

        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._tables = {}

    @staticmethod
    def _init_canonical_depth(cfg) -> torch.Tensor:
        """tf_nerf.py:174-193: the 3DMM depth of the canonical frame (`canonical_depth_init_path`, .npy), its holes filled with
        the mean positive depth inside the head mask (`canonical_head_mask.jpg`, cv2.imread(...)/255 binarised, channel 0 of
        cv2's BGR order = the blue channel), zero outside it; or N(0,1) of (canonical_depth_height, canonical_depth_width) when
        no init path is configured."""
        import os
        import numpy as np
        m = cfg["model"]
        if "canonical_depth_init_path" not in m:
            return torch.randn((int(m["canonical_depth_height"]), int(m["canonical_depth_width"])))
        init = torch.from_numpy(np.load(m["canonical_depth_init_path"])).float()
        head = init.clone()
        head[head == 0] = head[head > 0].mean()
        from PIL import Image
        mask = np.asarray(Image.open(os.path.join(cfg["data"]["path"], "canonical_head_mask.jpg")).convert("RGB"))[:, :, 2] / 255
        mask = torch.from_numpy((mask > 0).astype(np.int32))
        head[mask == 0] = 0
        head[init > 0] = init[init > 0]
        return head

    # ------------------------------------------------------------------ weights
    def _hot_tensors(self):
        """The 42 hot-path parameters in the C-ABI's order.  The owning sub-modules' `_parameters` dictionaries are looked up once
        (walking `named_parameters()` -- the U-Net's tree included -- cost 0.2 ms per call and a training step makes ~50) and the
        parent -> child links to them are re-checked by identity on every call (`_modcache.TensorSlots`): a parameter that is
        re-assigned is found (the dictionary is the module's own), a sub-module replaced at any depth resolves the slots again."""
        return _modcache.tensor_slots(self, _abi.TENSOR_ORDER, "_hot_cache")

    def packed_weights(self) -> torch.Tensor:
        """Device blob in kernel layout; rebuilt whenever a parameter was modified in place,
        replaced or moved (tracked through tensor identity + version counters)."""
        lib = _abi.load()
        tensors = self._hot_tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed is None or key != self._packed_key:
            dev = tensors[0].device
            if dev.type != "cuda":
                raise _abi.S2LError(f"TalkingFace parameters are on {dev}; the lip-render path needs a GPU (no CPU fallback)")
            holders = [_dev_f32(t, dev, "parameter") for t in tensors]
            table = (ctypes.c_void_p * len(holders))(*[h.data_ptr() for h in holders])
            div = (ctypes.c_float * 10)(*[float(v) for v in self.time_embedder_new.div_term.tolist()])
            packed = torch.empty(int(lib.s2l_packed_floats()), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_pack_weights(table, div, _ptr(packed), _stream()), "s2l_pack_weights")
            self._packed, self._packed_key = packed, key
            self._tables = {}
            self._packed_bf16 = None
        return self._packed

    def packed_weights_bf16(self) -> torch.Tensor:
        """bf16 operand images of the MLP for the bf16 training mode (csrc/s2l_bf16.h), rebuilt with the fp32 blob."""
        lib = _abi.load()
        packed = self.packed_weights()
        if getattr(self, "_packed_bf16", None) is None:
            dev = packed.device
            holders = [_dev_f32(t, dev, "parameter") for t in self._hot_tensors()]
            table = (ctypes.c_void_p * len(holders))(*[h.data_ptr() for h in holders])
            pb = torch.empty(int(lib.s2l_bf16_packed_halves()), dtype=torch.int16, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_pack_bf16(table, _ptr(packed), _ptr(pb), _stream()), "s2l_pack_bf16")
            # (no synchronisation: `holders` are the parameters themselves -- or, for a non-fp32 / non-contiguous parameter, blocks of
            # the stream-ordered caching allocator, which cannot be handed to another stream before this launch has run)
            self._packed_bf16 = pb
        return self._packed_bf16

    def load_state_dict(self, state_dict, strict=False, **kw):
        """Reference loader is strict=False (checkpoints.py:106): U-Net / depth-head keys that
        are outside this path are ignored rather than fatal."""
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # ------------------------------------------------------------------ A4
    def audio_merge_forward(self, audio):
        """[B,16,29] DeepSpeech windows (or already-permuted [B,29,16]) -> [B,64].
        Reference: tf_nerf.py:197-213.  With autograd recording the result carries a grad_fn whose backward is
        s2l_audio_backward (speech2lip_amd.autograd), so the reference's loss.backward() (training.py:559) works."""
        if torch.is_grad_enabled() and isinstance(audio, torch.Tensor) and audio.shape[0] > 0 and \
                any(p.requires_grad for p in self._hot_tensors()[:12]):
            from .autograd import audio_encode
            return audio_encode(self, audio)
        return self._audio_encode(audio)

    def _audio_encode(self, audio):
        """The forward kernel alone (no graph)."""
        lib = _abi.load()
        packed = self.packed_weights()
        a = _dev_f32(audio, packed.device, "audio")
        if a.dim() != 3:
            raise ValueError(f"audio must be [B,16,29], got {tuple(a.shape)}")
        if a.shape[2] == 16 and a.shape[1] == 29:      # already channel-major (tf_nerf.py:203-204)
            a = a.permute(0, 2, 1).contiguous()
        if a.shape[1:] != (16, 29):
            raise ValueError(f"audio must be [B,16,29], got {tuple(a.shape)}")
        out = torch.empty(a.shape[0], 64, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            _abi.check(lib.s2l_audio_encode(_ptr(packed), _ptr(a), _ptr(out), a.shape[0], _stream()), "s2l_audio_encode")
        return out

    # ------------------------------------------------------------------ A5
    def rgb_forward(self, uv_audio_pts, time_pts=None, head_pose_pts=None, rgb_pts=None, lms_pts=None, text_pts=None):
        """rows [N, 2+64] + one frame index -> [N,3] (no output activation).
        Reference: tf_nerf.py:225-285.  `head_pose_pts`, `rgb_pts`, `lms_pts`, `text_pts` are
        accepted and ignored exactly as the reference ignores them under the May flags.  With autograd recording the result
        is differentiable w.r.t. the MLP parameters and the audio columns of the rows (speech2lip_amd.autograd)."""
        if torch.is_grad_enabled() and isinstance(uv_audio_pts, torch.Tensor) and uv_audio_pts.shape[0] > 0 and \
                (uv_audio_pts.requires_grad or any(p.requires_grad for p in self._hot_tensors()[12:])):
            from .autograd import rgb_forward
            return rgb_forward(self, uv_audio_pts, time_pts)
        return self._rgb_forward(uv_audio_pts, time_pts)

    def _rgb_forward(self, uv_audio_pts, time_pts):
        """The forward kernels alone (no graph)."""
        lib = _abi.load()
        packed = self.packed_weights()
        rows = _dev_f32(uv_audio_pts, packed.device, "uv_audio_pts")
        if rows.dim() != 2 or rows.shape[1] != 66:
            raise ValueError(f"uv_audio_pts must be [N,66], got {tuple(rows.shape)}")
        if time_pts is None:
            raise ValueError("time_pts is required (model.use_time)")
        # PositionalEncodingTime uses position[0] only (tf_nerf.py:437-440)
        t = int(time_pts.reshape(-1)[0].item()) if isinstance(time_pts, torch.Tensor) else int(time_pts)
        n = rows.shape[0]
        out = torch.empty(n, 3, dtype=torch.float32, device=rows.device)
        xbuf = torch.empty(max(n, 1) * 128, dtype=torch.float32, device=rows.device)
        with torch.cuda.device(rows.device):
            _abi.check(lib.s2l_rgb_forward(_ptr(packed), _ptr(rows), t, _ptr(xbuf), _ptr(out), n, _stream()),
                       "s2l_rgb_forward")
        return out

    # ------------------------------------------------------------------ A7
    def _pad_mode(self) -> int:
        p = self.data_path
        if "macron" in p or "obama_adnerf" in p or "obama2_face_crop" in p or "may" in p:   # tf_nerf.py:345-348
            return _abi.S2L_PAD_MAY
        return _abi.S2L_PAD_DEFAULT

    def post_fusion2_onlylip(self, rgb_lip_warped, rgb_face_canonical, rgb_gt, mask_lip_canonical, lip_lefttop_x,
                             lip_lefttop_y, coord, use_canonical_space=False, change_pose=-1, mask_face_canonical=None,
                             wav2lip=None, mask_head_observed=None, use_post_fusion_blackaug=False, _frames_as_calls=False):
        """Paste the lip into the canonical face, warp by `coord`, blend with the observed frame.
        Reference: tf_nerf.py:287-304 -> post_fusion2_onlylip_light :320-389.
        Returns (rgb_recon, rgb_merged_new, rgb_merged_canonical), all [B,FH,FW,3]; rgb_recon is the post-fusion U-Net
        output (csrc/unet.hip): with the running statistics when `post_fusion_unet` is in eval mode (inference, and training
        once the net is fixed, train.py:188-197), with batch statistics (and parameter gradients) while it is in train mode;
        None when model.use_post_fusion is off.
        use_post_fusion_blackaug=True is the training call (training.py:436/445): with probability 1/2 black holes are
        punched (tf_nerf.py:371-384).  With autograd recording and a lip that requires grad, the outputs are
        differentiable w.r.t. the lip (speech2lip_amd.autograd)."""
        if not self.use_light_unet:
            return None  # the reference method falls through and returns None as well (:299-304)
        holes = None
        if use_post_fusion_blackaug and random.random() > 0.5:          # the coin of tf_nerf.py:371
            holes = self.draw_hole_noise(rgb_gt)
        unet = getattr(self, "post_fusion_unet", None)
        # rgb_recon = post_fusion_unet(rgb_merged_new) (tf_nerf.py:387): the sub-module's own mode decides -- train mode
        # (BatchNorm batch statistics, parameter gradients) until the reference fixes it, eval mode afterwards (train.py:188-197)
        graph = torch.is_grad_enabled() and isinstance(rgb_lip_warped, torch.Tensor) and rgb_lip_warped.requires_grad
        if graph:
            from .autograd import composite as composite_with_graph
            new, can = composite_with_graph(self, rgb_lip_warped, rgb_face_canonical, rgb_gt, mask_lip_canonical, lip_lefttop_x,
                                            lip_lefttop_y, coord, holes)
        else:
            new, can = self.composite_clip(rgb_lip_warped, rgb_face_canonical, rgb_gt, mask_lip_canonical, lip_lefttop_x,
                                           lip_lefttop_y, coord, want_canonical=True, hole_noise=holes)
        if unet is None:
            return None, new, can
        prec = getattr(self, "train_precision", "fp32")      # "bf16": set by Trainer(precision="bf16"); training calls only
        if unet.training:
            if torch.is_grad_enabled():
                from .autograd import unet_train
                # (_frames_as_calls: B frames standing for B successive one-frame calls -- Trainer.train_stage1's sync window)
                return unet_train(unet, new, prec, frames=_frames_as_calls), new, can
            return unet.forward_train_nhwc(new, update_running=True)[0], new, can
        if graph:
            from .autograd import unet_eval
            return unet_eval(unet, new, prec), new, can
        return unet.forward_nhwc(new), new, can

    # where the two N(0,1) fields of the black-hole augmentation come from: "host" = the reference's own stream (below);
    # "device" (opt-in) = one field of FH*FW draws each from the GPU's Philox generator: the same distribution, a different
    # stream, and none of the 7.4 ms per frame that 2 x 750 000 CPU normal draws + their copy cost on an 8-core host
    hole_noise = "host"

    def draw_hole_noise(self, rgb_gt, device=None):
        """The two N(0,1) fields of `add_black_hole` (tf_nerf.py:306-318).  hole_noise == "host" (default): drawn the way the
        reference draws them -- `torch.randn(input_img.shape)` on the default (CPU) generator, channel 0 kept, moved to the
        device -- so that a seeded run consumes the generator exactly like the reference: first for the merged image, then for
        rgb_gt.  hole_noise == "device": `torch.randn(B, FH, FW, device=...)` twice on the device generator."""
        B, FH, FW = rgb_gt.shape[0], rgb_gt.shape[1], rgb_gt.shape[2]
        dev = torch.device(device) if device is not None else rgb_gt.device
        if getattr(self, "hole_noise", "host") == "device":
            if dev.type != "cuda":
                dev = self.packed_weights().device
            return (torch.randn(B, FH, FW, device=dev), torch.randn(B, FH, FW, device=dev))
        if self.hole_noise != "host":
            raise ValueError(f"TalkingFace.hole_noise must be 'host' or 'device', got {self.hole_noise!r}")
        n1 = torch.randn(B, 3, FH, FW)[:, 0].contiguous()
        n2 = torch.randn(B, 3, FH, FW)[:, 0].contiguous()
        return n1.to(dev), n2.to(dev)

    def _composite_geometry(self, lw):
        if self.expand_lip_mask:
            return lw // 12 if "obama2_face_crop" in self.data_path else lw // 5   # tf_nerf.py:357-360
        return -1

    def composite_clip(self, rgb_lip, rgb_face_canonical, rgb_gt, mask_lip_canonical, lip_lefttop_x, lip_lefttop_y,
                       coord, want_canonical=False, out=None, hole_noise=None):
        """Batched composite (tf_nerf.py:320-386 without the U-Net) for F frames at once:
        rgb_lip [F,h,w,3], rgb_gt [F,FH,FW,3], coord [F,FH,FW,2]; rgb_face_canonical and
        mask_lip_canonical either per frame [F,FH,FW,3] or per clip [1,FH,FW,3] / [FH,FW,3].
        hole_noise: None (inference branch) or (n1, n2) [F,FH,FW]: the black-hole augmentation of the training branch.
        Returns (rgb_merged_new [F,FH,FW,3], rgb_merged_canonical or None)."""
        lib = _abi.load()
        dev = self.packed_weights().device
        lip = _dev_f32(rgb_lip, dev, "rgb_lip")
        face = _dev_f32(rgb_face_canonical, dev, "rgb_face_canonical")
        gt = _dev_f32(rgb_gt, dev, "rgb_gt")
        mask = _dev_f32(mask_lip_canonical, dev, "mask_lip_canonical")
        grid = _dev_f32(coord, dev, "coord")
        B, lh, lw = lip.shape[0], lip.shape[1], lip.shape[2]
        FH, FW = gt.shape[1], gt.shape[2]
        if gt.shape != (B, FH, FW, 3) or grid.shape != (B, FH, FW, 2) or lip.shape[3] != 3:
            raise ValueError("composite: inconsistent shapes")

        def stride(t, name):
            if t.shape == (1, FH, FW, 3) or t.shape == (FH, FW, 3):
                return 0
            if t.shape == (B, FH, FW, 3):
                return FH * FW * 3
            raise ValueError(f"composite: {name} must be [B,FH,FW,3] or [1,FH,FW,3]")

        x0 = int(lip_lefttop_x.reshape(-1)[0].item()) if isinstance(lip_lefttop_x, torch.Tensor) else int(lip_lefttop_x)
        y0 = int(lip_lefttop_y.reshape(-1)[0].item()) if isinstance(lip_lefttop_y, torch.Tensor) else int(lip_lefttop_y)
        pad = self._composite_geometry(lw)
        h1 = h2 = None
        if hole_noise is not None:
            h1, h2 = (_dev_f32(n, dev, "hole_noise") for n in hole_noise)
            if h1.shape != (B, FH, FW) or h2.shape != (B, FH, FW):
                raise ValueError(f"hole_noise fields must be [{B},{FH},{FW}]")
        if out is not None and (out.shape != (B, FH, FW, 3) or out.dtype != torch.float32 or not out.is_contiguous()
                                or out.device != dev):
            raise ValueError(f"out must be a contiguous fp32 [{B},{FH},{FW},3] tensor on {dev}")
        new = out if out is not None else torch.empty(B, FH, FW, 3, dtype=torch.float32, device=dev)
        can = torch.empty(B, FH, FW, 3, dtype=torch.float32, device=dev) if want_canonical else None
        fs, ms = stride(face, "rgb_face_canonical"), stride(mask, "mask_lip_canonical")
        with torch.cuda.device(dev):
            bgm = None
            if fs == 0 and ms == 0 and B >= 2:   # per-clip constants: fuse them once, gather half as much per frame
                bgm = torch.empty(FH, FW, 4, dtype=torch.float32, device=dev)
                _abi.check(lib.s2l_composite_tables(_ptr(face), _ptr(mask), _ptr(bgm), FH, FW, _stream()),
                           "s2l_composite_tables")
            per = FH * FW
            aligned = all(t.data_ptr() % 16 == 0 for t in (gt, grid, new))
            if bgm is not None and pad >= 0 and h1 is None and can is None and per % 4 == 0 and aligned and B <= 65535:
                # inference fast path: spans that cannot touch the rectangle are vector copies, the rest one pixel per thread
                work = torch.empty(int(lib.s2l_composite_stream_work_bytes(lh, lw, FH, FW, B)), dtype=torch.uint8, device=dev)
                _abi.check(lib.s2l_composite_stream(_ptr(lip), _ptr(mask), _ptr(bgm), _ptr(gt), _ptr(grid), _ptr(new), _ptr(work), lh, lw,
                                                    FH, FW, x0, y0, self._pad_mode(), pad, B, _stream()), "s2l_composite_stream")
            else:
                _abi.check(lib.s2l_composite_train(_ptr(lip), _ptr(face), fs, _ptr(mask), ms, _ptr(gt), _ptr(grid), _ptr(h1), _ptr(h2),
                                                   _ptr(new), _ptr(can), _ptr(bgm), lh, lw, FH, FW, x0, y0, self._pad_mode(), pad, B,
                                                   _stream()), "s2l_composite_train")
        return new, can

    def composite_backward_lip(self, d_new, rgb_face_canonical, mask_lip_canonical, lip_lefttop_x, lip_lefttop_y, coord,
                               lip_h: int, lip_w: int, hole_noise=None):
        """d loss / d rgb_merged_new [F,FH,FW,3] -> d loss / d rgb_lip [F,h,w,3]: the autograd of tf_nerf.py:339-386 for the lip
        (same constants and, in the training branch, the same hole_noise as the forward call)."""
        lib = _abi.load()
        dev = self.packed_weights().device
        d = _dev_f32(d_new, dev, "d_new")
        face = _dev_f32(rgb_face_canonical, dev, "rgb_face_canonical")
        mask = _dev_f32(mask_lip_canonical, dev, "mask_lip_canonical")
        grid = _dev_f32(coord, dev, "coord")
        B, FH, FW = d.shape[0], d.shape[1], d.shape[2]
        if grid.shape != (B, FH, FW, 2) or d.shape[3] != 3:
            raise ValueError("composite_backward_lip: inconsistent shapes")

        def stride(t, name):
            if t.shape == (1, FH, FW, 3) or t.shape == (FH, FW, 3):
                return 0
            if t.shape == (B, FH, FW, 3):
                return FH * FW * 3
            raise ValueError(f"composite: {name} must be [B,FH,FW,3] or [1,FH,FW,3]")

        x0 = int(lip_lefttop_x.reshape(-1)[0].item()) if isinstance(lip_lefttop_x, torch.Tensor) else int(lip_lefttop_x)
        y0 = int(lip_lefttop_y.reshape(-1)[0].item()) if isinstance(lip_lefttop_y, torch.Tensor) else int(lip_lefttop_y)
        h1 = h2 = None
        if hole_noise is not None:
            h1, h2 = (_dev_f32(n, dev, "hole_noise") for n in hole_noise)
        d_lip = torch.empty(B, int(lip_h), int(lip_w), 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _abi.check(lib.s2l_composite_backward_lip(_ptr(d), _ptr(face), stride(face, "rgb_face_canonical"), _ptr(mask),
                                                      stride(mask, "mask_lip_canonical"), _ptr(grid), _ptr(h1), _ptr(h2), _ptr(d_lip),
                                                      int(lip_h), int(lip_w), FH, FW, x0, y0, self._pad_mode(),
                                                      self._composite_geometry(int(lip_w)), B, _stream()), "s2l_composite_backward_lip")
        return d_lip

    # ------------------------------------------------------------------ A6 (batched driver)
    def pixel_tables(self, height: int, width: int):
        """Per-clip tables p0/p5 [HW,256] for the regular pixel grid (cached per size)."""
        from .rendering import shared_coords
        lib = _abi.load()
        packed = self.packed_weights()
        key = (int(height), int(width))
        if key not in self._tables:
            coords = shared_coords(width, height, packed.device)
            hw = coords.shape[0]
            rows = (hw + 15) // 16 * 16   # opaque renderer layout: 16-pixel groups
            p0 = torch.empty(rows, 256, dtype=torch.float32, device=packed.device)
            p5 = torch.empty_like(p0)
            with torch.cuda.device(packed.device):
                _abi.check(lib.s2l_pixel_tables(_ptr(packed), _ptr(coords), _ptr(p0), _ptr(p5), hw, _stream()),
                           "s2l_pixel_tables")
            self._tables[key] = (p0, p5)
        return self._tables[key]

    def packed_weights_split(self) -> torch.Tensor:
        """(hi | lo) IEEE-half A-operand slabs of the MLP for render_clip(precision="split") (csrc/render16.hip), rebuilt with the
        fp32 blob."""
        lib = _abi.load()
        packed = self.packed_weights()
        if getattr(self, "_packed_split", None) is None or self._packed_split_of is not packed:
            blob = torch.empty(int(lib.s2l_render16_packed_halves()), dtype=torch.int16, device=packed.device)
            with torch.cuda.device(packed.device):
                _abi.check(lib.s2l_pack_render16(_ptr(packed), _ptr(blob), _stream()), "s2l_pack_render16")
            self._packed_split, self._packed_split_of = blob, packed
        return self._packed_split

    def render_clip(self, audio, frame_idx, height: int, width: int, out: Optional[torch.Tensor] = None, _events=None,
                    precision: str = "fp32"):
        """audio [F,16,29] + frame indices [F] -> lip frames [F,H,W,3].
        precision: "fp32" (default) the exact fp32-MFMA kernel -- the parity mode and the headline; "split" the opt-in speed
        mode: every operand of the 256x256 layers as hi + lo IEEE-half parts (valid for |weight|, |pre-activation| < 65504; beyond it the
        parts saturate: finite, wrong), three f16 MFMAs per product, fp32 accumulation
        (~1e-6 of the output scale from the exact frames, far inside the north-star's RMSE <= 1e-4).

        Same function of its inputs as running the reference's per-frame loop
        (inference.py:140-159) F times, without its redundancy: the encoder runs once per frame
        (not once per pixel), coordinates/embeddings once per clip, and one fused launch renders
        all F*H*W samples."""
        lib = _abi.load()
        packed = self.packed_weights()
        dev = packed.device
        a = _dev_f32(audio, dev, "audio")
        if a.dim() != 3 or a.shape[1:] != (16, 29):
            raise ValueError(f"audio must be [F,16,29], got {tuple(a.shape)}")
        F = a.shape[0]
        idx = torch.as_tensor(frame_idx, device=dev).to(torch.int64).reshape(-1).contiguous()
        if idx.numel() != F:
            raise ValueError("frame_idx must have one entry per audio window")
        if precision not in ("fp32", "split"):
            raise ValueError("precision must be 'fp32' or 'split'")
        split = self.packed_weights_split() if precision == "split" else None
        hw = int(height) * int(width)
        p0, p5 = self.pixel_tables(height, width)
        if out is None:
            out = torch.empty(F, int(height), int(width), 3, dtype=torch.float32, device=dev)
        elif (out.shape != (F, int(height), int(width), 3) or out.dtype != torch.float32 or not out.is_contiguous()
              or out.device != dev):
            raise ValueError(f"out must be a contiguous fp32 [F,H,W,3] tensor on {dev}")
        q0 = torch.empty(F, 256, dtype=torch.float32, device=dev)
        q5 = torch.empty_like(q0)
        with torch.cuda.device(dev):
            st = _stream()      # the current stream OF dev: read it inside the guard
            if F < 4:           # the reference's mode, one frame per call: encoder + frame vectors in one launch (the same bits)
                _abi.check(lib.s2l_frame_front(_ptr(packed), _ptr(a), _ptr(idx), None, _ptr(q0), _ptr(q5), F, st), "s2l_frame_front")
            else:
                feat = torch.empty(F, 64, dtype=torch.float32, device=dev)
                _abi.check(lib.s2l_audio_encode(_ptr(packed), _ptr(a), _ptr(feat), F, st), "s2l_audio_encode")
                _abi.check(lib.s2l_frame_vectors(_ptr(packed), _ptr(feat), _ptr(idx), _ptr(q0), _ptr(q5), F, st),
                           "s2l_frame_vectors")
            if _events is not None:   # bench: HIP events on the launch stream around the dominant kernel
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if precision == "split":
                _abi.check(lib.s2l_render_lip_split(_ptr(packed), _ptr(split), _ptr(p0), _ptr(p5), _ptr(q0), _ptr(q5), _ptr(out), hw, F, st),
                           "s2l_render_lip_split")
            else:
                _abi.check(lib.s2l_render_lip(_ptr(packed), _ptr(p0), _ptr(p5), _ptr(q0), _ptr(q5), _ptr(out), hw, F, st),
                           "s2l_render_lip")
            if _events is not None:
                ev1.record()
                _events.append((ev0, ev1))
        return out


class FrameGraph:
    """The per-frame drop-in mode without launch overhead: `render_clip` (+ `composite_clip` [+ the post-fusion U-Net]) for a FIXED
    number of frames, captured once in a HIP graph (`torch.cuda.CUDAGraph`); every call copies the new inputs into the captured
    tensors and replays the graph -- one launch from the host instead of three to ~thirty.  The reference's loop hands the model
    one frame at a time (inference.py:128-140, DataLoader batch_size 1); this is that mode at the cost of the kernels alone.
    Outputs are the graph's own tensors: consume (or clone) them before the next call.  The library's contract makes this
    legal: no entry point allocates, frees or synchronises (include/s2l_hip.h:9-15; tests/test_gpu_concurrency.py pins
    replay == eager).

        fg = FrameGraph(model, frames=1, height=96, width=96)                       # lips only
        lip = fg(audio_window[None], [index])                                        # [1,96,96,3]
        fg = FrameGraph(model, 1, 128, 128, face=(rgb_face_canonical, mask_lip_canonical, x0, y0, 500, 500), unet=True)
        lip, merged_new, recon = fg(audio, idx, rgb_gt=frame, coord=grid)            # the whole of inference.py:140-172
    """

    def __init__(self, model: "TalkingFace", frames: int, height: int, width: int, face=None, unet: bool = False,
                 precision: str = "fp32"):
        self.model, self.F, self.h, self.w, self.precision = model, int(frames), int(height), int(width), precision
        dev = model.packed_weights().device
        self.audio = torch.zeros(self.F, 16, 29, dtype=torch.float32, device=dev)
        self.idx = torch.zeros(self.F, dtype=torch.int64, device=dev)
        self.face = None
        if face is not None:
            fc, mk, x0, y0, FH, FW = face
            self.face = (_dev_f32(fc, dev, "rgb_face_canonical"), _dev_f32(mk, dev, "mask_lip_canonical"), int(x0), int(y0))
            self.gt = torch.zeros(self.F, int(FH), int(FW), 3, dtype=torch.float32, device=dev)
            self.coord = torch.zeros(self.F, int(FH), int(FW), 2, dtype=torch.float32, device=dev)
        if unet and (face is None or getattr(model, "post_fusion_unet", None) is None or model.post_fusion_unet.training):
            raise ValueError("unet=True needs face=... and an eval-mode post_fusion_unet")
        self.unet = bool(unet)
        with torch.cuda.device(dev):
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):          # every cache (packs, tables, LDS opt-ins) is filled before the capture
                self._run()
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outputs = self._run()

    def _run(self):
        m = self.model
        lip = m.render_clip(self.audio, self.idx, self.h, self.w, precision=self.precision)
        if self.face is None:
            return (lip,)
        fc, mk, x0, y0 = self.face
        new, _ = m.composite_clip(lip, fc, self.gt, mk, x0, y0, self.coord)
        if not self.unet:
            return lip, new
        return lip, new, m.post_fusion_unet.forward_nhwc(new, precision="split" if self.precision == "split" else "fp32")

    def __call__(self, audio, frame_idx, rgb_gt=None, coord=None):
        self.audio.copy_(torch.as_tensor(audio).reshape(self.F, 16, 29), non_blocking=True)
        self.idx.copy_(torch.as_tensor(frame_idx).to(torch.int64).reshape(self.F), non_blocking=True)
        if self.face is not None:
            if rgb_gt is None or coord is None:
                raise ValueError("this graph composites: pass rgb_gt and coord")
            self.gt.copy_(rgb_gt.reshape(self.gt.shape), non_blocking=True)
            self.coord.copy_(coord.reshape(self.coord.shape), non_blocking=True)
        self.graph.replay()
        return self.outputs[0] if len(self.outputs) == 1 else self.outputs
