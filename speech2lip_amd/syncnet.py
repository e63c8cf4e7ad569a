"""`SyncNet_color` and the sync contrastive loss on the HIP path.

`SyncNet_color` has the module tree and state-dict keys of `src/face_simple/models/syncnet.py:7-67`
(`face_encoder.{i}.conv_block.{0,1}.*`, `audio_encoder.{i}.conv_block.{0,1}.*`), so a `lipsync_expert.pth` loads
unchanged.  It runs in eval mode only (the reference freezes it, training.py:85-90): BatchNorm is folded with its
running statistics when the weights are packed.  `SyncLoss` mirrors `Trainer.cosine_loss` /
`Trainer.get_sync_contrastive_loss` (training.py:576-603) and also returns the gradient of the loss with respect
to the generated window, which the reference obtains from autograd.  Everything runs in `csrc/syncnet.hip`; there
is no CPU fallback.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _abi
from .weights import SYNCNET_AUDIO, SYNCNET_BLOCKS, SYNCNET_FACE, SYNCNET_TENSORS


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Conv2d(nn.Module):
    """Parameter holder with the layout of conv.py:5-19 (conv -> BatchNorm [-> + x] -> ReLU)."""

    def __init__(self, cin, cout, kernel_size, stride, padding, residual=False):
        super().__init__()
        self.conv_block = nn.Sequential(nn.Conv2d(cin, cout, kernel_size, stride, padding), nn.BatchNorm2d(cout))
        self.act = nn.ReLU()
        self.residual = residual

    def forward(self, x):
        raise _abi.S2LError("SyncNet blocks run fused inside libs2l_hip.so; call SyncNet_color.forward")


class SyncNet_color(nn.Module):
    FACE_SHAPE = (15, 48, 96)     # [B,15,48,96]: 5 BGR frames, lower half of a 96x96 crop
    MEL_SHAPE = (1, 80, 16)

    def __init__(self):
        super().__init__()
        self.face_encoder = nn.Sequential(*[Conv2d(ci, co, k, s, p, r) for ci, co, k, s, p, r in SYNCNET_FACE])
        self.audio_encoder = nn.Sequential(*[Conv2d(ci, co, k, s, p, r) for ci, co, k, s, p, r in SYNCNET_AUDIO])
        self._packed = None
        self._packed_key = None
        self._work = None
        # what `precision=None` means in the embed calls.  "fp32": the exact-fp32 convolutions (default; what the parity tests pin).
        # "split": operands as hi + lo bf16 parts, three bf16 MFMAs per product, fp32 accumulation (csrc/conv_gemm.h: ~1e-5 relative)
        # -- the bf16-precision training steps ask for it per call.
        self.conv_precision = "fp32"
        self.eval()

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("SyncNet_color is a frozen eval-mode expert here, as in training.py:85-90")
        return super().train(False)

    def _tensors(self):
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        return [sd[f"{enc}.{i}.{t}"] for enc, i, _ in SYNCNET_BLOCKS for t in SYNCNET_TENSORS]

    def packed_weights(self) -> torch.Tensor:
        lib = _abi.load()
        tensors = self._tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed is None or key != self._packed_key:
            dev = tensors[0].device
            if dev.type != "cuda":
                raise _abi.S2LError(f"SyncNet parameters are on {dev}; the HIP path needs a GPU (no CPU fallback)")
            hold = [t.detach().to(torch.float32).contiguous() for t in tensors]
            table = (ctypes.c_void_p * len(hold))(*[h.data_ptr() for h in hold])
            packed = torch.empty(int(lib.s2l_syncnet_packed_floats()), dtype=torch.float32, device=dev)
            eps = float(self.face_encoder[0].conv_block[1].eps)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_syncnet_pack(table, ctypes.c_float(eps), _p(packed), _st()), "s2l_syncnet_pack")
                torch.cuda.current_stream().synchronize()      # `hold` may be temporaries
            self._packed, self._packed_key = packed, key
        return self._packed

    # -- the raw entry points (NHWC face windows) ---------------------------------------------------------------
    def _workspace(self, batch: int, dev) -> torch.Tensor:
        n = int(_abi.load().s2l_syncnet_work_floats(batch))
        if self._work is None or self._work.numel() < n or self._work.device != dev:
            self._work = torch.empty(n, dtype=torch.float32, device=dev)
        return self._work

    def embed_nhwc(self, mel: torch.Tensor, face_nhwc: torch.Tensor, precision: str = None):
        """mel [B,1,80,16] / [B,80,16]; face [B,48,96,15] -> (audio_emb [B,512], face_emb [B,512]), both L2-normalised.
        The activations stay in the module's workspace until the next call (face_backward uses them)."""
        lib = _abi.load()
        packed = self.packed_weights()
        dev = packed.device
        if mel.device != dev or face_nhwc.device != dev:
            raise _abi.S2LError("SyncNet inputs must be on the GPU that holds its weights (no CPU fallback)")
        mel = mel.detach().to(torch.float32).contiguous()
        face = face_nhwc.detach().to(torch.float32).contiguous()
        B = face.shape[0]
        if tuple(face.shape[1:]) != (48, 96, 15) or mel.numel() != B * 80 * 16:
            raise ValueError(f"SyncNet expects face [B,48,96,15] and mel [B,1,80,16]; got {tuple(face.shape)}, {tuple(mel.shape)}")
        a = torch.empty(B, 512, dtype=torch.float32, device=dev)
        v = torch.empty(B, 512, dtype=torch.float32, device=dev)
        work = self._workspace(B, dev)
        split = _conv_split(precision or self.conv_precision)
        with torch.cuda.device(dev):
            if split:
                _abi.check(lib.s2l_syncnet_forward_pair_split(_p(packed), _p(mel), _p(face), _p(work), _p(a), _p(v), B, B, _st()),
                           "s2l_syncnet_forward_pair_split")
            else:
                _abi.check(lib.s2l_syncnet_forward(_p(packed), _p(mel), _p(face), _p(work), _p(a), _p(v), B, _st()),
                           "s2l_syncnet_forward")
        self._last = (face, B, split)
        return a, v

    def embed_pair_nhwc(self, mel: torch.Tensor, face_nhwc: torch.Tensor, precision: str = None):
        """mel [B,...]; face [n B,48,96,15], n >= 1 (the generated windows first, then e.g. the negative ones of the same audio,
        training.py:592-601) -> (audio_emb [B,512], face_emb [n B,512]): each encoder runs ONCE (s2l_syncnet_forward_pair)."""
        lib = _abi.load()
        packed = self.packed_weights()
        dev = packed.device
        if mel.device != dev or face_nhwc.device != dev:
            raise _abi.S2LError("SyncNet inputs must be on the GPU that holds its weights (no CPU fallback)")
        mel = mel.detach().to(torch.float32).contiguous()
        face = face_nhwc.detach().to(torch.float32).contiguous()
        FB, B = face.shape[0], mel.numel() // (80 * 16)
        if tuple(face.shape[1:]) != (48, 96, 15) or mel.numel() != B * 80 * 16 or B < 1 or FB < B or FB % B:
            raise ValueError(f"SyncNet expects face [n B,48,96,15] and mel [B,1,80,16]; got {tuple(face.shape)}, {tuple(mel.shape)}")
        a = torch.empty(B, 512, dtype=torch.float32, device=dev)
        v = torch.empty(FB, 512, dtype=torch.float32, device=dev)
        work = self._workspace(FB, dev)
        split = _conv_split(precision or self.conv_precision)
        fn = "s2l_syncnet_forward_pair_split" if split else "s2l_syncnet_forward_pair"
        with torch.cuda.device(dev):
            _abi.check(getattr(lib, fn)(_p(packed), _p(mel), _p(face), _p(work), _p(a), _p(v), B, FB, _st()), fn)
        self._last = (face, FB, split)
        return a, v

    def face_backward(self, d_face_emb: torch.Tensor) -> torch.Tensor:
        """d loss / d face [B,48,96,15] from d loss / d face_emb [B,512], for the FIRST B windows of the last embed call."""
        lib = _abi.load()
        face, FB, split = self._last
        d = d_face_emb.detach().to(torch.float32).contiguous()
        B = d.shape[0]
        if B > FB or d.shape[1:] != (512,):
            raise ValueError(f"d_face_emb must be [B <= {FB},512], got {tuple(d.shape)}")
        out = torch.empty(B, *face.shape[1:], dtype=torch.float32, device=face.device)
        fn = "s2l_syncnet_face_backward_prefix_split" if split else "s2l_syncnet_face_backward_prefix"
        with torch.cuda.device(face.device):
            _abi.check(getattr(lib, fn)(_p(self._packed), _p(face), _p(self._work), _p(d), _p(out), B, FB, _st()), fn)
        return out

    def forward(self, audio_sequences, face_sequences):
        """syncnet.py:57-67: ([B,1,80,16], [B,15,48,96] NCHW) -> (audio_embedding, face_embedding)."""
        return self.embed_nhwc(audio_sequences, face_sequences.permute(0, 2, 3, 1))


def _conv_split(precision) -> bool:
    if precision not in ("fp32", "split"):
        raise ValueError(f"conv_precision must be 'fp32' or 'split', got {precision!r}")
    return precision == "split"


def sync_window(g_rgb: torch.Tensor, syncnet_T: int = 5, out: torch.Tensor = None) -> torch.Tensor:
    """[B,3,T,H,W] RGB window -> face [B,H-H//2,W,3T] NHWC (BGR, lower half rows, frames on channels; training.py:588-590).
    out: a contiguous [B,H-H//2,W,3T] fp32 view to write into (e.g. one half of a pair batch)."""
    g = g_rgb.detach().to(torch.float32).contiguous()
    if g.device.type != "cuda":
        raise _abi.S2LError("sync_window: input must be on the GPU (no CPU fallback)")
    B, C, T, H, W = g.shape
    if C != 3 or T != syncnet_T:
        raise ValueError(f"rgb window must be [B,3,{syncnet_T},H,W], got {tuple(g.shape)}")
    shape = (B, H - H // 2, W, 3 * T)
    if out is None:
        face = torch.empty(shape, dtype=torch.float32, device=g.device)
    else:
        if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != g.device:
            raise ValueError(f"sync_window: out must be a contiguous fp32 {shape} tensor on {g.device}")
        face = out
    with torch.cuda.device(g.device):
        _abi.check(_abi.load().s2l_sync_window(_p(g), _p(face), T, H, W, B, _st()), "s2l_sync_window")
    return face


def sync_window_backward(d_face: torch.Tensor, T: int, H: int, W: int) -> torch.Tensor:
    B = d_face.shape[0]
    d = d_face.detach().to(torch.float32).contiguous()
    out = torch.empty(B, 3, T, H, W, dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _abi.check(_abi.load().s2l_sync_window_backward(_p(d), _p(out), T, H, W, B, _st()), "s2l_sync_window_backward")
    return out


class SyncLoss:
    """`Trainer.cosine_loss` + `Trainer.get_sync_contrastive_loss` (training.py:576-603) for a frozen `SyncNet_color`."""

    def __init__(self, syncnet: SyncNet_color, syncnet_T: int = 5, precision: str = None):
        """precision: None (the module's `conv_precision`), "fp32" or "split" -- the form of the SyncNet's convolutions."""
        self.syncnet, self.syncnet_T, self.precision = syncnet, syncnet_T, precision

    def cosine_loss(self, a, v, y, weight: float = 1.0, want_grad: bool = False):
        """BCELoss(cosine_similarity(a, v).unsqueeze(1), y) [* weight]; with want_grad also d loss / d v."""
        lib = _abi.load()
        a = a.detach().to(torch.float32).contiguous()
        v = v.detach().to(torch.float32).contiguous()
        y = y.detach().to(torch.float32).reshape(-1).contiguous()
        B = a.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        scratch = torch.empty(B, dtype=torch.float32, device=a.device)
        dv = torch.empty_like(v) if want_grad else None
        with torch.cuda.device(a.device):
            _abi.check(lib.s2l_sync_loss(_p(a), _p(v), _p(y), ctypes.c_float(weight), _p(scratch), _p(loss), 0, _p(dv), B, _st()),
                       "s2l_sync_loss")
        return (loss[0], dv) if want_grad else loss[0]

    def get_sync_contrastive_loss(self, mel, g_rgb_pos, g_rgb_neg, syncnet_T=None, weight: float = 1.0, want_grad: bool = False):
        """loss = BCE(cos(a, v(pos)), 1) + BCE(cos(a, v(neg)), 0), times `weight` (the reference multiplies by w_syncloss at
        the call site, training.py:552).  With want_grad: (loss, d loss / d g_rgb_pos [B,3,T,H,W])."""
        T = syncnet_T or self.syncnet_T
        dev = g_rgb_pos.device
        B, _, _, H, W = g_rgb_pos.shape
        ones = torch.ones(B, dtype=torch.float32, device=dev)
        if tuple(g_rgb_neg.shape) != tuple(g_rgb_pos.shape):
            raise ValueError(f"positive and negative windows differ in shape: {tuple(g_rgb_pos.shape)}, {tuple(g_rgb_neg.shape)}")
        # the reference embeds (mel, pos) and (mel, neg) in two SyncNet calls; the net is frozen and in eval mode, every window's
        # embedding is a function of that window alone, so both face batches go through the encoder as one batch of 2B and the
        # audio encoder runs once
        face = torch.empty(2 * B, H - H // 2, W, 3 * T, dtype=torch.float32, device=dev)
        sync_window(g_rgb_pos, T, out=face[:B])
        sync_window(g_rgb_neg, T, out=face[B:])
        a, v = self.syncnet.embed_pair_nhwc(mel, face, precision=self.precision)
        if want_grad:
            pos, dv = self.cosine_loss(a, v[:B], ones, weight, True)
            d_pos = sync_window_backward(self.syncnet.face_backward(dv), T, H, W)
        else:
            pos = self.cosine_loss(a, v[:B], ones, weight)
        neg = self.cosine_loss(a, v[B:], torch.zeros_like(ones), weight)
        loss = pos + neg
        return (loss, d_pos) if want_grad else loss
