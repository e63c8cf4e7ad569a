"""Pose -> warp grid on the GPU: host mirror of the reference's geometry helpers in
`src/face_simple/models/utils.py`, backed by `csrc/warp.hip` through the C-ABI.

The reference ships the composite's warp grid as `coords/%05d.npy` (2 MB per 500x500 frame), written by
`preprocess/face_tracker.py:583-608` from a depth map and the relative pose Tc.inv(T).  The same grid is
regenerated here in HBM from `euler`/`trans` (24 B per frame) and a depth map, so that a clip needs audio
windows, poses and the observed frames only.  Function names and argument meaning follow the reference; the
batch is a whole clip instead of one frame, and there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _abi
from .talking_face import _dev_f32, _ptr, _stream

POSE_MODES = {"obs2can": _abi.S2L_POSE_OBS2CAN, "can2obs": _abi.S2L_POSE_CAN2OBS, "can2obs_inverse": _abi.S2L_POSE_CAN2OBS_INV}


def _rel_pose(canonical_euler, canonical_trans, euler, trans, mode: int) -> torch.Tensor:
    dev = euler.device
    e, t = _dev_f32(euler, dev, "euler").reshape(-1, 3), _dev_f32(trans, dev, "trans").reshape(-1, 3)
    ce, ct = _dev_f32(canonical_euler, dev, "canonical_euler").reshape(-1), _dev_f32(canonical_trans, dev, "canonical_trans").reshape(-1)
    if e.shape != t.shape or ce.numel() < 3 or ct.numel() < 3:
        raise ValueError("euler/trans must be [F,3]; canonical_euler/canonical_trans [3] or [1,3]")
    out = torch.empty(e.shape[0], 4, 4, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):   # launch on the tensors' device and ITS current stream, whatever torch's current device is
        _abi.check(_abi.load().s2l_rel_pose(_ptr(e), _ptr(t), _ptr(ce), _ptr(ct), mode, _ptr(out), e.shape[0], _stream()),
                   "s2l_rel_pose")
    return out


def compute_rel_pose_from_obs2can(canonical_euler, canonical_trans, euler, trans, img_batch_size=None, device=None):
    """Tc . inv(T)  (utils.py:54-58; the pose behind coords/*.npy, face_tracker.py:583-584)."""
    return _rel_pose(canonical_euler, canonical_trans, euler, trans, _abi.S2L_POSE_OBS2CAN)


def compute_rel_pose(canonical_euler, canonical_trans, euler, trans, img_batch_size=None, device=None):
    """T . inv(Tc)  (utils.py:66-71, training.py:263-268)."""
    return _rel_pose(canonical_euler, canonical_trans, euler, trans, _abi.S2L_POSE_CAN2OBS)


compute_rel_pose_from_can2obs = compute_rel_pose      # utils.py:60-64 is the same product


def compute_rel_pose_inverse(canonical_euler, canonical_trans, euler, trans, img_batch_size=None, device=None):
    """inv(T . inv(Tc))  (utils.py:73-77, training.py:270-275)."""
    return _rel_pose(canonical_euler, canonical_trans, euler, trans, _abi.S2L_POSE_CAN2OBS_INV)


def warp_grid(depth: torch.Tensor, rel_pose: torch.Tensor, focal: float, clamp: bool = False, return_z: bool = False,
              out: Optional[torch.Tensor] = None):
    """BackprojectDepth + Project3D (utils.py:115-169) with K from `focal` and the image centre (training.py:298-302).
    depth [H,W] (shared by the clip) or [F,H,W]; rel_pose [F,4,4] -> grid [F,H,W,2] (and z [F,1,H,W])."""
    dev = rel_pose.device
    T = _dev_f32(rel_pose, dev, "rel_pose").reshape(-1, 16)
    d = _dev_f32(depth, dev, "depth")
    F = T.shape[0]
    if d.dim() == 3 and d.shape[0] == 1:
        d = d[0]
    if d.dim() == 3 and d.shape[0] != F:
        raise ValueError(f"depth has {d.shape[0]} frames, rel_pose {F}")
    H, W = d.shape[-2:]
    stride = 0 if d.dim() == 2 else H * W
    grid = out if out is not None else torch.empty(F, H, W, 2, device=dev, dtype=torch.float32)
    if grid.shape != (F, H, W, 2) or not grid.is_contiguous() or grid.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 [F,H,W,2] tensor")
    z = torch.empty(F, 1, H, W, device=dev, dtype=torch.float32) if return_z else None
    if grid.device != dev:
        raise ValueError(f"out is on {grid.device}, rel_pose on {dev}")
    with torch.cuda.device(dev):
        _abi.check(_abi.load().s2l_warp_grid(_ptr(d), stride, _ptr(T), float(focal), int(bool(clamp)), _ptr(grid), _ptr(z), H, W,
                                             F, _stream()), "s2l_warp_grid")
    return (grid, z) if return_z else grid


def grid_sample(img_nhwc: torch.Tensor, grid: torch.Tensor, padding_mode: str = "zeros") -> torch.Tensor:
    """F.grid_sample(bilinear, align_corners=False) for NHWC 3-channel images; img [H,W,3] / [1,H,W,3] is shared by
    all frames of `grid` [F,Ho,Wo,2].  Returns NHWC [F,Ho,Wo,3]."""
    dev = grid.device
    g = _dev_f32(grid, dev, "grid")
    im = _dev_f32(img_nhwc, dev, "img")
    if im.dim() == 4 and im.shape[0] == 1:
        im = im[0]
    F, Ho, Wo = g.shape[:3]
    if im.shape[-1] != 3 or (im.dim() == 4 and im.shape[0] != F):
        raise ValueError("img must be [H,W,3], [1,H,W,3] or [F,H,W,3]")
    IH, IW = im.shape[-3:-1]
    pad = {"zeros": _abi.S2L_SAMPLE_ZEROS, "border": _abi.S2L_SAMPLE_BORDER}[padding_mode]
    out = torch.empty(F, Ho, Wo, 3, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _abi.check(_abi.load().s2l_grid_sample(_ptr(im), 0 if im.dim() == 3 else IH * IW * 3, _ptr(g), _ptr(out), IH, IW, Ho,
                                               Wo, pad, F, _stream()), "s2l_grid_sample")
    return out


def inverse_warping(cfg, tgt_depth: torch.Tensor, rel_pose: torch.Tensor, src_img: torch.Tensor, face_mask=None,
                    device=None, return_z: bool = False):
    """utils.py:202-226 / Trainer.inverse_warping (training.py:296-314): sample `src_img` [F or 1,H,W,3] at the grid of
    (tgt_depth [H,W], rel_pose [F,4,4]) with border padding.  Returns NCHW like the reference (a permuted view of
    the NHWC result), plus cam_points_z [F,1,H,W] when return_z."""
    focal = cfg["data"]["face_img_focal"]
    res = warp_grid(tgt_depth, rel_pose, focal, return_z=return_z)
    grid, z = res if return_z else (res, None)
    img = grid_sample(src_img, grid, "border").permute(0, 3, 1, 2)
    return (img, z) if return_z else img


def coords_for_clip(depth: torch.Tensor, canonical_euler, canonical_trans, euler, trans, focal: float,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The `coords/%05d.npy` grids of a clip, as face_tracker.py:583-606 computes them (obs->can pose, [-1,1] clamp)."""
    T = compute_rel_pose_from_obs2can(canonical_euler, canonical_trans, euler, trans)
    return warp_grid(depth, T, focal, clamp=True, out=out)


def depth_photo_loss(cfg, tgt_depth: torch.Tensor, rel_pose: torch.Tensor, src_img: torch.Tensor, target: torch.Tensor,
                     mask: Optional[torch.Tensor] = None, weights: float = 1.0, want_grad: bool = False):
    """Canonical-depth photometric loss (training.py:462-477): `src_img` [F,H,W,3] (rgb_face_gt) warped into the canonical view
    by `tgt_depth` [H,W] (model.canonical_depth_head) and `rel_pose` [F,4,4] (compute_rel_pose_inverse), against `target`
    (rgb_face_canonical, [H,W,3] / [1,H,W,3] / [F,H,W,3]) under `mask` (same shapes, or None):
        weights * sum((pred - target)^2 * mask) / (sum(mask) + 1e-6)          (add_loss_canonical_depth_photo, :621-634)
    One fused kernel evaluates the projection, the border-padded bilinear sampling, the loss and -- want_grad -- d loss / d depth.
    With autograd recording and a depth that requires grad the returned loss is differentiable (the reference's
    loss.backward() then fills canonical_depth_head.grad)."""
    if not want_grad and torch.is_grad_enabled() and isinstance(tgt_depth, torch.Tensor) and tgt_depth.requires_grad:
        return _DepthPhotoLoss.apply(tgt_depth, cfg, rel_pose, src_img, target, mask, float(weights))
    lib = _abi.load()
    dev = rel_pose.device
    T = _dev_f32(rel_pose, dev, "rel_pose").reshape(-1, 16)
    d = _dev_f32(tgt_depth, dev, "tgt_depth")
    src = _dev_f32(src_img, dev, "src_img")
    F, H, W = src.shape[0], src.shape[1], src.shape[2]
    if d.shape != (H, W) or T.shape[0] != F or src.shape[3] != 3:
        raise ValueError("depth_photo_loss: depth [H,W], rel_pose [F,4,4], src_img [F,H,W,3]")

    def shared(t, name):
        t = _dev_f32(t, dev, name)
        if t.shape in ((H, W, 3), (1, H, W, 3)):
            return t, 0
        if t.shape == (F, H, W, 3):
            return t, H * W * 3
        raise ValueError(f"depth_photo_loss: {name} must be [H,W,3], [1,H,W,3] or [F,H,W,3]")

    tg, ts = shared(target, "target")
    mk, ms = shared(mask, "mask") if mask is not None else (None, 0)
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    dd = torch.empty(H, W, dtype=torch.float32, device=dev) if want_grad else None
    work = torch.empty(int(lib.s2l_depth_photo_work_floats(H, W)), dtype=torch.float32, device=dev)
    import ctypes
    with torch.cuda.device(dev):
        _abi.check(lib.s2l_depth_photo_loss(_ptr(d), _ptr(T), ctypes.c_float(float(cfg["data"]["face_img_focal"])), _ptr(src), _ptr(tg), ts,
                                            _ptr(mk), ms, ctypes.c_float(float(weights)), _ptr(work), _ptr(loss), _ptr(dd), H, W, F,
                                            _stream()), "s2l_depth_photo_loss")
    return (loss[0], dd) if want_grad else loss[0]


class _DepthPhotoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, cfg, rel_pose, src_img, target, mask, weights):
        loss, dd = depth_photo_loss(cfg, depth.detach(), rel_pose, src_img, target, mask, weights, want_grad=True)
        ctx.save_for_backward(dd)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        (dd,) = ctx.saved_tensors
        return dd * d_loss, None, None, None, None, None, None
