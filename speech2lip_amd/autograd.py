"""`torch.autograd.Function`s over the hand-written forward/backward kernels, so that the reference's training loop --
forward through the module methods, a loss built with torch ops, `loss["loss"].backward()` (src/face_simple/training.py:559),
`optimizer.step()` -- runs unchanged on the drop-in module.

Each Function's forward is the same C-ABI call the plain method makes (plus saved activations); its backward is the matching
hand-written backward kernel.  torch's autograd engine is the plumbing that chains them (its only arithmetic here is scaling a
stored gradient by the upstream scalar of a loss node).
The parameters are passed to `apply` as inputs so that the engine delivers their gradients to `.grad`.

    audio_encode          TalkingFace.audio_merge_forward   tf_nerf.py:197-213    backward: s2l_audio_backward
    rgb_forward           TalkingFace.rgb_forward           tf_nerf.py:225-285    backward: s2l_train_backward + s2l_wgrad + un-fold
    predict_lip_image     Trainer.predict_lip_image         training.py:158-251   the fused 4-tap ensemble (LipTrainStep)
    composite             post_fusion2_onlylip_light        tf_nerf.py:320-386    backward: s2l_composite_backward_lip (d lip)
    unet_eval             post_fusion_unet (frozen, eval)   SimpleUnetLight.py:99-111   backward: s2l_unet_backward (d input)
    unet_train            post_fusion_unet (train mode)     SimpleUnetLight.py:99-111   backward: s2l_unet_train_backward (d input + 32 tensors)
    crop_resize           crop + transforms.Resize          training.py:541-544   backward: s2l_crop_resize_backward
    sync_contrastive_loss get_sync_contrastive_loss         training.py:581-603   backward: s2l_syncnet_face_backward (d window)
    mse                   add_photometric_loss              training.py:605-619   backward: the gradient s2l_mse returns
Gradients flow to: the 42 hot-path parameters, the audio columns of `rgb_forward`'s rows, the lip image, the U-Net input, the
generated sync window, the prediction of the MSE.  Inputs the reference treats as data (audio windows, pixel coordinates,
observed frames, warp grids) get none.
"""
from __future__ import annotations

import ctypes

import torch

from . import _abi
from .talking_face import _dev_f32, _ptr, _stream

_MLP = slice(12, 42)
_AUD = slice(0, 12)


def _f(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


class _AudioEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, audio, *params):
        ctx.model = model
        a = _dev_f32(audio, model.packed_weights().device, "audio")
        if a.dim() == 3 and a.shape[2] == 16 and a.shape[1] == 29:
            a = a.permute(0, 2, 1).contiguous()
        ctx.save_for_backward(a)
        return model._audio_encode(a)

    @staticmethod
    def backward(ctx, dfeat):
        from .training import AUDIO_TENSORS, audio_backward
        (a,) = ctx.saved_tensors
        with torch.cuda.device(a.device):
            g = audio_backward(ctx.model, a, dfeat.contiguous().float(), _stream())
        return (None, None, *[g[n] for n in AUDIO_TENSORS])


def audio_encode(model, audio):
    return _AudioEncode.apply(model, audio, *model._hot_tensors()[_AUD])


class _RgbForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rows, time_index, *params):
        from .training import MlpState, mlp_forward
        lib = _abi.load()
        packed = model.packed_weights()
        dev = packed.device
        r = _dev_f32(rows, dev, "uv_audio_pts")
        if r.dim() != 2 or r.shape[1] != 66:
            raise ValueError(f"uv_audio_pts must be [N,66], got {tuple(r.shape)}")
        n = r.shape[0]
        st = MlpState("fp32", n, dev, lib)
        out = _f(dev, n, 3)
        with torch.cuda.device(dev):
            s = _stream()
            _abi.check(lib.s2l_embed_rows(_ptr(packed), _ptr(r), int(time_index), _ptr(st.x), n, s), "s2l_embed_rows")
            mlp_forward(model, st, out, s)
        ctx.model, ctx.st = model, st
        return out

    @staticmethod
    def backward(ctx, dout):
        from .training import MLP_TENSORS, mlp_backward
        st = ctx.st
        d = dout.contiguous().float()
        with torch.cuda.device(d.device):
            g, dxa = mlp_backward(ctx.model, st, d, _stream())
        ctx.st = None
        d_rows = torch.cat([torch.zeros(st.N, 2, dtype=torch.float32, device=d.device), dxa], dim=1)   # coordinates are data
        return (None, d_rows, None, *[g[n] for n in MLP_TENSORS])


def rgb_forward(model, rows, time_pts):
    if time_pts is None:
        raise ValueError("time_pts is required (model.use_time)")
    t = int(time_pts.reshape(-1)[0].item()) if isinstance(time_pts, torch.Tensor) else int(time_pts)   # position[0] only
    return _RgbForward.apply(model, rows, t, *model._hot_tensors()[_MLP])


class _PredictLipImage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, audio, index, height, width, u01, precision, *params):
        from .training import LipTrainStep
        step = LipTrainStep(model, height, width, precision)
        many = isinstance(index, (list, tuple))
        pred = step.forward(audio, list(index) if many else [index], u01 if isinstance(u01, torch.Tensor) else ([u01] if not many else u01))
        ctx.step, ctx.many = step, many
        return pred if many else pred[0]

    @staticmethod
    def backward(ctx, dpred):
        d = dpred.contiguous().float()
        g, _ = ctx.step.backward(d if ctx.many else d[None])
        ctx.step = None
        return (None, None, None, None, None, None, None, *[g[n] for n in _abi.TENSOR_ORDER])


_REGULAR_GRIDS = {}      # (data_ptr, shape) -> (height, width, the tensor: kept alive so that its address cannot be re-used)


def register_regular_grid(coords, height, width):
    """Declare `coords` (a tensor the caller will not modify) to be get_coords(width, height): Trainer.prepare_coords does, once
    per size, which replaces a device-to-host copy and an element-wise comparison per rendered frame."""
    _REGULAR_GRIDS[(coords.data_ptr(), tuple(coords.shape))] = (int(height), int(width), coords)


def predict_lip_image(model, coords, audio, index, height, width, u01, precision="fp32"):
    """The regular-grid 4-tap ensemble of one frame with a graph (fp32 parity mode by default; precision="bf16": the bf16 MFMA
    kernels of BASELINE config 5).  `coords` must be the regular pixel grid of (height, width) -- what Trainer.prepare_coords
    returns -- because the fused kernels rebuild it."""
    from .rendering import get_coords
    known = _REGULAR_GRIDS.get((coords.data_ptr(), tuple(coords.shape)))
    if not (known is not None and known[:2] == (int(height), int(width))) and \
            (coords.shape[0] != height * width or not torch.equal(coords.to(torch.float32).cpu(), get_coords(width, height, "cpu"))):
        raise ValueError("predict_lip_image with autograd supports the regular pixel grid of (height, width) only")
    if audio.shape[0] != 1:
        raise ValueError("predict_lip_image renders one frame: audio must be [1,16,29]")
    idx = int(index.reshape(-1)[0].item()) if isinstance(index, torch.Tensor) else int(index)
    u = u01 if isinstance(u01, torch.Tensor) and u01.is_cuda else float(u01)      # a device draw stays on the device
    return _PredictLipImage.apply(model, audio, idx, int(height), int(width), u, precision, *model._hot_tensors())


def predict_lip_images(model, coords, audio, indices, height, width, u01, precision="fp32"):
    """`predict_lip_image` for B frames in one call: audio [B,16,29], B frame indices, B draws (a device tensor or floats) ->
    [B,HW,3].  Every frame's rows are what its own call would compute (the rows of a frame never mix with another's); one MLP
    forward / backward instead of B, which is what makes a 5-frame sync window cost one launch set."""
    from .rendering import get_coords
    known = _REGULAR_GRIDS.get((coords.data_ptr(), tuple(coords.shape)))
    if not (known is not None and known[:2] == (int(height), int(width))) and \
            (coords.shape[0] != height * width or not torch.equal(coords.to(torch.float32).cpu(), get_coords(width, height, "cpu"))):
        raise ValueError("predict_lip_images with autograd supports the regular pixel grid of (height, width) only")
    idx = [int(i) for i in indices]
    if audio.shape[0] != len(idx):
        raise ValueError("predict_lip_images: one audio window per frame index")
    u = u01 if isinstance(u01, torch.Tensor) and u01.is_cuda else [float(v) for v in u01]
    return _PredictLipImage.apply(model, audio, idx, int(height), int(width), u, precision, *model._hot_tensors())


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, lip, face, gt, mask, x0, y0, coord, holes):
        new, can = model.composite_clip(lip, face, gt, mask, x0, y0, coord, want_canonical=True, hole_noise=holes)
        ctx.model, ctx.args, ctx.lip_hw = model, (face, mask, x0, y0, coord, holes), (lip.shape[1], lip.shape[2])
        ctx.mark_non_differentiable(can)      # rgb_merged_canonical is a by-product no loss of the reference reads
        return new, can

    @staticmethod
    def backward(ctx, d_new, _d_can):
        face, mask, x0, y0, coord, holes = ctx.args
        d_lip = ctx.model.composite_backward_lip(d_new, face, mask, x0, y0, coord, ctx.lip_hw[0], ctx.lip_hw[1], hole_noise=holes)
        return None, d_lip, None, None, None, None, None, None, None


def composite(model, lip, face, gt, mask, x0, y0, coord, holes=None):
    return _Composite.apply(model, lip, face, gt, mask, x0, y0, coord, holes)


class _UnetEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, unet, x, precision):
        out, saved = unet.forward_for_backward(x, precision=precision) if precision != "fp32" else unet.forward_saved_nhwc(x)
        ctx.unet, ctx.saved = unet, saved
        return out

    @staticmethod
    def backward(ctx, d_out):
        dx = ctx.unet.backward_input(ctx.saved, d_out)
        ctx.saved = None
        return None, dx, None


def unet_eval(unet, x_nhwc, precision="fp32"):
    """Frozen eval-mode post-fusion U-Net with an input gradient (no parameter gradients: the reference has set
    requires_grad=False on them by the time this path is used, train.py:188-197).  precision "bf16": bf16 operands and
    tensors between the kernels (the half-width chain)."""
    return _UnetEval.apply(unet, x_nhwc, precision)


class _UnetTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, unet, x, precision, frames, *params):
        ctx.need_params = any(p_.requires_grad for p_ in params)
        # `frames`: every frame of x is its own statistics group, as if the net had been called once per frame in frame order (the
        # mode-following frames route of forward_for_backward: the bits of those calls) -- what train_stage1's batched sync window
        # asks for.  Otherwise x is ONE batch, statistics over all of it, as calling the module means in the reference; one bf16
        # frame is the same thing either way and takes the half-width chain of the frames route
        ctx.unet, ctx.need_dx = unet, x.requires_grad
        ctx.route = bool(frames) or (precision != "fp32" and x.shape[0] == 1)
        if ctx.route:
            out, ctx.saved = unet.forward_for_backward(x, precision=precision)
            return out
        out, saved = unet.forward_train_nhwc(x, update_running=True, precision=precision)
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, d_out):
        # a frozen net in train-mode BatchNorm (the reference's loop after it > 100000) needs no weight-gradient kernels
        if ctx.route:
            grads = {} if ctx.need_params else None
            dx = ctx.unet.backward_to_input(ctx.saved, d_out, grads)
            grads = grads or {}
        else:
            dx, grads = ctx.unet.backward_train(ctx.saved, d_out, want_input_grad=ctx.need_dx, want_param_grads=ctx.need_params)
        ctx.saved = None
        return (None, dx, None, None, *[grads.get(n) if ctx.need_params else None for n in ctx.unet.grad_names()])


def unet_train(unet, x_nhwc, precision="fp32", frames=False):
    """Post-fusion U-Net in TRAIN mode (BatchNorm batch statistics, running statistics updated) with gradients for its input
    and every parameter -- the network as the reference trains it until `it > 100000` (train.py:188-197).  precision "bf16":
    bf16 operands (and, for one frame or `frames`, bf16 tensors between the kernels: the half-width chain of csrc/unet_half.inc).
    frames=True: each frame of x is normalised with its own statistics and moves the running statistics once, in frame order --
    x.shape[0] successive one-frame calls in one set of launches."""
    from ._modcache import param_map
    params = param_map(unet)
    return _UnetTrain.apply(unet, x_nhwc, precision, bool(frames), *[params[n] for n in unet.grad_names()])


class _CropResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bbox, size, window_t):
        lib = _abi.load()
        src = x.detach().to(torch.float32).contiguous()
        if src.device.type != "cuda":
            raise _abi.S2LError("crop_resize: input must be on the GPU (no CPU fallback)")
        F_, H, W, _ = src.shape
        bx, by, bx2, by2 = (int(v) for v in bbox[:4])
        oh, ow = int(size[0]), int(size[1])
        out = _f(src.device, F_ // window_t, 3, window_t, oh, ow) if window_t else _f(src.device, F_, oh, ow, 3)
        with torch.cuda.device(src.device):
            _abi.check(lib.s2l_crop_resize(_ptr(src), H, W, bx, by, bx2, by2, _ptr(out), oh, ow, int(window_t), F_, _stream()),
                       "s2l_crop_resize")
        ctx.geom = (F_, H, W, bx, by, bx2, by2, oh, ow, int(window_t))
        return out

    @staticmethod
    def backward(ctx, d_out):
        F_, H, W, bx, by, bx2, by2, oh, ow, t = ctx.geom
        d = d_out.contiguous().float()
        dx = _f(d.device, F_, H, W, 3)
        with torch.cuda.device(d.device):
            _abi.check(_abi.load().s2l_crop_resize_backward(_ptr(d), H, W, bx, by, bx2, by2, _ptr(dx), oh, ow, t, F_, _stream()),
                       "s2l_crop_resize_backward")
        return dx, None, None, None


def crop_resize(x_nhwc, bbox, size=(96, 96), window_t: int = 0):
    """x [F,H,W,3] -> [F,oh,ow,3], or with window_t = T the rgb_window layout [F/T,3,T,oh,ow] (frame f = s*T + t)."""
    return _CropResize.apply(x_nhwc, bbox, size, window_t)


class _SyncLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sync, mel, pos, neg, weight):
        loss, d_pos = sync.get_sync_contrastive_loss(mel, pos, neg, weight=weight, want_grad=True)
        ctx.save_for_backward(d_pos)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        (d_pos,) = ctx.saved_tensors
        return None, None, d_pos * d_loss, None, None


def sync_contrastive_loss(sync, mel, g_rgb_pos, g_rgb_neg, weight: float = 1.0):
    return _SyncLoss.apply(sync, mel, g_rgb_pos, g_rgb_neg, float(weight))


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        lib = _abi.load()
        p = pred.detach().to(torch.float32).contiguous()
        if p.device.type != "cuda":
            raise _abi.S2LError("mse: prediction must be on the GPU (no CPU fallback)")
        t = target.detach().to(torch.float32).contiguous().to(p.device)
        dp, loss, work = torch.empty_like(p), _f(p.device, 1), _f(p.device, 1024)
        with torch.cuda.device(p.device):
            _abi.check(lib.s2l_mse(_ptr(p), _ptr(t), ctypes.c_float(float(weight)), _ptr(dp), _ptr(work), _ptr(loss), p.numel(),
                                   _stream()), "s2l_mse")
        ctx.save_for_backward(dp)
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        (dp,) = ctx.saved_tensors
        return (dp * d_loss).reshape(ctx.shape), None, None


def mse(prediction, target, weights: float = 1.0):
    """mean((prediction - target)^2) * weights (training.py:605-619, mask=None)."""
    return _Mse.apply(prediction, target, weights)


class _LpipsDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, in0, in1, from01, precision=None):
        need = in0.requires_grad
        if in1.requires_grad:
            raise NotImplementedError("LPIPS on the HIP path differentiates with respect to its first image only (the second is the "
                                      "ground truth in training.py:655-674)")
        if need:
            out, ctx.state = module.distance_nhwc(in0, in1, from01, keep=True, precision=precision)
        else:
            out = module.distance_nhwc(in0, in1, from01, precision=precision)
        ctx.module, ctx.shape = module, in0.shape
        return out

    @staticmethod
    def backward(ctx, d_out):
        return None, ctx.module.backward_nhwc(ctx.state, d_out).reshape(ctx.shape), None, None, None


def lpips_distance(module, in0_nhwc, in1_nhwc, from01: bool = False, precision: str = None):
    """lpips.LPIPS(net='alex')(in0, in1) for NHWC images -> [N]; differentiable in in0 (csrc/lpips.hip).  precision: None (the
    module's `conv_precision`), "fp32" or "split" (conv2..conv5 on hi + lo bf16 operands)."""
    return _LpipsDistance.apply(module, in0_nhwc, in1_nhwc, bool(from01), precision)
