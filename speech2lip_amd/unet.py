"""`SimpleUnetLight` -- drop-in for the reference's post-fusion U-Net.

Same module tree and state-dict keys as `src/face_simple/models/SimpleUnetLight.py:16-111`
(`inc.double_conv.{0,1,3,4}`, `down{1,2}.maxpool_conv.1.double_conv.*`, `up{1,2}.conv.double_conv.*`,
`outc.conv.*`), so `post_fusion_unet.*` keys of a reference checkpoint load unchanged.  `forward`
(NCHW, as in the reference) and `forward_nhwc` run the HIP implicit-GEMM path of `csrc/unet.hip`
in eval mode (BatchNorm folded with its running statistics at pack time).  Inference only.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from . import _abi, _modcache
from ._modcache import param_map
from .weights import UNET_CONVS


def _tensor_names():
    names = []
    for name, _, _ in UNET_CONVS:
        head, idx = name.rsplit(".", 1)
        bn = f"{head}.{int(idx) + 1}"
        names += [f"{name}.weight", f"{bn}.weight", f"{bn}.bias", f"{bn}.running_mean", f"{bn}.running_var"]
    return names + ["outc.conv.weight", "outc.conv.bias"]


_TENSOR_NAMES = _tensor_names()      # the C-ABI's order (s2l_unet_pack): per 3x3 layer weight + its BatchNorm's four, then the 1x1 layer


def nhwc_to_c32(x: torch.Tensor) -> torch.Tensor:
    """[F,H,W,C] -> the half-width chain's layout [F,C/32,H,W,32] (32-channel planes; csrc/unet_half.inc).  A test / tool helper:
    the chain itself never converts."""
    F_, H, W, C = x.shape
    return x.reshape(F_, H, W, C // 32, 32).permute(0, 3, 1, 2, 4).contiguous()


def c32_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    F_, P, H, W, _ = x.shape
    return x.permute(0, 2, 3, 1, 4).reshape(F_, H, W, P * 32).contiguous()


def _double_conv(cin, cout, mid=None):
    mid = mid or cout
    seq = nn.Sequential(nn.Conv2d(cin, mid, 3, padding=1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(inplace=True),
                        nn.Conv2d(mid, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
    holder = nn.Module()
    holder.double_conv = seq
    return holder


class SimpleUnetLight(nn.Module):
    def __init__(self, cfg=None, n_channels=3, n_classes=3, bilinear=True):
        super().__init__()
        if n_channels != 3 or n_classes != 3 or not bilinear:
            raise NotImplementedError("HIP path is specialised for the 3 -> 3 channel bilinear U-Net of the May config")
        self.cfg, self.n_channels, self.n_classes, self.bilinear = cfg, n_channels, n_classes, bilinear
        self.inc = _double_conv(3, 64)
        self.down1 = nn.Module()
        self.down1.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _double_conv(64, 128))
        self.down2 = nn.Module()
        self.down2.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _double_conv(128, 128))
        self.up1 = nn.Module()
        self.up1.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.up1.conv = _double_conv(256, 64, 128)
        self.up2 = nn.Module()
        self.up2.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.up2.conv = _double_conv(128, 64, 64)
        self.outc = nn.Module()
        self.outc.conv = nn.Conv2d(64, 3, kernel_size=1)
        self._packed = None
        self._packed_key = None

    def _tensors(self):
        """Weights, BatchNorm parameters and running statistics in the C-ABI's order (looked up through the owning modules' own
        `_parameters` / `_buffers` dictionaries, found once and re-validated link by link: see TalkingFace._hot_tensors)."""
        return _modcache.tensor_slots(self, _TENSOR_NAMES, "_tensor_cache")

    def packed_weights(self) -> torch.Tensor:
        lib = _abi.load()
        tensors = self._tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed is None or key != self._packed_key:
            dev = tensors[0].device
            if dev.type != "cuda":
                raise _abi.S2LError(f"U-Net parameters are on {dev}; the HIP path needs a GPU (no CPU fallback)")
            hold = [t.detach().to(torch.float32).contiguous() for t in tensors]
            table = (ctypes.c_void_p * len(hold))(*[h.data_ptr() for h in hold])
            packed = torch.empty(int(lib.s2l_unet_packed_floats()), dtype=torch.float32, device=dev)
            eps = float(self.inc.double_conv[1].eps)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_unet_pack(table, ctypes.c_float(eps), ctypes.c_void_p(packed.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_unet_pack")
            self._packed, self._packed_key = packed, key
        return self._packed

    def packed_weights_bf16(self) -> torch.Tensor:
        """The nine 3x3 layers in bf16 operand form (forward and input-gradient), for forward_saved_nhwc(precision="bf16")."""
        lib = _abi.load()
        tensors = self._tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if getattr(self, "_packed16", None) is None or key != self._packed16_key:
            dev = tensors[0].device
            if dev.type != "cuda":
                raise _abi.S2LError(f"U-Net parameters are on {dev}; the HIP path needs a GPU (no CPU fallback)")
            hold = [t.detach().to(torch.float32).contiguous() for t in tensors]
            table = (ctypes.c_void_p * len(hold))(*[h.data_ptr() for h in hold])
            packed16 = torch.empty(int(lib.s2l_unet_packed16_halves()), dtype=torch.int16, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_unet_pack16(table, ctypes.c_float(float(self.inc.double_conv[1].eps)), ctypes.c_void_p(packed16.data_ptr()),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_unet_pack16")
                # (no synchronisation: temporaries in `hold` are blocks of the stream-ordered caching allocator)
            self._packed16, self._packed16_key = packed16, key
        return self._packed16

    def packed_weights_split(self) -> torch.Tensor:
        """The nine 3x3 layers in split (hi, lo IEEE halves) operand form, for forward_nhwc(precision="split")."""
        lib = _abi.load()
        tensors = self._tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if getattr(self, "_packed16x3", None) is None or key != self._packed16x3_key:
            dev = tensors[0].device
            if dev.type != "cuda":
                raise _abi.S2LError(f"U-Net parameters are on {dev}; the HIP path needs a GPU (no CPU fallback)")
            hold = [t.detach().to(torch.float32).contiguous() for t in tensors]
            table = (ctypes.c_void_p * len(hold))(*[h.data_ptr() for h in hold])
            blob = torch.empty(int(lib.s2l_unet_packed16x3_halves()), dtype=torch.int16, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_unet_pack16x3(table, ctypes.c_float(float(self.inc.double_conv[1].eps)), ctypes.c_void_p(blob.data_ptr()),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_unet_pack16x3")
                torch.cuda.current_stream().synchronize()      # `hold` may be temporaries
            self._packed16x3, self._packed16x3_key = blob, key
        return self._packed16x3

    def forward_nhwc(self, x: torch.Tensor, out: torch.Tensor = None, precision: str = "fp32") -> torch.Tensor:
        """x [F,H,W,3] -> [F,H,W,3] (the layout the composite produces and the caller wants).  precision:
          "fp32"  (default) exact fp32 MFMA: the parity mode, bit-reproducible against the C++ kernel;
          "split": every operand of the 3x3 convolutions as hi + lo IEEE-half parts, three 16-bit MFMAs per product, fp32
                  accumulation -- ~1e-6 of the output scale from the fp32 result (far inside the north-star's PSNR >= 50 dB /
                  RMSE <= 1e-4) at a multiple of the fp32 rate: the inference speed mode;
          "bf16"  plain bf16 operands (~5e-3 relative error: 45 dB, OUTSIDE the inference tolerance; it exists for the training
                  chain, where BASELINE config 5 names bf16)."""
        lib = _abi.load()
        if precision not in ("fp32", "bf16", "split"):
            raise ValueError("precision must be 'fp32', 'split' or 'bf16'")
        packed = self.packed_weights()
        packed16 = self.packed_weights_bf16() if precision == "bf16" else self.packed_weights_split() if precision == "split" else None
        if x.device.type != "cuda":
            raise _abi.S2LError("U-Net input must be on the GPU (no CPU fallback)")
        if self.training:
            raise NotImplementedError("forward_nhwc is the eval-mode network (running statistics); in train mode call "
                                      "forward_train_nhwc / the module itself")
        x = x.detach().to(torch.float32).contiguous()
        F_, H, W, C = x.shape
        if C != 3 or H < 4 or W < 4:
            raise ValueError(f"U-Net input must be [F,H>=4,W>=4,3], got {tuple(x.shape)}")
        if out is None:
            out = torch.empty(F_, H, W, 3, dtype=torch.float32, device=x.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        # frames go through in groups: every convolution is one launch per group, and a launch needs many waves of workgroups
        # (512 fit at a time) to amortise its ramp and tail.  The activation workspace of a group is capped at 16 GiB and at
        # half of the memory that is free right now (training state or a smaller device may have claimed the rest)
        per_frame = int(lib.s2l_unet_work_floats(H, W, 1))
        free_bytes, _ = torch.cuda.mem_get_info(x.device)
        free_bytes += torch.cuda.memory_reserved(x.device) - torch.cuda.memory_allocated(x.device)     # torch's own cached blocks
        budget_floats = max(per_frame, min(1 << 32, free_bytes // 8))
        group = max(1, min(F_, budget_floats // max(per_frame, 1)))
        work = torch.empty(per_frame * group, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            for s in range(0, F_, group):
                n = min(group, F_ - s)
                fwd = lib.s2l_unet_forward_split if precision == "split" else lib.s2l_unet_forward
                _abi.check(fwd(ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(0 if packed16 is None else packed16.data_ptr()),
                               ctypes.c_void_p(x[s:].data_ptr()), ctypes.c_void_p(work.data_ptr()), ctypes.c_void_p(out[s:].data_ptr()),
                               H, W, n, st), "s2l_unet_forward")
        return out

    # ------------------------------------------------------------------ train mode (BatchNorm batch statistics)
    def _table(self, tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def grad_names(self):
        """state-dict names of the tensors whose gradients s2l_unet_train_backward returns, in its order."""
        names = []
        for name, _, _ in UNET_CONVS:
            head, idx = name.rsplit(".", 1)
            bn = f"{head}.{int(idx) + 1}"
            names += [f"{name}.weight", f"{bn}.weight", f"{bn}.bias"]
        return names + ["outc.conv.weight", "outc.conv.bias"]

    def _raw_blobs(self, tensors, table, want16: bool):
        """The un-folded weights in the kernels' layouts (fp32: s2l_unet_pack_raw; bf16: s2l_unet_pack16 with bn_eps < 0), cached on
        the versions of the WEIGHT tensors alone: the running statistics the train-mode forward rewrites are not in these blobs,
        so a frozen net packs once and a training net once per optimizer step."""
        lib = _abi.load()
        dev = tensors[0].device
        key = tuple((t.data_ptr(), t._version) for i, t in enumerate(tensors) if i % 5 == 0 or i >= 50)
        if getattr(self, "_raw_key", None) != key:
            self._raw = self._raw16 = None
            self._raw_key = key
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if self._raw is None:
                raw = torch.empty(int(lib.s2l_unet_packed_floats()), dtype=torch.float32, device=dev)
                _abi.check(lib.s2l_unet_pack_raw(table, p(raw), st), "s2l_unet_pack_raw")
                self._raw = raw
            if want16 and self._raw16 is None:
                raw16 = torch.empty(int(lib.s2l_unet_packed16_halves()), dtype=torch.int16, device=dev)
                _abi.check(lib.s2l_unet_pack16(table, ctypes.c_float(-1.0), p(raw16), st), "s2l_unet_pack16")
                self._raw16 = raw16
        return self._raw, (self._raw16 if want16 else None)

    def forward_train_nhwc(self, x: torch.Tensor, update_running: bool = True, precision: str = "fp32"):
        """The network in TRAIN mode (the reference until it > 100000, train.py:188-197): x [F,H,W,3] -> (out, ctx).  BatchNorm
        uses the statistics of the batch; with update_running the running_mean / running_var buffers are updated in place
        and num_batches_tracked is incremented, as nn.BatchNorm2d does.  ctx feeds backward_train.
        precision "bf16": the 3x3 convolutions here and their input gradients in backward_train take bf16 operands (fp32
        accumulation, statistics and tensors), as forward_saved_nhwc(precision="bf16") does for the eval-mode net."""
        lib = _abi.load()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if x.device.type != "cuda":
            raise _abi.S2LError("U-Net input must be on the GPU (no CPU fallback)")
        tensors = self._tensors()
        dev = tensors[0].device
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise _abi.S2LError("U-Net parameters and buffers must be contiguous fp32 tensors on one GPU")
        x = x.detach().to(torch.float32).contiguous()
        F_, H, W, C = x.shape
        if C != 3 or H < 4 or W < 4:
            raise ValueError(f"U-Net input must be [F,H>=4,W>=4,3], got {tuple(x.shape)}")
        table = self._table(tensors)
        raw, raw16 = self._raw_blobs(tensors, table, precision == "bf16")
        out = torch.empty(F_, H, W, 3, dtype=torch.float32, device=dev)
        saved = torch.empty(int(lib.s2l_unet_train_saved_floats(H, W, F_)), dtype=torch.float32, device=dev)
        scratch = torch.empty(262144, dtype=torch.float32, device=dev)
        bn = self.inc.double_conv[1]
        momentum = 0.1 if bn.momentum is None else float(bn.momentum)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if raw16 is None:
                _abi.check(lib.s2l_unet_train_forward(p(raw), table, ctypes.c_float(float(bn.eps)), ctypes.c_float(momentum),
                                                      1 if update_running else 0, p(x), p(saved), p(scratch), p(out), H, W, F_, st),
                           "s2l_unet_train_forward")
            else:
                _abi.check(lib.s2l_unet_train_forward_bf16(p(raw), p(raw16), table, ctypes.c_float(float(bn.eps)),
                                                           ctypes.c_float(momentum), 1 if update_running else 0, p(x), p(saved),
                                                           p(scratch), p(out), H, W, F_, st), "s2l_unet_train_forward_bf16")
        if update_running:
            # (one fused launch for the ten counters: as ten `+= 1` they were 400 launches per sync step in train-mode BatchNorm)
            torch._foreach_add_([mod.num_batches_tracked for mod in self.modules() if isinstance(mod, nn.BatchNorm2d)], 1)
            # the folded eval-mode blobs are stale now: the kernel rewrote the running statistics in place, which does not bump
            # the tensors' version counters (the cache keys), and BOTH the fp32 and the bf16 blob fold them
            self._packed = self._packed_key = None
            self._packed16 = self._packed16_key = None
            self._packed16x3 = self._packed16x3_key = None
        return out, (raw, x, saved, (F_, H, W), raw16)

    def backward_train(self, ctx, d_out: torch.Tensor, want_input_grad: bool = True, want_param_grads: bool = True):
        """d loss / d out [F,H,W,3] -> (d loss / d x or None, {state-dict name: gradient}) for every conv / BatchNorm / outc
        parameter (what loss.backward() leaves in .grad while the post-fusion net trains).  want_param_grads=False (a frozen
        net in train-mode BatchNorm): the weight-gradient kernels are not launched and the dict is empty."""
        if not (want_input_grad or want_param_grads):
            raise ValueError("backward_train: nothing to compute")
        lib = _abi.load()
        raw, x, saved, (F_, H, W), raw16 = ctx
        dev = x.device
        d = d_out.detach().to(torch.float32).contiguous()
        if d.shape != (F_, H, W, 3) or d.device != dev:
            raise ValueError(f"d_out must be [{F_},{H},{W},3] on {dev}")
        tensors = self._tensors()
        table = self._table(tensors)
        dx = torch.empty_like(d) if want_input_grad else None
        flat = torch.empty(int(lib.s2l_unet_grad_floats()), dtype=torch.float32, device=dev) if want_param_grads else None
        work = torch.empty(int(lib.s2l_unet_train_work_floats(H, W, F_)), dtype=torch.float32, device=dev)
        p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if raw16 is None:
                _abi.check(lib.s2l_unet_train_backward(p(raw), table, p(x), p(saved), p(d), p(work), p(dx), p(flat), H, W, F_, st),
                           "s2l_unet_train_backward")
            else:
                _abi.check(lib.s2l_unet_train_backward_bf16(p(raw), p(raw16), table, p(x), p(saved), p(d), p(work), p(dx), p(flat),
                                                            H, W, F_, st), "s2l_unet_train_backward_bf16")
        if not want_param_grads:
            return dx, {}
        params = param_map(self)
        grads, off = {}, 0
        for name in self.grad_names():
            n = params[name].numel()
            grads[name] = flat[off:off + n].reshape(params[name].shape)
            off += n
        return dx, grads

    def forward_train_frames_nhwc(self, x: torch.Tensor, update_running: bool = True, precision: str = "fp32", fuse_norm=None):
        """x [F,H,W,3] -> (out, ctx): F successive ONE-FRAME train-mode calls in one set of launches (s2l_unet_train_forward_frames):
        each frame normalised with its own batch statistics, the running statistics moved once per frame in frame order,
        num_batches_tracked += F -- what the reference's loop does to the (frozen) net, bit for bit what F calls of
        forward_train_nhwc(x[f:f+1]) compute.  ctx feeds backward_train_frames (input gradient only).
        precision "bf16h": bf16 operands AND bf16 tensors between the kernels (s2l_unet_train_forward_frames_h: half the memory
        traffic of "bf16"; fp32 accumulation and statistics; x / out / gradients at the boundary stay fp32).
        fuse_norm ("bf16h" only; default: on for a FROZEN net, i.e. when no parameter requires a gradient): the activations whose only
        reader is the next convolution at the same resolution (a0, a2, a4, a6, a8) or the up-sampling (a5, a7) are never stored -- the
        reader normalises its input itself (s2l_unet_train_forward_frames_h_fused: the same bits; `bn_relu_h_kernel` leaves the chain).
        The state then serves the input gradient only: `backward_train_frames(..., want_param_grads=True)` refuses it."""
        lib = _abi.load()
        if precision not in ("fp32", "bf16", "bf16h"):
            raise ValueError("precision must be 'fp32', 'bf16' or 'bf16h'")
        if x.device.type != "cuda":
            raise _abi.S2LError("U-Net input must be on the GPU (no CPU fallback)")
        tensors = self._tensors()
        dev = tensors[0].device
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise _abi.S2LError("U-Net parameters and buffers must be contiguous fp32 tensors on one GPU")
        x = x.detach().to(torch.float32).contiguous()
        F_, H, W, C = x.shape
        if C != 3 or H < 4 or W < 4 or F_ < 1:
            raise ValueError(f"U-Net input must be [F>=1,H>=4,W>=4,3], got {tuple(x.shape)}")
        table = self._table(tensors)
        half = precision == "bf16h"
        if fuse_norm is None:      # (S2L_NO_FUSE_NORM=1: the A/B switch of tools/bench_train.py)
            fuse_norm = half and not any(t.requires_grad for t in tensors) and not os.environ.get("S2L_NO_FUSE_NORM")
        if fuse_norm and not half:
            raise ValueError("fuse_norm belongs to the half-width chain (precision 'bf16h')")
        raw, raw16 = self._raw_blobs(tensors, table, precision != "fp32")
        out = torch.empty(F_, H, W, 3, dtype=torch.float32, device=dev)
        if half:
            saved = torch.empty(int(lib.s2l_unet_train_frames_h_saved_halves(H, W, F_)), dtype=torch.int16, device=dev)
            scratch = torch.empty(int(lib.s2l_unet_train_frames_h_scratch_floats(F_)), dtype=torch.float32, device=dev)
        else:
            saved = torch.empty(int(lib.s2l_unet_train_frames_saved_floats(H, W, F_)), dtype=torch.float32, device=dev)
            scratch = torch.empty(int(lib.s2l_unet_train_frames_scratch_floats(F_)), dtype=torch.float32, device=dev)
        bn = self.inc.double_conv[1]
        momentum = 0.1 if bn.momentum is None else float(bn.momentum)
        p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if half:
                entry = lib.s2l_unet_train_forward_frames_h_fused if fuse_norm else lib.s2l_unet_train_forward_frames_h
                _abi.check(entry(p(raw), p(raw16), table, ctypes.c_float(float(bn.eps)), ctypes.c_float(momentum),
                                 1 if update_running else 0, p(x), p(saved), p(scratch), p(out), H, W, F_, st),
                           "s2l_unet_train_forward_frames_h" + ("_fused" if fuse_norm else ""))
            else:
                _abi.check(lib.s2l_unet_train_forward_frames(p(raw), p(raw16), table, ctypes.c_float(float(bn.eps)), ctypes.c_float(momentum),
                                                             1 if update_running else 0, p(x), p(saved), p(scratch), p(out), H, W, F_, st),
                           "s2l_unet_train_forward_frames")
        if update_running:
            torch._foreach_add_([mod.num_batches_tracked for mod in self.modules() if isinstance(mod, nn.BatchNorm2d)], F_)
            self._packed = self._packed_key = None
            self._packed16 = self._packed16_key = None
            self._packed16x3 = self._packed16x3_key = None
        return out, (raw, x, saved, (F_, H, W), raw16, bool(fuse_norm))

    def backward_train_frames(self, ctx, d_out: torch.Tensor, want_param_grads: bool = False):
        """d loss / d x [F,H,W,3] for a forward_train_frames_nhwc state.  want_param_grads (a net that still trains; fp32 tensors only):
        returns (d_x, {state-dict name: gradient}) -- the parameter gradients of the F one-frame calls, summed."""
        lib = _abi.load()
        raw, x, saved, (F_, H, W), raw16 = ctx[:5]
        if want_param_grads and len(ctx) > 5 and ctx[5]:
            raise _abi.S2LError("this forward state was made with fuse_norm (a frozen net's: the activations the weight gradients read were "
                                "never stored); run forward_train_frames_nhwc(..., fuse_norm=False) for a net that trains")
        dev = x.device
        d = d_out.detach().to(torch.float32).contiguous()
        if d.shape != (F_, H, W, 3) or d.device != dev:
            raise ValueError(f"d_out must be [{F_},{H},{W},3] on {dev}")
        tensors = self._tensors()
        table = self._table(tensors)
        dx = torch.empty_like(d)
        p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        def named(flat):
            params = param_map(self)
            grads, off = {}, 0
            for name in self.grad_names():
                n = params[name].numel()
                grads[name] = flat[off:off + n].reshape(params[name].shape)
                off += n
            return grads
        if saved.dtype == torch.int16:      # the half-width chain (precision "bf16h")
            if want_param_grads:
                work = torch.empty(int(lib.s2l_unet_train_frames_h_work_halves_grads(H, W, F_)), dtype=torch.int16, device=dev)
                flat = torch.empty(int(lib.s2l_unet_grad_floats()), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    _abi.check(lib.s2l_unet_train_backward_frames_h_grads(p(raw), p(raw16), table, p(x), p(saved), p(d), p(work), p(dx), p(flat),
                                                                          H, W, F_, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                               "s2l_unet_train_backward_frames_h_grads")
                return dx, named(flat)
            work = torch.empty(int(lib.s2l_unet_train_frames_h_work_halves(H, W, F_)), dtype=torch.int16, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_unet_train_backward_frames_h(p(raw), p(raw16), table, p(saved), p(d), p(work), p(dx), H, W, F_,
                                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                           "s2l_unet_train_backward_frames_h")
            return dx
        work = torch.empty(int(lib.s2l_unet_train_frames_work_floats(H, W, F_)), dtype=torch.float32, device=dev)
        if want_param_grads:
            flat = torch.empty(int(lib.s2l_unet_grad_floats()), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_unet_train_backward_frames_grads(p(raw), p(raw16), table, p(x), p(saved), p(d), p(work), p(dx), p(flat), H, W, F_,
                                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                           "s2l_unet_train_backward_frames_grads")
            return dx, named(flat)
        with torch.cuda.device(dev):
            _abi.check(lib.s2l_unet_train_backward_frames(p(raw), p(raw16), table, p(x), p(saved), p(d), p(work), p(dx), H, W, F_,
                                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "s2l_unet_train_backward_frames")
        return dx

    def forward_saved_nhwc(self, x: torch.Tensor, window=None, precision: str = "fp32"):
        """Training-time forward of the frozen eval-mode network: x [F,H,W,3] -> (out [F,H,W,3], saved), where `saved` holds
        every activation `backward_input` needs (504 MB per 500x500 frame).  The caller bounds F.
        window = (full_h, full_w, origin_y, origin_x): x is a crop of a full frame (s2l_unet_forward_saved_window); values
        within 32 pixels of a crop edge that is not a frame edge are not the full-frame values.
        precision "bf16": the 3x3 convolutions (and their input gradients in backward_input) take bf16 operands on the
        32x32x16 MFMA -- fp32 accumulation, fp32 tensors in memory -- the precision BASELINE config 5 names for training.
        precision "bf16h": the same operands AND bf16 tensors between the kernels (s2l_unet_forward_saved_h: 32-channel planes,
        half the memory traffic; x / out / the gradients at the boundary stay fp32)."""
        lib = _abi.load()
        if precision not in ("fp32", "bf16", "bf16h"):
            raise ValueError("precision must be 'fp32', 'bf16' or 'bf16h'")
        packed = self.packed_weights()
        packed16 = self.packed_weights_bf16() if precision != "fp32" else None
        if x.device.type != "cuda":
            raise _abi.S2LError("U-Net input must be on the GPU (no CPU fallback)")
        if self.training:
            raise NotImplementedError("forward_saved_nhwc is the frozen eval-mode network; in train mode use forward_train_nhwc")
        x = x.detach().to(torch.float32).contiguous()
        F_, H, W, C = x.shape
        if C != 3 or H < 4 or W < 4:
            raise ValueError(f"U-Net input must be [F,H>=4,W>=4,3], got {tuple(x.shape)}")
        out = torch.empty(F_, H, W, 3, dtype=torch.float32, device=x.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        p16 = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        fh, fw, oy, ox = (H, W, 0, 0) if window is None else (int(v) for v in window)
        if precision == "bf16h":
            saved = torch.empty(int(lib.s2l_unet_saved_h_halves(H, W, F_)), dtype=torch.int16, device=x.device)
            with torch.cuda.device(x.device):
                _abi.check(lib.s2l_unet_forward_saved_h(p(packed), p(packed16), p(x), p(saved), p(out), H, W, fh, fw, oy, ox, F_,
                                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_unet_forward_saved_h")
            return out, (saved, packed, (F_, H, W), (fh, fw, oy, ox), packed16)
        saved = torch.empty(int(lib.s2l_unet_saved_floats(H, W, F_)), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _abi.check(lib.s2l_unet_forward_saved_window(p(packed), p16(packed16), p(x), p(saved), p(out), H, W, fh, fw, oy, ox, F_,
                                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "s2l_unet_forward_saved_window")
        return out, (saved, packed, (F_, H, W), (fh, fw, oy, ox), packed16)

    def backward_input(self, saved_ctx, d_out: torch.Tensor) -> torch.Tensor:
        """d loss / d x [F,H,W,3] from d loss / d out, through the frozen network (what autograd propagates once the
        post-fusion net is fixed, train.py:188-197)."""
        lib = _abi.load()
        saved, packed, (F_, H, W), (fh, fw, oy, ox), packed16 = saved_ctx
        d = d_out.detach().to(torch.float32).contiguous()
        if d.shape != (F_, H, W, 3) or d.device != saved.device:
            raise ValueError(f"d_out must be [{F_},{H},{W},3] on {saved.device}")
        dx = torch.empty_like(d)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        if saved.dtype == torch.int16:      # the half-width chain (precision "bf16h")
            work = torch.empty(int(lib.s2l_unet_backward_h_work_halves(H, W, F_)), dtype=torch.int16, device=d.device)
            with torch.cuda.device(d.device):
                _abi.check(lib.s2l_unet_backward_h(p(packed), p(packed16), p(saved), p(d), p(work), p(dx), H, W, fh, fw, oy, ox, F_,
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_unet_backward_h")
            return dx
        work = torch.empty(int(lib.s2l_unet_backward_work_floats(H, W, F_)), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _abi.check(lib.s2l_unet_backward_window(p(packed), ctypes.c_void_p(0 if packed16 is None else packed16.data_ptr()), p(saved), p(d),
                                                    p(work), p(dx), H, W, fh, fw, oy, ox, F_,
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s2l_unet_backward_window")
        return dx

    # ------------------------------------------------------------------ mode-following pair for the fused training steps
    def forward_for_backward(self, x: torch.Tensor, window=None, precision: str = "fp32"):
        """What `post_fusion_unet(x)` computes in the module's CURRENT mode, with the state its backward needs:
        eval mode -> forward_saved_nhwc (running statistics; `window` / bf16 operands allowed);
        train mode -> BatchNorm batch statistics, ONE FRAME PER CALL in frame order -- the reference only ever hands the net one
        frame at a time (tf_nerf.py:387 inside train_stage1's batch-1 calls), so each frame is normalised with its own
        statistics and the running statistics move once per frame, in that order.  This is also the mode of a FROZEN net inside
        the reference's loop after it > 100000: Trainer.train_step's self.model.train() (training.py:150) undoes the .eval() of
        train.py:195.  A crop is not equivalent then (statistics are over the whole frame); `precision` as in eval mode."""
        if not self.training:
            if precision == "bf16" and getattr(self, "half_width_tensors", True) and x.shape[1] <= 255 * 32 and x.shape[2] <= 255 * 16:
                precision = "bf16h"      # (bf16 tensors between the kernels too: the half-width chain, csrc/unet_half.inc)
            return self.forward_saved_nhwc(x, window=window, precision=precision)
        if window is not None:
            raise ValueError("train-mode BatchNorm needs whole frames: statistics are taken over the full image")
        if not any(p.requires_grad for p in self.parameters()):
            # the frozen net of the loop after it > 100000: the frames go through in groups of whole frames, every frame still its own
            # statistics group (s2l_unet_train_forward_frames: the same bits as one call per frame, ~1/F of the launches)
            # precision "bf16" here also means bf16 TENSORS between the kernels (the half-width chain, csrc/unet_half.inc: the same
            # operands, half the memory traffic) unless `half_width_tensors` is set to False on the module
            F_, H, W = x.shape[0], x.shape[1], x.shape[2]
            lib = _abi.load()
            if precision == "bf16" and getattr(self, "half_width_tensors", True) and H <= 255 * 32 and W <= 255 * 16:
                precision = "bf16h"
                per_frame = 2 * (int(lib.s2l_unet_train_frames_h_saved_halves(H, W, 1)) + int(lib.s2l_unet_train_frames_h_work_halves(H, W, 1)))
            else:
                per_frame = 4 * (int(lib.s2l_unet_train_frames_saved_floats(H, W, 1)) + int(lib.s2l_unet_train_frames_work_floats(H, W, 1)))
            group = max(1, min(F_, int(getattr(self, "train_frames_budget_bytes", 16 << 30)) // max(per_frame, 1)))
            if getattr(self, "train_frames_per_group", None):
                group = max(1, min(F_, int(self.train_frames_per_group)))
            if precision == "bf16h":
                group = min(group, 8191)      # the half-width entry points take at most 8191 frames per call (S2L_E_SIZE beyond)
            outs, ctxs = [], []
            for s0 in range(0, F_, group):
                o, c = self.forward_train_frames_nhwc(x[s0:s0 + group], update_running=True, precision=precision)
                outs.append(o)
                ctxs.append((s0, min(F_, s0 + group), c))
            return torch.cat(outs, 0), ("train_frames", ctxs)
        # a net that still trains (it <= 100000): the same frames route with fp32 tensors (every frame its own statistics group,
        # running statistics in frame order), its backward also returns the parameter gradients summed over the frames
        # (`batch_train_frames = False` on the module: one forward_train_nhwc / backward_train call per frame instead)
        if getattr(self, "batch_train_frames", True):
            F_, H, W = x.shape[0], x.shape[1], x.shape[2]
            lib = _abi.load()
            if precision == "bf16" and getattr(self, "half_width_tensors", True) and H <= 255 * 32 and W <= 255 * 16:
                precision = "bf16h"      # (bf16 tensors between the kernels, weight gradients straight from the bf16 planes)
            budget = int(getattr(self, "train_frames_budget_bytes", 16 << 30))
            if precision == "bf16h":      # the half-width route's own sizes; its backward also allocates the split-K partials of the weight
                # gradients once per call (the difference of the two work sizes at one frame: ~300 MB), which comes off the budget first
                fixed = 2 * (int(lib.s2l_unet_train_frames_h_work_halves_grads(H, W, 1)) - int(lib.s2l_unet_train_frames_h_work_halves(H, W, 1)))
                per_frame = 2 * (int(lib.s2l_unet_train_frames_h_saved_halves(H, W, 1)) + int(lib.s2l_unet_train_frames_h_work_halves(H, W, 1)))
                budget = max(budget - fixed, per_frame)
            else:
                per_frame = 4 * (int(lib.s2l_unet_train_frames_saved_floats(H, W, 1)) + int(lib.s2l_unet_train_frames_work_floats(H, W, 1)))
            group = max(1, min(F_, budget // max(per_frame, 1)))
            if precision == "bf16h":
                group = min(group, 8191)
            outs, ctxs = [], []
            for s0 in range(0, F_, group):
                o, c = self.forward_train_frames_nhwc(x[s0:s0 + group], update_running=True, precision=precision)
                outs.append(o)
                ctxs.append((s0, min(F_, s0 + group), c))
            return (outs[0] if len(outs) == 1 else torch.cat(outs, 0)), ("train_frames_grads", ctxs)
        outs, ctxs = [], []
        for f in range(x.shape[0]):
            o, c = self.forward_train_nhwc(x[f:f + 1], update_running=True, precision=precision)
            outs.append(o)
            ctxs.append(c)
        return torch.cat(outs, 0), ("train", ctxs)

    def backward_to_input(self, ctx, d_out: torch.Tensor, param_grads: dict = None) -> torch.Tensor:
        """d loss / d x for a `forward_for_backward` state.  Train mode: the BatchNorm backward carries the batch-statistics terms
        whether or not the parameters are frozen; when `param_grads` is a dict the parameter gradients are ACCUMULATED into it
        under their state-dict names (the net while it still trains, it <= 100000)."""
        if isinstance(ctx, tuple) and len(ctx) == 2 and ctx[0] == "train_frames":
            return torch.cat([self.backward_train_frames(c, d_out[s0:s1]) for s0, s1, c in ctx[1]], 0)
        if isinstance(ctx, tuple) and len(ctx) == 2 and ctx[0] == "train_frames_grads":
            if param_grads is None and any(p_.requires_grad for p_ in self.parameters()):
                raise ValueError("backward_to_input: this train-mode U-Net still has trainable parameters; pass param_grads={} to "
                                 "receive their gradients (or freeze the net as train.py:188-197 does)")
            dxs = []
            for s0, s1, c in ctx[1]:
                if param_grads is None:
                    dxs.append(self.backward_train_frames(c, d_out[s0:s1]))
                    continue
                dx, grads = self.backward_train_frames(c, d_out[s0:s1], want_param_grads=True)
                dxs.append(dx)
                for k, v in grads.items():      # (views of this call's own flat gradient buffer: no copy needed to keep them)
                    param_grads[k] = v if k not in param_grads else param_grads[k] + v
            return dxs[0] if len(dxs) == 1 else torch.cat(dxs, 0)
        if not (isinstance(ctx, tuple) and len(ctx) == 2 and ctx[0] == "train"):
            return self.backward_input(ctx, d_out)
        if param_grads is None and any(p_.requires_grad for p_ in self.parameters()):
            raise ValueError("backward_to_input: this train-mode U-Net still has trainable parameters; pass param_grads={} to "
                             "receive their gradients (or freeze the net as train.py:188-197 does)")
        dxs = []
        for f, c in enumerate(ctx[1]):
            dx, grads = self.backward_train(c, d_out[f:f + 1], want_input_grad=True, want_param_grads=param_grads is not None)
            dxs.append(dx)
            if param_grads is not None:
                for k, v in grads.items():
                    param_grads[k] = v.clone() if k not in param_grads else param_grads[k] + v
        return torch.cat(dxs, 0)

    def forward(self, x, x_level1=None, x_level2=None):
        """NCHW in, NCHW out, as the reference's forward (SimpleUnetLight.py:99-111).  With autograd recording and an input
        that requires grad, the input gradient is available to loss.backward() (speech2lip_amd.autograd)."""
        if self.training:        # BatchNorm batch statistics + gradients for every parameter
            from .autograd import unet_train
            return unet_train(self, x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        if torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
            from .autograd import unet_eval
            return unet_eval(self, x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return self.forward_nhwc(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
