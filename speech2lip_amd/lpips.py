"""`LPIPS(net='alex', version='0.1')` on the HIP path -- the perceptual term of the stage-1 loss.

The reference builds `lpips.LPIPS(net='alex', version='0.1', model_path='models/lpips_weights_v0.1/alex.pth')`
(src/face_simple/training.py:76) from the third-party package `lpips==0.1.4` (requirement.txt:11) and calls it on NCHW images
in [-1, 1] (training.py:655-674).  This module has the package's module tree and state-dict keys -- `scaling_layer.{shift,scale}`,
`net.slice1.0`, `net.slice2.3`, `net.slice3.6`, `net.slice4.8`, `net.slice5.10` (torchvision's AlexNet feature indices),
`lin0..lin4.model.1.weight` and their `lins.{i}` aliases -- so `state_dict()` of a real `lpips.LPIPS(net='alex')` loads unchanged.
Forward and the gradient with respect to the first image run in `csrc/lpips.hip`; the net is frozen, as in the package
(`requires_grad=False`); there is no CPU fallback.  The package's weights are not in the reference repository (AlexNet comes
from torchvision's model zoo, the linear heads from the package's own `alex.pth`): parity is structural, on seeded weights.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from . import _abi

ALEX_CONVS = (("slice1", "0", 3, 64, 11, 4, 2), ("slice2", "3", 64, 192, 5, 1, 2), ("slice3", "6", 192, 384, 3, 1, 1),
              ("slice4", "8", 384, 256, 3, 1, 1), ("slice5", "10", 256, 256, 3, 1, 1))
ALEX_CHNS = (64, 192, 384, 256, 256)
MIN_SIZE = 31     # 11x11/4 convolution, then two 3x3/2 poolings


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])


class NetLinLayer(nn.Module):
    """A single linear layer which does a 1x1 conv (parameter holder; Dropout sits at index 0 as in the package)."""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class alexnet(nn.Module):
    """Parameter holder with the slices of lpips/pretrained_networks.py (module names = torchvision's feature indices)."""

    def __init__(self):
        super().__init__()
        for name, idx, cin, cout, k, s, p in ALEX_CONVS:
            seq = nn.Sequential()
            seq.add_module(idx, nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=p))
            setattr(self, name, seq)
        self.N_slices = 5


class LPIPS(nn.Module):
    def __init__(self, pretrained=True, net="alex", version="0.1", lpips=True, spatial=False, pnet_rand=False, pnet_tune=False,
                 use_dropout=True, model_path=None, eval_mode=True, verbose=False, trunk_path=None):
        super().__init__()
        if net not in ("alex", "alexnet") or version != "0.1" or not lpips or spatial or pnet_tune:
            raise NotImplementedError("the HIP path implements LPIPS(net='alex', version='0.1', lpips=True, spatial=False) with a "
                                      "frozen trunk: the configuration of training.py:76")
        self.pnet_type, self.version, self.chns, self.L = net, version, list(ALEX_CHNS), 5
        self.scaling_layer = ScalingLayer()
        self.net = alexnet()
        for i, c in enumerate(ALEX_CHNS):
            setattr(self, f"lin{i}", NetLinLayer(c, use_dropout=use_dropout))
        self.lins = nn.ModuleList([getattr(self, f"lin{i}") for i in range(5)])
        for p in self.parameters():
            p.requires_grad = False
        # What `lpips.LPIPS(pretrained=True, net='alex')` loads: (i) the AlexNet trunk = torchvision's ImageNet weights
        # (`tv.alexnet(pretrained=True).features`, lpips/pretrained_networks.py), (ii) the linear heads from `model_path` (the
        # package's weights/v0.1/alex.pth when None).  Neither file is in the reference repository or this image, and there is no
        # network: `trunk_path` (or $S2L_ALEXNET_WEIGHTS) names a torchvision AlexNet state dict (`features.N.weight/bias`), and a
        # perceptual loss on random features is never silent -- `weights_loaded` records what arrived, and a pretrained=True module
        # that is missing either part warns loudly here and raises at its first use unless `allow_random_weights` is set.
        self.weights_loaded = {"trunk": False, "lins": False}
        self.allow_random_weights = not pretrained
        if pretrained:
            trunk_path = trunk_path or os.environ.get("S2L_ALEXNET_WEIGHTS")
            if trunk_path is not None:
                self.load_trunk(torch.load(trunk_path, map_location="cpu"))
            if model_path is not None and os.path.exists(model_path):
                self.load_lins(torch.load(model_path, map_location="cpu"))
            missing = [k for k, v in self.weights_loaded.items() if not v]
            if missing:
                import warnings
                warnings.warn(f"LPIPS(pretrained=True): no weights loaded for {missing} (model_path={model_path!r}, trunk_path="
                              f"{trunk_path!r}); the module holds RANDOM values there and will refuse to run until load_trunk / "
                              "load_lins / load_state_dict supplies them or allow_random_weights is set", RuntimeWarning, stacklevel=2)
        self._packed = self._packed_key = self._work = None
        # what `precision=None` means in distance_nhwc.  "fp32": the exact-fp32 convolutions (default; what the parity tests pin).
        # "split": conv2..conv5 and their input gradients with operands as hi + lo bf16 parts (csrc/conv_gemm.h: ~1e-5 relative) --
        # the bf16-precision training steps ask for it per call.
        self.conv_precision = "fp32"
        self.eval()

    def train(self, mode: bool = True):
        return super().train(False)       # frozen expert: dropout never active, as eval_mode=True in the package

    def load_trunk(self, state):
        """torchvision AlexNet weights -> net.slice*: accepts `alexnet().state_dict()` (`features.0.weight` ...), its `.features`
        (`0.weight` ...), or this module's own `net.slice1.0.weight` names.  Every one of the five convolutions must be present."""
        state = state.get("state_dict", state) if isinstance(state, dict) else state
        own = {}
        for name, idx, *_ in ALEX_CONVS:
            for kind in ("weight", "bias"):
                for cand in (f"net.{name}.{idx}.{kind}", f"features.{idx}.{kind}", f"{idx}.{kind}"):
                    if cand in state:
                        own[f"net.{name}.{idx}.{kind}"] = state[cand]
                        break
                else:
                    raise KeyError(f"AlexNet trunk weights: no entry for features.{idx}.{kind}")
        res = self.load_state_dict(own, strict=False)
        assert not res.unexpected_keys
        self.weights_loaded["trunk"] = True
        return self

    def load_lins(self, state):
        """The package's weights/v0.1/alex.pth: `lin{i}.model.1.weight` for i = 0..4 (strict: all five, nothing else but their
        `lins.{i}` aliases)."""
        want = {f"lin{i}.model.1.weight" for i in range(5)}
        keys = {k for k in state if not k.startswith("lins.")}
        if keys != want:
            raise KeyError(f"LPIPS linear heads: expected exactly {sorted(want)}, got {sorted(keys)}")
        use_dropout = isinstance(self.lin0.model[0], nn.Dropout)
        own = {(k if use_dropout else k.replace("model.1", "model.0")): v for k, v in state.items() if k in want}
        res = self.load_state_dict(own, strict=False)
        assert not res.unexpected_keys
        self.weights_loaded["lins"] = True
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        if any(k.startswith("net.") for k in state_dict):
            self.weights_loaded["trunk"] = all(f"net.{n}.{i}.weight" in state_dict for n, i, *_ in ALEX_CONVS)
        if any(k.startswith("lin") for k in state_dict):
            self.weights_loaded["lins"] = self.weights_loaded["lins"] or all(
                any(k.startswith(f"lin{i}.") or k.startswith(f"lins.{i}.") for k in state_dict) for i in range(5))
        return res

    def _require_weights(self):
        if not self.allow_random_weights and not all(self.weights_loaded.values()):
            missing = [k for k, v in self.weights_loaded.items() if not v]
            raise _abi.S2LError(f"LPIPS(pretrained=True) has no weights for {missing}: a perceptual loss on random features would train "
                                "silently against noise.  Supply them (load_trunk / load_lins / load_state_dict, trunk_path= or "
                                "$S2L_ALEXNET_WEIGHTS) or set allow_random_weights = True for structural tests.")

    def _tensors(self):
        t = []
        for name, idx, *_ in ALEX_CONVS:
            conv = getattr(self.net, name)._modules[idx]
            t += [conv.weight, conv.bias]
        t += [getattr(self, f"lin{i}").model[-1].weight for i in range(5)]
        return t + [self.scaling_layer.shift, self.scaling_layer.scale]

    def packed_weights(self) -> torch.Tensor:
        lib = _abi.load()
        tensors = self._tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed is None or key != self._packed_key:
            dev = tensors[0].device
            if dev.type != "cuda":
                raise _abi.S2LError(f"LPIPS parameters are on {dev}; the HIP path needs a GPU (no CPU fallback)")
            hold = [t.detach().to(torch.float32).contiguous() for t in tensors]
            table = (ctypes.c_void_p * len(hold))(*[h.data_ptr() for h in hold])
            packed = torch.empty(int(lib.s2l_lpips_packed_floats()), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.s2l_lpips_pack(table, _p(packed), _st()), "s2l_lpips_pack")
                torch.cuda.current_stream().synchronize()      # `hold` may be temporaries
            self._packed, self._packed_key = packed, key
        return self._packed

    # -- the raw entry points (NHWC) ------------------------------------------------------------------------------------
    def distance_nhwc(self, in0: torch.Tensor, in1: torch.Tensor, from01: bool = False, keep: bool = False, precision: str = None):
        """in0, in1 [N,H,W,3] in [-1,1] (or [0,1] with from01: (x - 0.5) * 2 first, training.py:669-670) -> [N].
        keep=True: returns (out, state) where `state` owns the activations `backward_nhwc` needs (a workspace of its own,
        so that several calls can be pending, as the lip and the face term of one step are); otherwise the module's
        scratch workspace is reused."""
        lib = _abi.load()
        self._require_weights()
        packed = self.packed_weights()
        dev = packed.device
        if in0.device != dev or in1.device != dev:
            raise _abi.S2LError("LPIPS inputs must be on the GPU that holds its weights (no CPU fallback)")
        in0 = in0.detach().to(torch.float32).contiguous()
        in1 = in1.detach().to(torch.float32).contiguous()
        if in0.dim() != 4 or in0.shape[-1] != 3 or in0.shape != in1.shape:
            raise ValueError(f"LPIPS expects two [N,H,W,3] images; got {tuple(in0.shape)}, {tuple(in1.shape)}")
        N, H, W = in0.shape[:3]
        n = int(lib.s2l_lpips_work_floats(H, W, N))
        if N and n == 0:
            raise ValueError(f"LPIPS(alex) needs images of at least {MIN_SIZE}x{MIN_SIZE}; got {H}x{W}")
        if keep:
            work = torch.empty(n, dtype=torch.float32, device=dev)
        else:
            if self._work is None or self._work.numel() < n or self._work.device != dev:
                self._work = torch.empty(n, dtype=torch.float32, device=dev)
            work = self._work
        out = torch.empty(N, dtype=torch.float32, device=dev)
        precision = precision or self.conv_precision
        if precision not in ("fp32", "split"):
            raise ValueError(f"LPIPS precision must be 'fp32' or 'split', got {precision!r}")
        split = precision == "split"
        fn = "s2l_lpips_forward_split" if split else "s2l_lpips_forward"
        with torch.cuda.device(dev):
            _abi.check(getattr(lib, fn)(_p(packed), _p(in0), _p(in1), int(bool(from01)), _p(work), _p(out), H, W, N, _st()), fn)
        return (out, (work, N, H, W, bool(from01), split)) if keep else out

    def backward_nhwc(self, state, d_out: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """d loss / d in0 [N,H,W,3] for the `state` of a distance_nhwc(keep=True) call; added to `out` when given."""
        lib = _abi.load()
        work, N, H, W, from01, split = state
        packed = self.packed_weights()
        dev = packed.device
        d = d_out.detach().to(device=dev, dtype=torch.float32).reshape(N).contiguous()
        acc = out is not None
        if out is None:
            out = torch.empty(N, H, W, 3, dtype=torch.float32, device=dev)
        elif tuple(out.shape) != (N, H, W, 3) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != dev:
            raise ValueError("`out` must be a contiguous float32 [N,H,W,3] tensor on the module's GPU")
        fn = "s2l_lpips_backward_split" if split else "s2l_lpips_backward"
        with torch.cuda.device(dev):
            _abi.check(getattr(lib, fn)(_p(packed), _p(work), _p(d), int(from01), int(acc), _p(out), H, W, N, _st()), fn)
        return out

    # -- the package's signature ------------------------------------------------------------------------------------------
    def forward(self, in0, in1, retPerLayer=False, normalize=False):
        """in0, in1 [N,3,H,W] in [-1,1] ([0,1] with normalize=True: 2x - 1) -> [N,1,1,1]; differentiable in in0."""
        if retPerLayer:
            raise NotImplementedError("retPerLayer is not used by the reference (training.py:672)")
        from . import autograd as ag
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        d = ag.lpips_distance(self, in0.permute(0, 2, 3, 1), in1.permute(0, 2, 3, 1))
        return d.reshape(-1, 1, 1, 1)
