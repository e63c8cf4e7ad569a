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
        # T3 (training.py:83-91): the frozen lip-sync expert, only when the config asks for the sync loss
        self.use_syncloss = bool(kwargs.get("use_syncloss", self.cfg["training"].get("use_syncloss", False)))
        self.w_syncloss = float(kwargs.get("w_syncloss", self.cfg["training"].get("w_syncloss", 0.01)))
        self.syncnet = kwargs.get("syncnet")
        if self.use_syncloss and self.syncnet is None:
            from .syncnet import SyncNet_color
            self.syncnet = SyncNet_color().to(self.device)

    def load_checkpoint_syncnet(self, path, model=None):
        """training.py:130-138: `lipsync_expert.pth` holds {'state_dict': ...}, keys possibly prefixed by 'module.'."""
        ckpt = torch.load(path, map_location="cpu")
        sd = {k.replace("module.", ""): v for k, v in ckpt["state_dict"].items()}
        (model or self.syncnet).load_state_dict(sd)
        return model or self.syncnet

    def cosine_loss(self, a, v, y):
        """training.py:576-579."""
        from .syncnet import SyncLoss
        return SyncLoss(self.syncnet).cosine_loss(a, v, y)

    def get_sync_contrastive_loss(self, mel, g_rgb_pos, g_rgb_neg, syncnet_T=5, want_grad=False):
        """training.py:581-603.  want_grad=True also returns d loss / d g_rgb_pos (what autograd hands to the renderer)."""
        from .syncnet import SyncLoss
        return SyncLoss(self.syncnet, syncnet_T).get_sync_contrastive_loss(mel, g_rgb_pos, g_rgb_neg, want_grad=want_grad)

    def predict_lip_image(self, i, coords, audio, pose, data, rgb_zero, lms, seed):
        """Same arguments as the reference method; `pose`, `rgb_zero`, `lms` are unused under the
        May flags exactly as there.  One chunk = the whole lip image (batch_rays = H*W)."""
        chunk = coords[i:i + self.batch_rays, :]
        time_pts = data["index"] if seed is None else data["index"] + seed
        u01 = float(torch.rand(1, device=self.device))          # eps_shift draw (training.py:200)
        return predict_lip_image(self.model, chunk, audio, time_pts, self.height, self.width, u01)[:, :3]


class LipTrainStep:
    """Forward + backward of the lip-MLP training objective (BASELINE config 5, fp32 parity mode):

        loss = weight * mean_{frames, pixels, rgb} (predict_lip_image(frame) - target)^2

    i.e. `Trainer.predict_lip_image` (training.py:158-251) + `add_photometric_loss` (:605-619) and
    their autograd, as hand-written HIP kernels: rows -> forward with saved activations -> ensemble
    reduce -> MSE -> ensemble backward -> dz chain -> weight-gradient GEMMs.  Returns gradients
    keyed by the reference's state-dict names.  The tiny un-folding of the pack-time folds
    (G0 = W0 [Wuv|Wa|Wt] etc., four 256x256x126 products) uses torch.matmul on the device.
    """

    def __init__(self, model: TalkingFace, height: int, width: int, precision: str = "fp32"):
        """precision: 'fp32' (parity mode: exact-fp32 MFMA, saved state in fp32) or 'bf16' (BASELINE config 5: bf16 MFMA
        operands and saved state, fp32 accumulation / master weights / gradients; csrc/train_bf16.hip)."""
        from .rendering import get_coords
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.precision = precision
        self.model, self.h, self.w = model, int(height), int(width)
        self.lib = _abi.load()
        self.coords = get_coords(width, height, model.packed_weights().device)

    def _f(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.coords.device)

    def loss_and_grads(self, audio, frame_idx, targets, u01, weight: float = 1.0):
        lib, m = self.lib, self.model
        packed = m.packed_weights()
        dev = packed.device
        B, P = audio.shape[0], self.h * self.w
        N = 4 * P * B
        tgt = _dev_f32(targets, dev, "targets").reshape(B * P, 3)
        idx = [int(i) for i in (frame_idx.tolist() if isinstance(frame_idx, torch.Tensor) else frame_idx)]
        u = [float(v) for v in (u01.tolist() if isinstance(u01, torch.Tensor) else u01)]
        st = _stream()
        ck = _abi.check
        feat = m.audio_merge_forward(audio)                                  # [B,64]
        areas = self._f(N)
        bf16 = self.precision == "bf16"
        if bf16:
            Np = int(lib.s2l_bf16_rows_padded(N))
            pb = m.packed_weights_bf16()
            i16 = lambda n: torch.empty(n, dtype=torch.int16, device=dev)
            hT, dzT, xT = i16(8 * Np * 256), i16(8 * Np * 256), i16(Np * 128)
            masks = torch.empty(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
            x = None
        else:
            x = self._f(N, 128)
            hsave, dzsave = self._f(8, N, 256), self._f(8, N, 256)
        rgb, drgb, dxa = self._f(N, 3), self._f(N, 3), self._f(N, 64)
        pred, dpred = self._f(B * P, 3), self._f(B * P, 3)
        loss, mwork = self._f(1), self._f(1024)
        with torch.cuda.device(dev):
            if bf16:     # the embedded rows of the whole batch in one launch, straight to the bf16 operand image
                t_idx = torch.tensor(idx, dtype=torch.int64).to(dev, non_blocking=True)
                t_u = torch.tensor(u, dtype=torch.float32).to(dev, non_blocking=True)
                ck(lib.s2l_ensemble_rows_bf16(_ptr(packed), _ptr(self.coords), _ptr(feat), _ptr(t_idx), _ptr(t_u), self.w, self.h,
                                              _ptr(xT), _ptr(areas), P, B, st), "s2l_ensemble_rows_bf16")
            else:
                for b in range(B):   # rows of frame b: [b*4P, (b+1)*4P), tap-major inside
                    ck(lib.s2l_ensemble_rows(_ptr(packed), _ptr(self.coords), _ptr(feat[b]), idx[b], self.w, self.h,
                                             ctypes.c_float(u[b]), _ptr(x[b * 4 * P:]), _ptr(areas[b * 4 * P:]), P, st),
                       "s2l_ensemble_rows")
            if bf16:
                ck(lib.s2l_train_forward_bf16(_ptr(pb), _ptr(packed), _ptr(xT), _ptr(hT), _ptr(masks), _ptr(rgb), N, st),
                   "s2l_train_forward_bf16")
            else:
                ck(lib.s2l_train_forward(_ptr(packed), _ptr(x), _ptr(hsave), _ptr(rgb), N, st), "s2l_train_forward")
            ck(lib.s2l_ensemble_reduce_batch(_ptr(rgb), _ptr(areas), _ptr(pred), P, B, st), "s2l_ensemble_reduce_batch")
            ck(lib.s2l_mse(_ptr(pred), _ptr(tgt), ctypes.c_float(weight), _ptr(dpred), _ptr(mwork), _ptr(loss),
                           B * P * 3, st), "s2l_mse")
            ck(lib.s2l_ensemble_backward_batch(_ptr(dpred), _ptr(areas), _ptr(drgb), P, B, st), "s2l_ensemble_backward_batch")
            if bf16:
                ck(lib.s2l_train_backward_bf16(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dzT), _ptr(dxa), N, st),
                   "s2l_train_backward_bf16")
                work = self._f(int(lib.s2l_wgrad_bf16_work_floats()))
                lay = Np * 256

                def wgrad(k, inp, ldin, k_in, want_bias=True):          # dz of layer k against hT[inp] or the x tiles
                    out, db = self._f(256, k_in), (self._f(256) if want_bias else None)
                    src = xT if inp is None else hT[inp * lay:]
                    ck(lib.s2l_wgrad_bf16(_ptr(dzT[k * lay:]), _ptr(src), k_in, _ptr(work), _ptr(out), _ptr(db), N, st),
                       "s2l_wgrad_bf16")
                    return out, db
            else:
                ck(lib.s2l_train_backward(_ptr(packed), _ptr(drgb), _ptr(hsave), _ptr(dzsave), _ptr(dxa), N, st),
                   "s2l_train_backward")
                work = self._f(int(lib.s2l_split_work_floats(256 * 256)))

                def wgrad(k, inp, ldin, k_in, want_bias=True):
                    out, db = self._f(256, k_in), (self._f(256) if want_bias else None)
                    src = x if inp is None else hsave[inp]
                    ck(lib.s2l_wgrad(_ptr(dzsave[k]), 256, _ptr(src), ldin, k_in, _ptr(work), _ptr(out), _ptr(db), N, st),
                       "s2l_wgrad")
                    return out, db

            def colsum(src, c):
                out = self._f(c)
                ck(lib.s2l_small_outer(None, 0, 1, _ptr(src), c, c, _ptr(work), _ptr(out), N, st), "s2l_small_outer")
                return out

            g = {}
            for k in range(1, 8):                          # pts_linears[k]: h_{k-1} -> h_k
                dw, db = wgrad(k, k - 1, 256, 256)
                if k == 5:
                    dw5b, dc5 = dw, db
                else:
                    g[f"pts_linears.{k}.weight"], g[f"pts_linears.{k}.bias"] = dw, db
            dG0, dc0 = wgrad(0, None, 128, 128)
            dG5, _ = wgrad(5, None, 128, 128, want_bias=False)
            dwout = self._f(3, 256)
            if bf16:
                dbout = self._f(3)
                ck(lib.s2l_out_grad_bf16(_ptr(drgb), _ptr(hT[7 * lay:]), _ptr(work), _ptr(dwout), _ptr(dbout), N, st),
                   "s2l_out_grad_bf16")
                g["output_linear.weight"], g["output_linear.bias"] = dwout, dbout
            else:
                ck(lib.s2l_small_outer(_ptr(drgb), 3, 3, _ptr(hsave[7]), 256, 256, _ptr(work), _ptr(dwout), N, st),
                   "s2l_small_outer")
                g["output_linear.weight"], g["output_linear.bias"] = dwout, colsum(drgb, 3)
            # per-frame gradient of the audio feature (rows of frame b are contiguous), then the encoder backward
            da = dxa.reshape(B, 4 * P, 64).sum(dim=1).contiguous()
            na = int(lib.s2l_audio_grad_floats())
            awork, agrads = self._f(((B + 3) // 4) * na), self._f(na)
            a32 = _dev_f32(audio, dev, "audio")
            ck(lib.s2l_audio_backward(_ptr(packed), _ptr(a32), _ptr(da), _ptr(awork), _ptr(agrads), B, st),
               "s2l_audio_backward")
            off = 0
            for name in _abi.TENSOR_ORDER[:12]:
                p_ = dict(m.named_parameters())[name]
                g[name] = agrads[off:off + p_.numel()].reshape(p_.shape)
                off += p_.numel()

        # un-fold G0 = W0 [Wuv|Wa|Wt], c0 = W0 (buv+ba+bt) + b0 (and the skip twins) -- tiny device GEMMs
        sd = dict(m.named_parameters())

        def unfold(w_first, names, dG, dc):
            C = torch.cat([sd[f"{n}.weight"].detach() for n in names], dim=1)             # [256,126]
            bsum = sum(sd[f"{n}.bias"].detach() for n in names)
            dW = dG[:, :126] @ C.t() + torch.outer(dc, bsum)
            dC = w_first.t() @ dG[:, :126]
            dbs = w_first.t() @ dc
            out = {}
            for n, (lo, hi) in zip(names, ((0, 42), (42, 106), (106, 126))):
                out[f"{n}.weight"], out[f"{n}.bias"] = dC[:, lo:hi].contiguous(), dbs.clone()
            return dW, out

        W0 = sd["pts_linears.0.weight"].detach()
        W5a = sd["pts_linears.5.weight"].detach()[:, :256]
        dW0, part = unfold(W0, ("fc_uv", "fc_audio", "fc_time"), dG0, dc0)
        g.update(part)
        g["pts_linears.0.weight"], g["pts_linears.0.bias"] = dW0, dc0
        dW5a, part = unfold(W5a, ("fc_uv_skip", "fc_audio_skip", "fc_time_skip"), dG5, dc5)
        g.update(part)
        g["pts_linears.5.weight"], g["pts_linears.5.bias"] = torch.cat([dW5a, dw5b], dim=1), dc5
        return loss, g, {"pred": pred.reshape(B, P, 3), "d_audio_feat": da}


def apply_grads(model: TalkingFace, grads) -> None:
    """Install the gradients returned by `LipTrainStep.loss_and_grads` as `.grad` of the matching
    parameters, so a stock optimizer (the reference uses Adam(lr=1e-4), train.py:128) can step."""
    params = dict(model.named_parameters())
    for name, g in grads.items():
        p = params[name]
        p.grad = g.reshape(p.shape).to(p.dtype).contiguous()
