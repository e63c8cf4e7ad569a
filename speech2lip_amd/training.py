"""Training-time side of the lip path (BASELINE config 5, §8f-4 light half).

  * `Trainer` keeps the reference's method signatures (`src/face_simple/training.py:158, 576-603`) for the May flag set, so a
    training loop that calls them needs no other change; with autograd recording they return tensors that
    `loss.backward()` (training.py:559) can differentiate (speech2lip_amd.autograd).
  * `LipTrainStep` is the batched engine underneath: `forward` renders the 4-tap local ensemble of a batch of frames with
    saved activations, `backward` turns d loss / d pred into the gradients of all 42 hot-path tensors -- fp32 exact-parity
    mode or the bf16 mode BASELINE config 5 names.
  * `SyncChain` carries the lip-sync expert's loss back to the rendered lips through the crop/resize, the frozen
    post-fusion U-Net -- in whichever BatchNorm mode the sub-module is in: eval (running statistics; crop-window and bf16
    operand forms) or train (batch statistics per one-frame call: what the reference's loop actually runs, because
    Trainer.train_step calls self.model.train() on every step, training.py:150; golden G16) -- and the paste + head-pose-warp
    composite (training.py:491-557), and `StageOneStep` adds it (and the optional face photometric term, :458-459) to the MSE step.
Every arithmetic step is a hand-written HIP kernel behind the C-ABI; torch is device memory, streams and views.
"""
from __future__ import annotations

import ctypes

import torch

from . import _abi
from ._modcache import param_map, set_training, state_tensors
from .talking_face import TalkingFace, _dev_f32, _ptr, _stream

MLP_TENSORS = _abi.TENSOR_ORDER[12:]      # fc_* and pts_linears.* and output_linear.* (30 tensors)
AUDIO_TENSORS = _abi.TENSOR_ORDER[:12]    # encoder_conv.* and encoder_fc1.*


def _f(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


def predict_lip_image(model: TalkingFace, coords, audio, index, height: int, width: int, u01: float):
    """coords [HW,2], audio [1,16,29], frame index, the U(0,1) draw of training.py:200 -> [HW,3] (no graph)."""
    lib = _abi.load()
    packed = model.packed_weights()
    dev = packed.device
    c = _dev_f32(coords, dev, "coords")
    if c.dim() != 2 or c.shape[1] != 2:
        raise ValueError(f"coords must be [N,2], got {tuple(c.shape)}")
    feat = model._audio_encode(audio)               # encoder once, then shared by all pixels (:165/:171)
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

    _loss_conv_precision = None      # form of the frozen loss nets' convolutions: None = the nets' own default (exact fp32); "split" under precision="bf16"

    # keyword names of the reference constructor (training.py:21-37) whose May value is the only one the hot path implements:
    # passing anything else is an error, not a silently different model
    _MAY_ONLY = {"use_audio_net": True, "use_head_pose_net": False, "use_coords2audio": False, "use_delta_uv": False,
                 "use_canonical_loss": False, "use_temp_consist": False, "use_head_pose": False, "use_audio": True,
                 "use_loss_bg": False, "use_loss_face": False, "use_loss_facewoaudio": False, "use_loss_lip": False,
                 "use_coords_mapping": False, "add_noise_uv": False, "add_noise_audio": False, "use_time": True,
                 "use_merge_loss": False, "update_pose": False, "use_c_lip": False, "use_fusion_face": True,
                 "fusion_lip_only": True}
    # accepted and stored, no effect on this path (NeRF-era sampling knobs the face_simple trainer never reads back)
    _STORED = ("threshold", "n_sample_points", "n_sample_points_fine", "lindisp", "raw_noise_std", "perturb", "local_rank")

    def __init__(self, model, optimizer=None, device=None, out_dir=None, cfg=None, batch_rays=None, **kwargs):
        """Positional order and keyword names of the reference constructor (training.py:21-37):
        `Trainer(model, optimizer, device, out_dir, cfg=cfg, batch_rays=..., lambda_rgb=..., use_syncloss=..., ...)`, so
        `get_trainer` (src/face_simple/config.py:25-94) and positional callers bind as they do there.  Flags of the May set
        are accepted with their May value and refused with any other; `multi_gpu=True` does not wrap the model in DDP (the
        hand-written backward does not run through DDP's hooks): with an initialised process group `train_stage1` averages the
        gradients over the ranks in one bucket (`sharded.allreduce_grads`) before the optimizer step -- the same result.
        Extras of this build: `syncnet=`, `perceptual_loss_fn=` (pre-built frozen nets), `w_photometric_loss=` (alias of
        the reference's `lambda_rgb`); `precision="fp32"` (default: the exact kernels, what goldens G11 / G14 / G16 pin) or
        `"bf16"` (BASELINE config 5's arithmetic: bf16 MFMA operands in the MLP and the post-fusion U-Net, fp32 accumulation
        and master weights); `hole_noise="host"` (default: the black-hole fields come from the reference's CPU generator
        stream, tf_nerf.py:306-318) or `"device"` (drawn on the GPU: same distribution, another stream, no 7 ms of host
        `randn` per frame); the method `train_steps` (K frames per optimisation step); and `fused_step=True` (with
        `precision="bf16"`): `train_step` itself runs its one frame through that fused engine (`train_steps` with K = 1: the same
        draws in the same order, the same losses and gradients within the bf16 tolerances of G11 / G14, a third less wall time
        per iteration than the autograd route); flag sets the fused engine does not implement fall back to the autograd route."""
        if isinstance(device, dict) or isinstance(out_dir, dict):
            raise TypeError("Trainer(model, optimizer, device, out_dir, cfg=...): cfg is the FIFTH argument, as in the reference")
        self.model = model
        self.optimizer = optimizer
        self.cfg = cfg if cfg is not None else model.cfg
        self.device = torch.device(device) if device is not None else model.device
        self.out_dir = out_dir if out_dir is not None else self.cfg.get("training", {}).get("out_dir")
        for name, may in self._MAY_ONLY.items():
            if name in kwargs and bool(kwargs.pop(name)) != may:
                raise NotImplementedError(f"Trainer({name}={not may}) is outside the May flag set this path implements (SURVEY.md §8a)")
        for name in self._STORED:
            if name in kwargs:
                setattr(self, name, kwargs.pop(name))
        tc = self.cfg["training"]
        self.height = int(self.cfg["data"]["height"])
        self.width = int(self.cfg["data"]["width"])
        self.batch_rays = int(batch_rays if batch_rays is not None else tc.get("batch_rays", self.height * self.width))
        self.multi_gpu = bool(kwargs.pop("multi_gpu", False))
        if self.multi_gpu:      # what wrapping the model in DistributedDataParallel does at construction (training.py:41)
            from .sharded import broadcast_module_state
            broadcast_module_state(model, src=0)
        self.precision = kwargs.pop("precision", "fp32")
        if self.precision not in ("fp32", "bf16"):
            raise ValueError(f"Trainer(precision=...) must be 'fp32' or 'bf16', got {self.precision!r}")
        model.train_precision = self.precision
        self._loss_conv_precision = "split" if self.precision == "bf16" else "fp32"
        hn = kwargs.pop("hole_noise", None)
        if hn is not None:
            if hn not in ("host", "device"):
                raise ValueError(f"Trainer(hole_noise=...) must be 'host' or 'device', got {hn!r}")
            model.hole_noise = hn
        self.fused_step = bool(kwargs.pop("fused_step", False))
        self._stage_step = None
        self.use_audio = self.use_audio_net = self.use_time = True
        self.use_delta_uv = self.add_noise_audio = self.add_noise_uv = False
        self.use_head_pose = self.use_head_pose_net = self.use_coords2audio = self.use_coords_mapping = False
        self.audio_dims = model.audio_dims
        # T3 (training.py:83-91): the frozen lip-sync expert, only when the config asks for the sync loss
        self.use_syncloss = bool(kwargs.pop("use_syncloss", tc.get("use_syncloss", False)))
        self.w_syncloss = float(kwargs.pop("w_syncloss", tc.get("w_syncloss", 0.01)))
        self.syncnet = kwargs.pop("syncnet", None)
        ckpt = kwargs.pop("syncnet_checkpoint_path", "models/lipsync_expert.pth")      # training.py:88
        if self.use_syncloss and self.syncnet is None:
            import os
            from .syncnet import SyncNet_color
            self.syncnet = SyncNet_color().to(self.device)
            for p_ in self.syncnet.parameters():
                p_.requires_grad = False
            if os.path.exists(ckpt):
                self.load_checkpoint_syncnet(ckpt, self.syncnet)
            else:      # the expert's weights are not part of the reference repository either (README: a separate download)
                import logging
                logging.getLogger(__name__).warning("%s not found: the sync loss runs on a randomly initialised SyncNet", ckpt)
        if self.use_syncloss:
            self.use_low_resolution = bool(tc.get("use_low_resolution", False))
            if self.use_low_resolution:
                raise NotImplementedError("training.use_low_resolution is outside the May flag set")
        # the other loss switches of training.py:21-100 (May values unless overridden)
        self.use_post_fusion = bool(kwargs.pop("use_post_fusion", self.cfg["model"].get("use_post_fusion", True)))
        self.use_post_fusion_wface = bool(self.cfg["model"].get("use_post_fusion_wface", False))
        self.fusion_lip_only = self.use_fusion_face = True
        # lambda_rgb lives under cfg['model'] in the reference (src/face_simple/config.py:41, may.yaml:11); cfg['training'] is a fallback
        lam = kwargs.pop("lambda_rgb", self.cfg["model"].get("lambda_rgb", tc.get("lambda_rgb", 1.0)))
        self.w_photometric_loss = float(kwargs.pop("w_photometric_loss", lam))
        self.w_post_fusion = float(kwargs.pop("w_post_fusion", tc.get("w_post_fusion", 1.0)))
        self.use_perceptual_loss = bool(kwargs.pop("use_perceptual_loss", tc.get("use_perceptual_loss", False)))
        self.w_perceptual_loss = float(kwargs.pop("w_perceptual_loss", tc.get("w_perceptual_loss", 0.01)))
        self.perceptual_loss_fn = kwargs.pop("perceptual_loss_fn", None)
        if self.use_perceptual_loss and self.perceptual_loss_fn is None:      # training.py:75-76
            from .lpips import LPIPS
            self.perceptual_loss_fn = LPIPS(net="alex", version="0.1", model_path="models/lpips_weights_v0.1/alex.pth").to(self.device)
        if tc.get("fix_post_net", False) is True and getattr(model, "post_fusion_unet", None) is not None:      # training.py:121-129
            for p_ in model.post_fusion_unet.parameters():
                p_.requires_grad = False
            model.post_fusion_unet.eval()
        if kwargs:
            raise TypeError(f"Trainer got unexpected keyword arguments {sorted(kwargs)} (reference: training.py:21-37)")

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
        """training.py:581-603.  With autograd recording and a generated window that requires grad the returned loss is
        differentiable (its backward is s2l_syncnet_face_backward); want_grad=True instead returns (loss, d loss / d g_rgb_pos)."""
        from .syncnet import SyncLoss
        if not want_grad and torch.is_grad_enabled() and isinstance(g_rgb_pos, torch.Tensor) and g_rgb_pos.requires_grad:
            from .autograd import sync_contrastive_loss
            return sync_contrastive_loss(SyncLoss(self.syncnet, syncnet_T, self._loss_conv_precision), mel, g_rgb_pos, g_rgb_neg)
        return SyncLoss(self.syncnet, syncnet_T, self._loss_conv_precision).get_sync_contrastive_loss(mel, g_rgb_pos, g_rgb_neg, want_grad=want_grad)

    def prepare_coords(self, coord, b):
        """training.py:253-261 (use_coords_mapping off): the regular pixel grid, tiled b times.  The tensor is built once per
        (width, height, b) and registered as a known regular grid, so the fused renderer need not compare it with
        `get_coords` element by element on every call (callers must not write into it -- the May flags never do)."""
        from .rendering import get_coords
        from . import autograd
        key = (int(self.width), int(self.height), int(b))
        cache = self.__dict__.setdefault("_coords_cache", {})
        if key not in cache:
            c = get_coords(key[0], key[1], self.device).tile(b, 1)
            cache[key] = c
            if b == 1:
                autograd.register_regular_grid(c, key[1], key[0])
        return cache[key]

    def predict_lip_image(self, i, coords, audio, pose, data, rgb_zero, lms, seed):
        """Same arguments as the reference method; `pose`, `rgb_zero`, `lms` are unused under the May flags exactly as
        there.  One chunk = the whole lip image (batch_rays = H*W).  Differentiable when autograd is recording."""
        chunk = coords[i:i + self.batch_rays, :]
        time_pts = data["index"] if seed is None else data["index"] + seed
        u01 = torch.rand(1, device=self.device)                 # eps_shift draw (training.py:200); it never leaves the device
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.model._hot_tensors()):
            from .autograd import predict_lip_image as predict_with_graph
            return predict_with_graph(self.model, chunk, audio, time_pts, self.height, self.width, u01, self.precision)[:, :3]
        return predict_lip_image(self.model, chunk, audio, time_pts, self.height, self.width, float(u01))[:, :3]

    def compute_rel_pose(self, canonical_euler, canonical_trans, euler, trans, img_batch_size=1, device=None):
        """training.py:263-268."""
        from . import geometry
        return geometry.compute_rel_pose(canonical_euler, canonical_trans, euler, trans)

    def compute_rel_pose_inverse(self, canonical_euler, canonical_trans, euler, trans, img_batch_size=1, device=None):
        """training.py:270-275."""
        from . import geometry
        return geometry.compute_rel_pose_inverse(canonical_euler, canonical_trans, euler, trans)

    def inverse_warping(self, tgt_depth, rel_pose, src_img):
        """training.py:296-314 -> (predict_img NCHW, cam_points_z [F,1,H,W]) (forward only; the loss below is the
        differentiable form of the pair)."""
        from . import geometry
        return geometry.inverse_warping(self.cfg, tgt_depth, rel_pose, src_img, return_z=True)

    _defer_host = False      # inside train_stage1: the loss terms stay on the device until the step has been launched (no mid-step sync)

    def _term(self, t):
        """A loss term for the bookkeeping entries of the loss dict: the reference's `.detach().cpu()` (training.py:617, 633, 673) --
        a device synchronisation each -- or, while `train_stage1` is composing a step, the detached device tensor; `_loss_to_host`
        moves the entries to the host once the whole step (backward, optimizer) is in the stream."""
        return t.detach() if self._defer_host else t.detach().cpu()

    @staticmethod
    def _loss_to_host(loss):
        for k, v in loss.items():
            if k != "loss" and isinstance(v, torch.Tensor) and v.is_cuda:
                loss[k] = v.cpu()
        return loss

    def canonical_depth_photo_loss(self, tgt_depth, rel_pose, src_img, target, loss, mask=None, weights=1.0):
        """training.py:470-477 in one call: inverse_warping + add_loss_canonical_depth_photo (:621-634), fused on the device
        and differentiable w.r.t. `tgt_depth` (model.canonical_depth_head) when autograd is recording."""
        from . import geometry
        l = geometry.depth_photo_loss(self.cfg, tgt_depth, rel_pose, src_img, target, mask, weights)
        loss["loss"] = loss["loss"] + l
        loss["loss_canonical_depth_photo"] = loss.get("loss_canonical_depth_photo", 0) + self._term(l)
        return l

    def add_photometric_loss(self, prediction, target, loss, coarse=False, mask=None, weights=1.0):
        """training.py:605-619 (mask=None branch): differentiable through autograd.mse when recording."""
        from .autograd import mse
        loss_rgb = mse(prediction, target, weights)
        loss["loss"] = loss["loss"] + loss_rgb
        loss["loss_rgb"] = loss["loss_rgb"] + self._term(loss_rgb)


    def add_perceptual_loss(self, prediction, target, loss, mask=None, weights=1.0):
        """training.py:655-674: prediction, target [B,H,W,3] in [0,1] -> (x - 0.5) * 2 -> LPIPS(alex).mean() * weights.
        `mask` is the reference's [B,1|3,H,W] multiplier (it only ever passes ones, :455); `self.perceptual_loss_fn` is a
        speech2lip_amd.LPIPS (set it as the reference's constructor does, :76)."""
        from .autograd import lpips_distance
        if mask is not None:
            m = mask.permute(0, 2, 3, 1)
            prediction, target = m * prediction, m * target
        d = lpips_distance(self.perceptual_loss_fn, prediction, target, from01=True, precision=self._loss_conv_precision)
        loss_perceptual = d.mean() * weights
        loss["loss"] = loss["loss"] + loss_perceptual
        loss["loss_perceptual"] = loss.get("loss_perceptual", 0) + self._term(loss_perceptual)


    def train_step(self, data, data_zero=None, it=None, seed=None):
        """training.py:140-155, statement for statement: `self.model.train()`, then `train_stage1`; returns
        (loss_rgb.item(), loss dict).  NB `self.model.train()` puts EVERY sub-module into train mode, including a post-fusion
        U-Net that train.py:188-197 froze and switched to eval() when `it` passed 100000: from then on the reference's loop runs
        that U-Net with frozen parameters but BatchNorm batch statistics (and moving running statistics).  The drop-in does the
        same, because `post_fusion2_onlylip` follows the sub-module's own mode (golden G16)."""
        set_training(self.model, True)      # = self.model.train() (training.py:150) without the generator walk
        self._broadcast_buffers()
        if self.cfg["training"].get("stage", "stage1") == "stage1":
            if self.fused_step and self.precision == "bf16" and self._fused_step_covers():
                loss, loss_all = self.train_stage1_frames(data, it=it, seed=seed)
            else:
                loss, loss_all = self.train_stage1(data, it=it, seed=seed)
        else:
            raise NotImplementedError("only training.stage == 'stage1' exists in the reference (training.py:152)")
        return float(loss), loss_all

    def _broadcast_buffers(self):
        """DistributedDataParallel(broadcast_buffers=True) hands rank 0's buffers -- the BatchNorm running statistics -- to every rank
        before each forward: with per-rank frames they would otherwise drift apart rank by rank.  No-op without `multi_gpu`."""
        if self.multi_gpu and getattr(self.model, "post_fusion_unet", None) is not None:
            from .sharded import broadcast_module_state
            broadcast_module_state(self.model.post_fusion_unet, src=0)

    def _fused_step_covers(self) -> bool:
        """Whether `train_stage1_frames` implements the configured loss set (what it refuses with NotImplementedError)."""
        tc = self.cfg["training"]
        face_on = bool(self.use_post_fusion)
        lip_perc = self.use_perceptual_loss and tc.get("use_lip_perc_loss", "v1") == "v1"
        face_perc = self.use_perceptual_loss and tc.get("use_face_perc_loss", True) is True
        return not (bool(tc.get("use_canonical_depth_loss_photo_v2", False)) or self.batch_rays != self.height * self.width
                    or tc.get("use_lip_photo_loss", "v1") != "v1" or (face_on and tc.get("use_face_photo_loss", True) is not True)
                    or (self.use_perceptual_loss and face_on and lip_perc != face_perc))

    def visualize(self, visualize, logger, it):
        """training.py:676-740 under the May flags: eval mode, the 4-tap ensemble render of the validation frame (`seed=0`, one
        chunk = the whole lip image), PSNR = -10 log10(mean((pred - rgb)^2)) against `visualize['rgb']`; without a logger the PSNR is
        returned (what `evaluate` uses), with one the reference's four records are written; then train mode again."""
        import numpy as np
        from .rendering import get_coords
        self.model.eval()
        try:
            with torch.no_grad():
                image = visualize["rgb"].squeeze()
                rgb_zero = visualize["rgb_zero"].reshape(-1, 3) if "rgb_zero" in visualize else None
                H, W = int(visualize["height"]), int(visualize["width"])
                if (H, W) != (self.height, self.width) or self.batch_rays != H * W:
                    raise NotImplementedError("visualize renders the configured lip crop in one chunk (batch_rays = H*W)")
                audio = visualize["audio"].to(self.device)
                coords = get_coords(W, H, self.device)
                on = {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in visualize.items()}
                rgb_map = self.predict_lip_image(0, coords, audio, None, on, rgb_zero, None, seed=0).reshape(H, W, 3)
                mse = torch.mean((rgb_map.cpu() - image.cpu()) ** 2)
                psnr = -10.0 * torch.log(mse) / torch.log(torch.Tensor([10.0]))
                if logger is None:
                    return psnr
                to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)      # utils.py:6
                logger.add_image("rgb_prediction", to8b(rgb_map.cpu().numpy()).transpose([2, 0, 1]), it)
                logger.add_image("rgb_gt", image.cpu().numpy().transpose([2, 0, 1]), it)
                logger.add_scalar("val_mini/loss", mse, it)
                logger.add_scalar("val_mini/psnr", psnr, it)
        finally:
            self.model.train()

    def evaluate(self, val_loader, focal_length=None, batch_size=None, it=None):
        """training.py:742-751: mean PSNR of `visualize` over the validation loader."""
        psnr = torch.stack([self.visualize(inputs, None, it=it) for inputs in val_loader], 0).mean()
        return {"psnr": psnr}

    def _average_gradients_over_ranks(self):
        """What DistributedDataParallel (training.py:41) does for the reference: every rank ends the backward with the MEAN of the
        ranks' gradients.  One flattened bucket, one all-reduce (sharded.allreduce_grads); no-op without `multi_gpu`, without a
        process group, or with one rank."""
        import torch.distributed as dist
        if not (self.multi_gpu and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        from .sharded import allreduce_grads
        named = {n: p for n, p in param_map(self.model).items() if p.grad is not None}
        avg = allreduce_grads({n: p.grad for n, p in named.items()})
        for n, p in named.items():
            p.grad.copy_(avg[n])

    def train_stage1(self, data, eval_model=False, it=None, seed=None):
        """One optimisation step as the reference performs it (training.py:347-574) under the May flags, through the drop-in's
        methods and their hand-written backward kernels:  zero_grad -> lip render -> MSE [+ LPIPS] on the lip -> composite with
        black holes + post-fusion U-Net -> [LPIPS +] MSE on the face -> canonical-depth photo loss (v2) -> after it > 100000 the
        sync loss over the 5-frame window -> backward -> NaN check -> optimizer.step().  Returns (loss_rgb, loss dict) like the
        reference.  One frame per call (batch_size 1, batch_rays = H*W: the only case the reference's own crop code supports,
        :536-537)."""
        self._defer_host = True
        try:
            return self._train_stage1(data, eval_model, it, seed)
        finally:
            self._defer_host = False

    def _train_stage1(self, data, eval_model, it, seed):
        tc, m = self.cfg["training"], self.model
        if self.optimizer is None:
            raise ValueError("train_stage1 steps an optimizer: construct Trainer(model, optimizer=...)")
        b, self.height, self.width = int(data["rgb"].shape[0]), int(data["rgb"].shape[1]), int(data["rgb"].shape[2])
        if b != 1 or self.batch_rays != self.height * self.width:
            raise NotImplementedError("train_stage1: one frame per step with batch_rays = H*W (the May configuration)")
        depth_v2 = bool(tc.get("use_canonical_depth_loss_photo_v2", False))
        loss = {"loss_rgb": 0}
        if self.use_perceptual_loss:
            loss["loss_perceptual"] = 0
        if self.use_syncloss:
            loss["loss_sync"] = 0
        if depth_v2:
            loss["loss_canonical_depth_photo"] = 0
        dev = self.device
        # every tensor of the batch crosses to the device ONCE (the window loop below reads the canonical face, the mask and the
        # 10-MB pose-grid window five times each)
        data = {k: (v.to(dev, non_blocking=True) if isinstance(v, torch.Tensor) and v.is_floating_point() and v.numel() > 16 else v)
                for k, v in data.items()}
        on = lambda t: t.to(dev) if isinstance(t, torch.Tensor) else t
        H, W = self.height, self.width
        coords = self.prepare_coords(data.get("coord"), b)
        rgb = on(data["rgb"]).reshape(-1, 3)
        self.optimizer.zero_grad()
        loss["loss"] = 0
        first = int(data["index"].reshape(-1)[0]) if isinstance(data["index"], torch.Tensor) else int(data["index"])
        rgb_map = self.predict_lip_image(0, coords, on(data["audio"]), None, {"index": torch.tensor([first])}, None, None, seed=seed)
        if tc.get("use_lip_photo_loss", "v1") == "v1":
            self.add_photometric_loss(rgb_map, rgb, loss, weights=self.w_photometric_loss)
        if self.use_perceptual_loss and tc.get("use_lip_perc_loss", "v1") == "v1":
            self.add_perceptual_loss(rgb_map.reshape(1, H, W, 3), rgb.reshape(1, H, W, 3), loss, weights=self.w_perceptual_loss)
        x0, y0 = data["lip_lefttop_x"], data["lip_lefttop_y"]
        if self.use_post_fusion:
            face_gt, face_can = on(data["rgb_face_ori"]), on(data["rgb_face_zero"])
            recon, _, _ = m.post_fusion2_onlylip(rgb_map.reshape(1, H, W, 3), face_can, face_gt, on(data["mask_lip_canonical"]), x0, y0,
                                                 on(data["coord"]), mask_head_observed=None, use_post_fusion_blackaug=True)
            if self.use_perceptual_loss and tc.get("use_face_perc_loss", True) is True:
                self.add_perceptual_loss(recon, face_gt, loss, mask=torch.ones_like(recon).permute(0, 3, 1, 2),
                                         weights=self.w_perceptual_loss * self.w_post_fusion)
            if tc.get("use_face_photo_loss", True) is True:
                self.add_photometric_loss(recon, face_gt, loss, weights=self.w_photometric_loss * self.w_post_fusion)
            if depth_v2:      # :461-477: warp the observed frame into the canonical view with the learned depth
                rel_pose = self.compute_rel_pose_inverse(on(data["canonical_euler"]), on(data["canonical_trans"]), on(data["euler"]),
                                                         on(data["trans"]), device=dev)
                mask = on(data["mask_head_3DMM_canonical"]) * (1 - on(data["mask_face_3DMM_canonical"]))
                self.canonical_depth_photo_loss(m.canonical_depth_head, rel_pose, face_gt, face_can, loss, mask=mask)
        if self.use_syncloss and it is not None and it > 100000 and tc.get("stage", "stage1") == "stage1":
            from .autograd import crop_resize
            total = int(data["total_frame"].reshape(-1)[0]) if isinstance(data["total_frame"], torch.Tensor) else int(data["total_frame"])
            # the T window frames in ONE call each of the renderer, the composite + U-Net and the crop: every frame is what its own
            # call (training.py:504-548, one after the other) computes -- the draws are made in that loop's order, the train-mode
            # U-Net treats each frame as its own statistics group in frame order -- at a fifth of the launches
            from .autograd import predict_lip_images
            Tn = int(data["audio_window"].shape[1])
            u_win = torch.cat([torch.rand(1, device=dev) for _ in range(Tn)])            # eps_shift of each render (training.py:200)
            idxs = [min(first + t, total - 1) + (0 if seed is None else int(seed)) for t in range(Tn)]
            lips = predict_lip_images(m, self.prepare_coords(None, b), on(data["audio_window"])[0], idxs, H, W, u_win, self.precision)
            gt_f = on(data["rgb_face_ori"]).expand(Tn, -1, -1, -1)
            merged, _, _ = m.post_fusion2_onlylip(lips[:, :, :3].reshape(Tn, H, W, 3), on(data["rgb_face_zero"]), gt_f, on(data["mask_lip_canonical"]),
                                                  x0, y0, on(data["coord_window"])[0], use_canonical_space=False, _frames_as_calls=True)
            bbox = data["canonical_face_bbox"][0] if isinstance(data["canonical_face_bbox"], torch.Tensor) else data["canonical_face_bbox"]
            rgb_window = crop_resize(merged, [float(v) for v in bbox], (96, 96), window_t=Tn)      # [B,C,T,H,W] (:547-548)
            loss_sync = self.get_sync_contrastive_loss(on(data["mel"]), rgb_window, on(data["rgb_window_neg"])) * self.w_syncloss
            loss["loss_sync"] = loss["loss_sync"] + loss_sync
            loss["loss"] = loss["loss"] + loss_sync
        loss["loss"].backward()
        self._average_gradients_over_ranks()
        nan_check = self._step_optimizer()
        self._check_weights_report(nan_check)      # (the step's only wait besides the values it returns)
        self._loss_to_host(loss)
        return loss["loss_rgb"], loss

    def _check_weights(self):
        """check_weights (src/common.py:56-64): warn for every state-dict tensor that holds a NaN.  One multi-tensor norm and
        ONE device synchronisation for the whole state dict (a NaN anywhere makes that tensor's norm NaN) instead of one
        `isnan().any()` round trip per tensor (~100 with the U-Net: milliseconds of an 8-ms iteration)."""
        self._check_weights_report(self._check_weights_launch())

    def _check_weights_launch(self, fused_optimizer=None):
        """The device half of `_check_weights`, where the reference calls it (before optimizer.step(), training.py:572): the norms, the
        NaN flags and their copy to a pinned host buffer go into the stream; nothing waits.  With a `FusedAdam` the parameters it is about
        to step are left to ITS pass (the kernel flags a parameter that holds a NaN as it reads it, i.e. at this very point of the step):
        only the tensors it does not own -- buffers, frozen parameters -- take the norm scan."""
        skip = fused_optimizer.covers() if fused_optimizer is not None else ()
        every = [(k, v) for k, v in state_tensors(self.model) if v.dtype.is_floating_point and v.numel()]
        named = [(k, v) for k, v in every if id(v) not in skip]
        names = {id(v): k for k, v in every if id(v) in skip} if skip else {}
        host = done = None
        if named:
            bad = torch.isnan(torch.stack(torch._foreach_norm([v for _, v in named])))
            host = torch.empty(bad.numel(), dtype=torch.bool, pin_memory=bad.is_cuda)      # (per step: a pipelined caller reads it a step later)
            host.copy_(bad, non_blocking=True)
            if bad.is_cuda:
                done = torch.cuda.Event()
                done.record()
        return [named, host, done, names, fused_optimizer, None]

    @staticmethod
    def _check_weights_report(pending):
        """The host half: wait for the flags (by then the optimizer's kernels are queued behind them) and warn like the reference."""
        named, host, done, names, fused, jobs = pending
        bad_names = []
        if done is not None:
            done.synchronize()
        if host is not None and bool(host.any()):
            bad_names += [k for (k, _), b in zip(named, host.tolist()) if b]
        if fused is not None:
            bad_names += [names.get(id(p), "<parameter>") for p, b in fused.nan_report(jobs) if b]
        if bad_names:
            import logging
            for k in bad_names:
                logging.getLogger(__name__).warning("NaN Values detected in model weight %s." % k)

    def _step_optimizer(self):
        """check_weights + optimizer.step() as training.py:572-573 orders them; returns what `_check_weights_report` needs."""
        from .optim import FusedAdam
        fused = self.optimizer if isinstance(self.optimizer, FusedAdam) else None
        nan_check = self._check_weights_launch(fused)
        self.optimizer.step()
        if fused is not None:
            nan_check[5] = fused.take_nan_jobs()
        return nan_check

    def train_steps(self, batch, it=None, seed=None, wait: bool = True):
        """K frames of the reference's loop in ONE optimisation step: `batch` is a list of K `load_one_frame` dictionaries (or one
        collated dictionary whose tensors carry a leading batch axis K).  Every frame is evaluated exactly as `train_step`
        evaluates it alone -- its own 4-tap ensemble draw, its own black-hole coin and noise fields, its own 5-frame sync window
        after `it > 100000`, its own one-frame BatchNorm statistics group in the post-fusion U-Net (running statistics move
        frame by frame) -- but all K go through the fused engine (`StageOneStep`: one MLP forward / backward for the K main and
        5 K window frames, one composite / U-Net / SyncNet pass per group of frames) and the optimizer steps ONCE on the mean of
        the K per-frame losses (the gradient of K single calls averaged, as a batch_size-K DataLoader would give the reference
        if its crop code supported it, training.py:536-537).  The random draws are made frame by frame in `train_step`'s order,
        so K = 1 consumes the generators like one `train_step` call.  The canonical-depth photo loss is not part of this
        entry.  Returns (loss_rgb, loss dict) like `train_step`, the values being means over the K frames.
        `wait=False` returns a `PendingStep` instead: the step is queued and nothing has waited for the device; `.result()` gives the
        same pair (and issues the NaN warnings) -- a loop that calls it AFTER queueing the next step keeps the GPU fed across the step
        boundary (the synchronous form idles the device from the last kernel of a step to the first of the next)."""
        set_training(self.model, True)                      # self.model.train(): training.py:150, as train_step
        self._broadcast_buffers()                           # (DDP's per-forward buffer broadcast, as train_step)
        if self.cfg["training"].get("stage", "stage1") != "stage1":
            raise NotImplementedError("only training.stage == 'stage1' exists in the reference (training.py:152)")
        return self.train_stage1_frames(batch, it=it, seed=seed, wait=wait)

    def train_stage1_frames(self, batch, it=None, seed=None, wait: bool = True):
        """`train_steps` without the `model.train()` of training.py:150 -- the K-frame counterpart of `train_stage1`: the
        post-fusion U-Net runs in whatever mode it is in."""
        import random
        tc, m, dev = self.cfg["training"], self.model, self.device
        if self.optimizer is None:
            raise ValueError("train_steps steps an optimizer: construct Trainer(model, optimizer=...)")
        if bool(tc.get("use_canonical_depth_loss_photo_v2", False)):
            raise NotImplementedError("train_steps: the canonical-depth photo loss (use_canonical_depth_loss_photo_v2) runs through train_step only")
        if isinstance(batch, dict):
            K = int(batch["rgb"].shape[0])
            frames = [{k: (v[i] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == K else v) for k, v in batch.items()}
                      for i in range(K)]
        else:
            frames = list(batch)
            K = len(frames)
        if K == 0:
            raise ValueError("train_steps: empty batch")
        on = lambda t: t.to(dev, non_blocking=True) if isinstance(t, torch.Tensor) else t
        scalar = lambda v: int(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else int(v)
        f0 = frames[0]
        H, W = int(f0["rgb"].shape[-3]), int(f0["rgb"].shape[-2])
        if self.batch_rays != H * W:
            raise NotImplementedError("train_steps: batch_rays = H*W (the May configuration)")
        self.height, self.width = H, W
        x0, y0 = scalar(f0["lip_lefttop_x"]), scalar(f0["lip_lefttop_y"])
        for fr in frames[1:]:
            if (scalar(fr["lip_lefttop_x"]), scalar(fr["lip_lefttop_y"])) != (x0, y0) or fr["rgb"].shape != f0["rgb"].shape:
                raise ValueError("train_steps: the frames of one step must come from one clip (same lip box and sizes)")
        sync_on = bool(self.use_syncloss and it is not None and it > 100000 and tc.get("stage", "stage1") == "stage1")
        face_on = bool(self.use_post_fusion)
        lip_perc = self.use_perceptual_loss and tc.get("use_lip_perc_loss", "v1") == "v1"
        face_perc = self.use_perceptual_loss and tc.get("use_face_perc_loss", True) is True
        if tc.get("use_lip_photo_loss", "v1") != "v1" or (face_on and tc.get("use_face_photo_loss", True) is not True) or \
                (self.use_perceptual_loss and face_on and lip_perc != face_perc):
            raise NotImplementedError("train_steps implements the May loss set (lip + face photometric terms, LPIPS on both or neither)")
        key = (H, W, self.precision, sync_on, face_on, bool(lip_perc))
        if self._stage_step is None or self._stage_step[0] != key:
            self._stage_step = (key, StageOneStep(m, H, W, syncnet=self.syncnet if sync_on else None, precision=self.precision,
                                                  w_syncloss=self.w_syncloss, lambda_rgb=self.w_photometric_loss,
                                                  w_post_fusion=self.w_post_fusion, face_loss=face_on,
                                                  perceptual=self.perceptual_loss_fn if lip_perc else None,
                                                  w_perceptual_loss=self.w_perceptual_loss))
        step = self._stage_step[1]
        # ---- the random draws, frame by frame in train_step's order: u01 of the main render (device generator, training.py:200),
        # the black-hole coin (python's `random`, tf_nerf.py:371), its two noise fields, then the window's five u01 draws
        u_main, u_win, n1s, n2s = [], [], [], []
        ones = None
        for fr in frames:
            u_main.append(torch.rand(1, device=dev))
            if face_on:
                if random.random() > 0.5:
                    a_, b_ = m.draw_hole_noise(fr["rgb_face_ori"].reshape(1, *fr["rgb_face_ori"].shape[-3:]), device=dev)
                else:      # a field of ones punches no hole: noise >= 1e-6 everywhere == the branch not taken
                    if ones is None:
                        ones = torch.ones(1, *fr["rgb_face_ori"].shape[-3:-1], device=dev)
                    a_ = b_ = ones
                n1s.append(a_)
                n2s.append(b_)
            if sync_on:
                u_win.append(torch.cat([torch.rand(1, device=dev) for _ in range(int(fr["audio_window"].shape[-3]))]))
        stack = lambda k: torch.stack([on(fr[k]).reshape(fr[k].shape[-3:] if fr[k].dim() > 3 else fr[k].shape) for fr in frames], 0)
        audio = torch.stack([on(fr["audio"]).reshape(16, 29) for fr in frames], 0)
        first = [scalar(fr["index"]) for fr in frames]      # (unseeded: the window's clamp sees the data index, the seed is added after it)
        targets = torch.stack([on(fr["rgb"]).reshape(H * W, 3) for fr in frames], 0)
        face = sync = None
        if face_on:
            face = dict(rgb_face_canonical=on(f0["rgb_face_zero"]).reshape(1, *f0["rgb_face_zero"].shape[-3:]),
                        rgb_face_gt=stack("rgb_face_ori"), mask_lip_canonical=on(f0["mask_lip_canonical"]).reshape(1, *f0["mask_lip_canonical"].shape[-3:]),
                        lip_lefttop_x=x0, lip_lefttop_y=y0, coord=stack("coord"), hole_noise=(torch.cat(n1s, 0), torch.cat(n2s, 0)))
        if sync_on:
            T = int(f0["audio_window"].shape[-3])
            bbox = f0["canonical_face_bbox"][0] if isinstance(f0["canonical_face_bbox"], torch.Tensor) and f0["canonical_face_bbox"].dim() > 1 \
                else f0["canonical_face_bbox"]
            sync = dict(audio_window=torch.stack([on(fr["audio_window"]).reshape(T, 16, 29) for fr in frames], 0),
                        u01=torch.stack(u_win, 0), total_frame=scalar(f0["total_frame"]),
                        rgb_face_canonical=on(f0["rgb_face_zero"]).reshape(1, *f0["rgb_face_zero"].shape[-3:]), rgb_face_gt=stack("rgb_face_ori"),
                        mask_lip_canonical=on(f0["mask_lip_canonical"]).reshape(1, *f0["mask_lip_canonical"].shape[-3:]),
                        lip_lefttop_x=x0, lip_lefttop_y=y0,
                        coord_window=torch.stack([on(fr["coord_window"]).reshape(T, *fr["coord_window"].shape[-3:]) for fr in frames], 0),
                        canonical_face_bbox=[float(v) for v in bbox], mel=torch.stack([on(fr["mel"]).reshape(1, 80, 16) for fr in frames], 0),
                        rgb_window_neg=torch.stack([on(fr["rgb_window_neg"]).reshape(3, T, 96, 96) for fr in frames], 0))
        self.optimizer.zero_grad()
        total, grads, aux = step.loss_and_grads(audio, first, targets, torch.cat(u_main), sync=sync, face=face,
                                                seed=0 if seed is None else int(seed))
        apply_grads(m, grads)
        self._average_gradients_over_ranks()
        nan_check = self._step_optimizer()
        loss = {"loss": total, "loss_rgb": aux["loss_rgb"] + (aux["loss_face"] if "loss_face" in aux else 0)}
        for k in ("loss_perceptual", "loss_sync"):
            if k in aux:
                loss[k] = aux[k]
        pending = PendingStep(self, nan_check, loss)
        return pending.result() if wait else pending


class PendingStep:
    """A training step that has been queued (`Trainer.train_steps(..., wait=False)`): `result()` waits for it once, issues the reference's
    NaN warnings (check_weights, training.py:572) and returns (loss_rgb, loss dict) with host tensors, like the synchronous call."""

    def __init__(self, trainer, nan_check, loss):
        self._trainer, self._nan_check, self._loss, self._out = trainer, nan_check, loss, None

    def result(self):
        if self._out is None:
            self._trainer._check_weights_report(self._nan_check)
            self._trainer._loss_to_host(self._loss)      # host tensors, like `_train_stage1` and the reference (training.py:562-569)
            self._out = (self._loss["loss_rgb"], self._loss)
            self._nan_check = None
        return self._out


# ----------------------------------------------------------------------------------------------------------------------
class MlpState:
    """What one forward of the MLP leaves behind for its backward: fp32 mode x [N,128] + hsave [8,N,256]; bf16 mode the
    operand image xT, the activation images hT and the ReLU mask words."""

    def __init__(self, precision, N, dev, lib):
        self.precision, self.N = precision, N
        if precision == "bf16":
            self.Np = int(lib.s2l_bf16_rows_padded(N))
            i16 = lambda n: torch.empty(n, dtype=torch.int16, device=dev)
            self.hT, self.xT = i16(8 * self.Np * 256), i16(self.Np * 128)
            self.masks = torch.empty(8 * (self.Np // 64) * 256, dtype=torch.int64, device=dev)
        else:
            self.x, self.hsave = _f(dev, N, 128), _f(dev, 8, N, 256)


def mlp_forward(model: TalkingFace, st: MlpState, rgb: torch.Tensor, stream) -> None:
    """rows already embedded in st.x / st.xT -> rgb [N,3], activations kept in `st`."""
    lib, ck = _abi.load(), _abi.check
    packed = model.packed_weights()
    if st.precision == "bf16":
        ck(lib.s2l_train_forward_bf16(_ptr(model.packed_weights_bf16()), _ptr(packed), _ptr(st.xT), _ptr(st.hT), _ptr(st.masks),
                                      _ptr(rgb), st.N, stream), "s2l_train_forward_bf16")
    else:
        ck(lib.s2l_train_forward(_ptr(packed), _ptr(st.x), _ptr(st.hsave), _ptr(rgb), st.N, stream), "s2l_train_forward")


def mlp_backward(model: TalkingFace, st: MlpState, drgb: torch.Tensor, stream, tile_sums: bool = False):
    """d loss / d rgb [N,3] -> ({state-dict name: gradient} for the 30 MLP tensors, dxa [N,64] = gradient of the audio columns).
    dz chain, weight-gradient GEMMs, and the un-fold of the pack-time folds (s2l_unfold_first_layer) -- all HIP.
    tile_sums (bf16 mode): the generated-assembly backward kernel, whose audio gradient comes back summed over each 256-row tile
    ([ceil(N / 256), 64]) -- for callers whose frames are whole numbers of tiles."""
    lib, ck = _abi.load(), _abi.check
    packed = model.packed_weights()
    dev, N = packed.device, st.N
    bf16 = st.precision == "bf16"
    if bf16:
        pb = model.packed_weights_bf16()
        lay = st.Np * 256
        dzT = torch.empty(8 * lay, dtype=torch.int16, device=dev)
        if tile_sums:
            dxa = _f(dev, st.Np // 256, 64)
            ck(lib.s2l_train_backward_bf16_tiles(_ptr(pb), _ptr(drgb), _ptr(st.masks), _ptr(dzT), _ptr(dxa), N, stream),
               "s2l_train_backward_bf16_tiles")
        else:
            dxa = _f(dev, N, 64)
            ck(lib.s2l_train_backward_bf16(_ptr(pb), _ptr(drgb), _ptr(st.masks), _ptr(dzT), _ptr(dxa), N, stream),
               "s2l_train_backward_bf16")
        work = _f(dev, int(lib.s2l_wgrad_bf16_work_floats()))

        def wgrad(k, inp, k_in, want_bias=True):          # dz of layer k against hT[inp] or the x tiles
            out, db = _f(dev, 256, k_in), (_f(dev, 256) if want_bias else None)
            src = st.xT if inp is None else st.hT[inp * lay:]
            ck(lib.s2l_wgrad_bf16(_ptr(dzT[k * lay:]), _ptr(src), k_in, _ptr(work), _ptr(out), _ptr(db), N, stream), "s2l_wgrad_bf16")
            return out, db
    else:
        dxa = _f(dev, N, 64)
        dzsave = _f(dev, 8, N, 256)
        ck(lib.s2l_train_backward(_ptr(packed), _ptr(drgb), _ptr(st.hsave), _ptr(dzsave), _ptr(dxa), N, stream), "s2l_train_backward")
        work = _f(dev, int(lib.s2l_split_work_floats(256 * 256)))

        def wgrad(k, inp, k_in, want_bias=True):
            out, db = _f(dev, 256, k_in), (_f(dev, 256) if want_bias else None)
            src, ldin = (st.x, 128) if inp is None else (st.hsave[inp], 256)
            ck(lib.s2l_wgrad(_ptr(dzsave[k]), 256, _ptr(src), ldin, k_in, _ptr(work), _ptr(out), _ptr(db), N, stream), "s2l_wgrad")
            return out, db

    g = {}
    dw5b = dc5 = None
    for k in range(1, 8):                          # pts_linears[k]: h_{k-1} -> h_k
        dw, db = wgrad(k, k - 1, 256)
        if k == 5:
            dw5b, dc5 = dw, db
        else:
            g[f"pts_linears.{k}.weight"], g[f"pts_linears.{k}.bias"] = dw, db
    dG0, dc0 = wgrad(0, None, 128)
    dG5, _ = wgrad(5, None, 128, want_bias=False)
    dwout, dbout = _f(dev, 3, 256), _f(dev, 3)
    if bf16:
        ck(lib.s2l_out_grad_bf16(_ptr(drgb), _ptr(st.hT[7 * lay:]), _ptr(work), _ptr(dwout), _ptr(dbout), N, stream), "s2l_out_grad_bf16")
    else:
        ck(lib.s2l_small_outer(_ptr(drgb), 3, 3, _ptr(st.hsave[7]), 256, 256, _ptr(work), _ptr(dwout), N, stream), "s2l_small_outer")
        ck(lib.s2l_small_outer(None, 0, 1, _ptr(drgb), 3, 3, _ptr(work), _ptr(dbout), N, stream), "s2l_small_outer")
    g["output_linear.weight"], g["output_linear.bias"] = dwout, dbout

    # un-fold G0 = W0 [Wuv|Wa|Wt], c0 = W0 (buv+ba+bt) + b0 (and the skip twins)
    sd = param_map(model)
    w = lambda name: _dev_f32(sd[name], dev, name)

    def unfold(first, ld, names, dG, dc, right):
        d_first = _f(dev, 256, ld)
        outs = [_f(dev, 256, 42), _f(dev, 256, 64), _f(dev, 256, 20), _f(dev, 256)]
        ck(lib.s2l_unfold_first_layer(_ptr(dG), _ptr(dc), _ptr(w(first)), ld, *[_ptr(w(f"{n}.weight")) for n in names],
                                      *[_ptr(w(f"{n}.bias")) for n in names], _ptr(right), _ptr(d_first), *[_ptr(o) for o in outs],
                                      stream), "s2l_unfold_first_layer")
        for n, o in zip(names, outs[:3]):
            g[f"{n}.weight"], g[f"{n}.bias"] = o, outs[3].clone()
        return d_first

    g["pts_linears.0.weight"] = unfold("pts_linears.0.weight", 256, ("fc_uv", "fc_audio", "fc_time"), dG0, dc0, None)
    g["pts_linears.0.bias"] = dc0
    g["pts_linears.5.weight"] = unfold("pts_linears.5.weight", 512, ("fc_uv_skip", "fc_audio_skip", "fc_time_skip"), dG5, dc5, dw5b)
    g["pts_linears.5.bias"] = dc5
    return g, dxa


def audio_backward(model: TalkingFace, audio32: torch.Tensor, dfeat: torch.Tensor, stream):
    """d loss / d feat [B,64] -> gradients of the 12 audio-encoder tensors (autograd of tf_nerf.py:197-213)."""
    lib = _abi.load()
    dev = dfeat.device
    B = audio32.shape[0]
    na = int(lib.s2l_audio_grad_floats())
    awork, agrads = _f(dev, ((B + 3) // 4) * na), _f(dev, na)
    _abi.check(lib.s2l_audio_backward(_ptr(model.packed_weights()), _ptr(audio32), _ptr(dfeat), _ptr(awork), _ptr(agrads), B, stream),
               "s2l_audio_backward")
    params, g, off = param_map(model), {}, 0
    for name in AUDIO_TENSORS:
        p_ = params[name]
        g[name] = agrads[off:off + p_.numel()].reshape(p_.shape)
        off += p_.numel()
    return g


class LipTrainStep:
    """Forward + backward of the lip-MLP training objective (BASELINE config 5) for a batch of frames:

        pred[f] = predict_lip_image(frame f)     4-tap local ensemble, training.py:158-251
        grads   = d loss / d (all 42 hot-path tensors)  given  d loss / d pred

    as hand-written HIP kernels: rows -> forward with saved activations -> ensemble reduce | ensemble backward -> dz
    chain -> weight-gradient GEMMs -> un-fold -> audio-encoder backward.  `loss_and_grads` closes the loop with the
    photometric loss (training.py:605-619).  Gradients are keyed by the reference's state-dict names."""

    def __init__(self, model: TalkingFace, height: int, width: int, precision: str = "fp32"):
        """precision: 'fp32' (parity mode: exact-fp32 MFMA, saved state in fp32) or 'bf16' (BASELINE config 5: bf16 MFMA
        operands and saved state, fp32 accumulation / master weights / gradients; csrc/train_bf16.hip)."""
        from .rendering import shared_coords
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.precision = precision
        self.model, self.h, self.w = model, int(height), int(width)
        self.lib = _abi.load()
        self.coords = shared_coords(width, height, model.packed_weights().device)      # (read only)
        self.bf16_backward_kernel = "asm"      # "cpp": always the C++ kernel (per-row audio gradient); same dz images bit for bit
        self._ctx = None

    def forward(self, audio, frame_idx, u01) -> torch.Tensor:
        """audio [B,16,29], frame indices [B], U(0,1) draws [B] (training.py:200) -> pred [B,HW,3]; keeps what
        `backward` needs until the next forward."""
        lib, m, ck = self.lib, self.model, _abi.check
        packed = m.packed_weights()
        dev = packed.device
        a32 = _dev_f32(audio, dev, "audio")
        B, P = a32.shape[0], self.h * self.w
        N = 4 * P * B
        idx = [int(i) for i in (frame_idx.tolist() if isinstance(frame_idx, torch.Tensor) else frame_idx)]
        if isinstance(u01, torch.Tensor) and u01.is_cuda:      # draws that never left the device (Trainer.train_steps): no round trip
            t_u = u01.detach().to(dev, torch.float32).reshape(-1).contiguous()
            n_u = t_u.numel()
        else:
            u = [float(v) for v in (u01.tolist() if isinstance(u01, torch.Tensor) else u01)]
            t_u, n_u = torch.tensor(u, dtype=torch.float32).to(dev, non_blocking=True), len(u)
        if len(idx) != B or n_u != B:
            raise ValueError("frame_idx and u01 need one entry per audio window")
        feat = m._audio_encode(a32)                                          # [B,64]
        st = MlpState(self.precision, N, dev, lib)
        areas, rgb, pred = _f(dev, N), _f(dev, N, 3), _f(dev, B * P, 3)
        t_idx = torch.tensor(idx, dtype=torch.int64).to(dev, non_blocking=True)
        with torch.cuda.device(dev):
            s = _stream()
            if self.precision == "bf16":     # the embedded rows of the whole batch in one launch, straight to the bf16 operand image
                ck(lib.s2l_ensemble_rows_bf16(_ptr(packed), _ptr(self.coords), _ptr(feat), _ptr(t_idx), _ptr(t_u), self.w, self.h,
                                              _ptr(st.xT), _ptr(areas), P, B, s), "s2l_ensemble_rows_bf16")
            else:                            # rows of frame b: [b*4P, (b+1)*4P), tap-major inside
                ck(lib.s2l_ensemble_rows_batch(_ptr(packed), _ptr(self.coords), _ptr(feat), _ptr(t_idx), _ptr(t_u), self.w, self.h,
                                               _ptr(st.x), _ptr(areas), P, B, s), "s2l_ensemble_rows_batch")
            mlp_forward(m, st, rgb, s)
            ck(lib.s2l_ensemble_reduce_batch(_ptr(rgb), _ptr(areas), _ptr(pred), P, B, s), "s2l_ensemble_reduce_batch")
        self._ctx = (st, areas, a32, B, P)
        return pred.reshape(B, P, 3)

    def backward(self, dpred: torch.Tensor):
        """d loss / d pred [B,HW,3] -> ({name: gradient} for all 42 tensors, {'d_audio_feat': [B,64]})."""
        if self._ctx is None:
            raise RuntimeError("LipTrainStep.backward needs a preceding forward")
        lib, m, ck = self.lib, self.model, _abi.check
        st, areas, a32, B, P = self._ctx
        dev = areas.device
        dp = _dev_f32(dpred, dev, "dpred").reshape(B * P, 3)
        drgb = _f(dev, st.N, 3)
        with torch.cuda.device(dev):
            s = _stream()
            ck(lib.s2l_ensemble_backward_batch(_ptr(dp), _ptr(areas), _ptr(drgb), P, B, s), "s2l_ensemble_backward_batch")
            # bf16 mode, frames that are whole numbers of 256-row tiles (96x96, 64x64, 128x128 ...): the assembly backward kernel,
            # which sums the audio gradient per tile itself; otherwise the per-row form
            tiles = self.precision == "bf16" and (4 * P) % 256 == 0 and self.bf16_backward_kernel == "asm"
            g, dxa = mlp_backward(m, st, drgb, s, tile_sums=tiles)
            # per-frame gradient of the audio feature (rows of frame b are contiguous), then the encoder backward
            da, swork = _f(dev, B, 64), _f(dev, B * 32 * 64)
            ck(lib.s2l_segment_colsums(_ptr(dxa), 64, 64, (4 * P) // 256 if tiles else 4 * P, B, _ptr(swork), _ptr(da), s), "s2l_segment_colsums")
            g.update(audio_backward(m, a32, da, s))
        self._ctx = None
        return g, {"d_audio_feat": da}

    def loss_and_grads(self, audio, frame_idx, targets, u01, weight: float = 1.0):
        """loss = weight * mean_{frames, pixels, rgb} (pred - target)^2 and its gradients."""
        pred = self.forward(audio, frame_idx, u01)
        dev = pred.device
        tgt = _dev_f32(targets, dev, "targets").reshape(pred.shape)
        dpred, loss, mwork = torch.empty_like(pred), _f(dev, 1), _f(dev, 1024)
        with torch.cuda.device(dev):
            _abi.check(self.lib.s2l_mse(_ptr(pred), _ptr(tgt), ctypes.c_float(weight), _ptr(dpred), _ptr(mwork), _ptr(loss),
                                        pred.numel(), _stream()), "s2l_mse")
        g, aux = self.backward(dpred)
        aux["pred"] = pred
        return loss, g, aux


def apply_grads(model: TalkingFace, grads) -> None:
    """Install the gradients returned by `LipTrainStep` as `.grad` of the matching parameters, so a stock optimizer (the
    reference uses Adam(lr=1e-4), train.py:128) can step."""
    params = param_map(model)
    for name, g in grads.items():
        p = params[name]
        g = g.reshape(p.shape)
        p.grad = (g if g.dtype == p.dtype else g.to(p.dtype)).contiguous()


# ----------------------------------------------------------------------------------------------------------------------
class SyncChain:
    """The path between the rendered lips of a 5-frame window and the lip-sync expert's loss (training.py:491-557):

        lip [S*T,h,w,3] -> paste + head-pose warp against the MAIN frame's observed image (post_fusion2_onlylip, :527-536)
                        -> frozen eval-mode post-fusion U-Net (its first return value is rgb_recon, tf_nerf.py:387-389)
                        -> crop to data['canonical_face_bbox'] + Resize([96,96]) (:541-544) -> rgb_window [S,3,T,96,96]
                        -> get_sync_contrastive_loss(mel, rgb_window, rgb_window_neg) * w_syncloss (:551-552)

    and its adjoint back to d loss / d lip.  Samples go through in groups that bound the U-Net state (about 1 GB per
    500x500 frame for the saved activations and the backward scratch)."""

    UNET_RADIUS = 40   # pixels: >= the U-Net's dependency radius (32: 2 + 4 + 8 down, 4 + 4 + 2 + 2 up, pooling alignment) + slack

    def __init__(self, model: TalkingFace, syncnet, syncnet_T: int = 5, w_syncloss: float = 0.01, out_hw=(96, 96),
                 max_frames_per_group: int = 40, window: bool = True, unet_precision: str = "fp32", loss_conv_precision: str = None):
        from .syncnet import SyncLoss
        if getattr(model, "post_fusion_unet", None) is None:
            raise ValueError("SyncChain needs model.use_post_fusion (the window is the U-Net's output)")
        self.model, self.T, self.w = model, int(syncnet_T), float(w_syncloss)
        self.sync = SyncLoss(syncnet, syncnet_T, loss_conv_precision)      # (None: the SyncNet module's own `conv_precision`)
        self.out_hw = (int(out_hw[0]), int(out_hw[1]))
        self.group = max(1, int(max_frames_per_group) // self.T)
        # window=True: the U-Net runs on the canonical-face box dilated by its dependency radius, not on the whole frame -- the
        # sync loss reads nothing else of its output (training.py:541-544) and every value it reads, and every gradient that
        # comes back, is bit-identical to the full-frame evaluation (s2l_unet_forward_saved_window)
        self.window = bool(window)
        self.unet_param_grads = None      # a dict: train-mode U-Net parameter gradients are accumulated into it (StageOneStep sets it)
        self.unet_precision = unet_precision      # "bf16": the frozen U-Net's 3x3 convolutions on the bf16 MFMA (fp32 accumulation)

    def unet_window(self, bbox, FH: int, FW: int):
        """(x0, y0, x1, y1) of the crop the U-Net runs on: the box dilated by UNET_RADIUS, on the 4-pixel grid of the two pooling
        levels, clipped to the frame; the whole frame with window=False."""
        if not self.window:
            return 0, 0, FW, FH
        x, y, x2, y2 = (int(v) for v in list(bbox)[:4])
        r = self.UNET_RADIUS
        return (max(0, (x - r) // 4 * 4), max(0, (y - r) // 4 * 4), min(FW, -(-(x2 + r) // 4) * 4), min(FH, -(-(y2 + r) // 4) * 4))

    def loss_and_dlip(self, lips, rgb_face_canonical, rgb_face_gt, mask_lip_canonical, lip_lefttop_x, lip_lefttop_y,
                      coord_window, canonical_face_bbox, mel, rgb_window_neg):
        """lips [S*T,h,w,3] (sample-major: frame s*T + t); rgb_face_canonical, mask_lip_canonical [1,FH,FW,3] (per clip);
        rgb_face_gt [S,FH,FW,3] (the main frame of each sample); coord_window [S,T,FH,FW,2]; canonical_face_bbox
        (x, y, x2, y2[, score]); mel [S,1,80,16]; rgb_window_neg [S,3,T,96,96].
        Returns (loss = w_syncloss * mean over samples, d loss / d lips, rgb_window [S,3,T,96,96])."""
        lib, m, ck, T = _abi.load(), self.model, _abi.check, self.T
        dev = m.packed_weights().device
        lips = _dev_f32(lips, dev, "lips")
        S = lips.shape[0] // T
        if lips.shape[0] != S * T or S == 0:
            raise ValueError(f"lips must hold {T} frames per sample")
        gt = _dev_f32(rgb_face_gt, dev, "rgb_face_gt")
        cw = _dev_f32(coord_window, dev, "coord_window")
        FH, FW = gt.shape[1], gt.shape[2]
        if gt.shape[0] != S or cw.shape != (S, T, FH, FW, 2):
            raise ValueError("rgb_face_gt must be [S,FH,FW,3] and coord_window [S,T,FH,FW,2]")
        x, y, x2, y2 = (int(v) for v in list(canonical_face_bbox)[:4])
        x2, y2 = min(x2, FW), min(y2, FH)          # rgb_merged[:, y:y2, x:x2] clips a box that leaves the frame (training.py:541-543)
        oh, ow = self.out_hw
        unet = m.post_fusion_unet
        # the sub-module's own mode decides, as in post_fusion2_onlylip: a net left in train mode (the reference's loop: train_step's
        # self.model.train(), training.py:150) normalises every frame with its own statistics -- whole frames, exact fp32
        use_window = self.window and not unet.training
        wx0, wy0, wx1, wy1 = self.unet_window((x, y, x2, y2), FH, FW) if use_window else (0, 0, FW, FH)
        win = (FH, FW, wy0, wx0) if not unet.training else None
        mel = _dev_f32(mel, dev, "mel")
        neg = _dev_f32(rgb_window_neg, dev, "rgb_window_neg")
        d_lips = torch.empty_like(lips)
        window = _f(dev, S, 3, T, oh, ow)
        total = torch.zeros((), dtype=torch.float32, device=dev)
        for s0 in range(0, S, self.group):
            s1 = min(S, s0 + self.group)
            n, fr = s1 - s0, slice(s0 * T, s1 * T)
            gt_f = gt[s0:s1].repeat_interleave(T, dim=0)                      # each window frame sees its sample's main frame
            coord_f = cw[s0:s1].reshape(n * T, FH, FW, 2)
            new, _ = m.composite_clip(lips[fr], rgb_face_canonical, gt_f, mask_lip_canonical, lip_lefttop_x, lip_lefttop_y, coord_f)
            crop = new[:, wy0:wy1, wx0:wx1].contiguous() if use_window else new
            ch, cw_ = crop.shape[1], crop.shape[2]
            recon, saved = unet.forward_for_backward(crop, window=win, precision=self.unet_precision)
            wnd = window[s0:s1]
            with torch.cuda.device(dev):
                ck(lib.s2l_crop_resize(_ptr(recon), ch, cw_, x - wx0, y - wy0, x2 - wx0, y2 - wy0, _ptr(wnd), oh, ow, T, n * T, _stream()),
                   "s2l_crop_resize")
            # BCE is a mean over the batch: this group's share of the mean over all S samples
            loss, d_win = self.sync.get_sync_contrastive_loss(mel[s0:s1], wnd, neg[s0:s1], weight=self.w * n / S, want_grad=True)
            total = total + loss
            d_recon = torch.empty_like(recon)
            with torch.cuda.device(dev):
                ck(lib.s2l_crop_resize_backward(_ptr(d_win), ch, cw_, x - wx0, y - wy0, x2 - wx0, y2 - wy0, _ptr(d_recon), oh, ow, T, n * T,
                                                _stream()), "s2l_crop_resize_backward")
            d_new = unet.backward_to_input(saved, d_recon, self.unet_param_grads)
            del saved, recon, d_recon
            if use_window:   # the gradient is exactly zero outside the crop (its support is the box dilated by <= 32 pixels)
                full = torch.zeros(n * T, FH, FW, 3, dtype=torch.float32, device=dev)
                full[:, wy0:wy1, wx0:wx1] = d_new
                d_new = full
            d_lips[fr] = m.composite_backward_lip(d_new, rgb_face_canonical, mask_lip_canonical, lip_lefttop_x, lip_lefttop_y,
                                                  coord_f, lips.shape[1], lips.shape[2])
        return total, d_lips, window


class StageOneStep:
    """One optimisation step's loss and gradients as the reference forms them after `it > 100000` (train_stage1,
    training.py:347-574, May flags):

        loss = lambda_rgb * MSE(lip)  [+ w_perceptual * LPIPS(lip)]                          (:417-421, `perceptual`)
             [+ w_post_fusion * (MSE + w_perceptual * LPIPS)(U-Net(composite_blackaug(lip)), rgb_face_ori)]   (:436-459, `face_loss`)
             + w_syncloss * sync_contrastive(window of T more renders -> composite -> U-Net -> crop/resize)   (:491-557)

    for B main frames (each a sample) of which the first S carry a sync window.  All frames -- B main + S*T window -- go
    through ONE batched LipTrainStep forward/backward."""

    def __init__(self, model: TalkingFace, height: int, width: int, syncnet=None, precision: str = "bf16", syncnet_T: int = 5,
                 w_syncloss: float = 0.01, lambda_rgb: float = 1.0, w_post_fusion: float = 1.0, face_loss: bool = False,
                 perceptual=None, w_perceptual_loss: float = 0.01):
        """perceptual: a speech2lip_amd.LPIPS (training.py:76) or None; its terms are means over the B main frames, like the MSE."""
        self.perceptual, self.w_perc = perceptual, float(w_perceptual_loss)
        self.model, self.h, self.w = model, int(height), int(width)
        self.step = LipTrainStep(model, height, width, precision)
        # precision "bf16" (BASELINE config 5): bf16 operands in the MLP kernels AND in the frozen U-Net's 3x3 convolutions
        self.unet_precision = "bf16" if precision == "bf16" else "fp32"
        # ... and hi + lo bf16 operands (three bf16 MFMAs per product, ~1e-5 relative) in the frozen loss nets' convolutions
        self.loss_conv_precision = "split" if precision == "bf16" else "fp32"
        self.chain = SyncChain(model, syncnet, syncnet_T, w_syncloss, unet_precision=self.unet_precision,
                               loss_conv_precision=self.loss_conv_precision) if syncnet is not None else None
        self.T, self.lambda_rgb, self.w_post_fusion, self.face_loss = int(syncnet_T), float(lambda_rgb), float(w_post_fusion), face_loss

    def loss_and_grads(self, audio, frame_idx, targets, u01, sync=None, face=None, seed: int = 0):
        """audio [B,16,29], frame_idx [B], targets [B,HW,3], u01 [B]: the main frames.  `seed` is added to every frame index the
        renderer sees AFTER the window's clamp to the last frame, as the reference does (training.py:515-518 clamp `index + t`,
        then predict_lip_image adds the seed, :183-186).
        sync (optional): dict(audio_window [S,T,16,29], u01 [S,T], total_frame, rgb_face_canonical, rgb_face_gt [S,...],
        mask_lip_canonical, lip_lefttop_x, lip_lefttop_y, coord_window [S,T,FH,FW,2], canonical_face_bbox, mel, rgb_window_neg)
        -- sample s belongs to main frame s.
        face (optional, `face_loss`): dict(rgb_face_canonical, rgb_face_gt [B,...], mask_lip_canonical, lip_lefttop_x,
        lip_lefttop_y, coord [B,FH,FW,2], hole_noise=None | (n1, n2) [B,FH,FW])."""
        lib, m, ck = _abi.load(), self.model, _abi.check
        dev = m.packed_weights().device
        a = _dev_f32(audio, dev, "audio")
        B, P = a.shape[0], self.h * self.w
        idx = [int(i) for i in (frame_idx.tolist() if isinstance(frame_idx, torch.Tensor) else frame_idx)]
        on_device = isinstance(u01, torch.Tensor) and u01.is_cuda      # (Trainer.train_steps: the draws stay on the device)
        u = u01.detach().float().reshape(-1) if on_device else [float(v) for v in (u01.tolist() if isinstance(u01, torch.Tensor) else u01)]
        S = 0
        if sync is not None:
            if self.chain is None:
                raise ValueError("StageOneStep was built without a syncnet")
            aw = _dev_f32(sync["audio_window"], dev, "audio_window")
            S, T = aw.shape[0], aw.shape[1]
            if T != self.T or S > B:
                raise ValueError("audio_window must be [S<=B,T,16,29]")
            total = int(sync["total_frame"])
            if on_device:
                u = torch.cat([u, sync["u01"].detach().to(dev, torch.float32).reshape(-1)])
            else:
                uw = sync["u01"].tolist() if isinstance(sync["u01"], torch.Tensor) else sync["u01"]
            for s in range(S):              # cur_data['index'] = index + t, clamped to the last frame (training.py:515-518)
                for t in range(T):
                    idx.append(idx[s] + t if idx[s] + t < total else total - 1)
                    if not on_device:
                        u.append(float(uw[s][t]))
            a = torch.cat([a, aw.reshape(S * T, 16, 29)], 0)
        if seed:
            idx = [i + int(seed) for i in idx]
        pred = self.step.forward(a, idx, u)                                   # [B + S*T, P, 3]
        dpred = torch.zeros_like(pred)
        losses = {}
        # a post-fusion U-Net that is in train mode AND still trains (the reference until it > 100000) gets its parameter gradients
        # too, under `post_fusion_unet.<name>`; a frozen one in train mode (the reference's loop afterwards, see SyncChain) only
        # passes the input gradient on
        unet = getattr(m, "post_fusion_unet", None)
        unet_grads = {} if (unet is not None and unet.training and any(p.requires_grad for p in unet.parameters())) else None
        if self.chain is not None:
            self.chain.unet_param_grads = unet_grads
        tgt = _dev_f32(targets, dev, "targets").reshape(B, P, 3)
        loss, mwork = _f(dev, 1), _f(dev, 1024)
        with torch.cuda.device(dev):
            ck(lib.s2l_mse(_ptr(pred[:B]), _ptr(tgt), ctypes.c_float(self.lambda_rgb), _ptr(dpred[:B]), _ptr(mwork), _ptr(loss),
                           B * P * 3, _stream()), "s2l_mse")
        losses["loss_rgb"] = loss[0]
        total_loss = loss[0]
        if self.perceptual is not None:          # training.py:420-421 (use_lip_perc_loss 'v1')
            d, st = self.perceptual.distance_nhwc(pred[:B].reshape(B, self.h, self.w, 3), tgt.reshape(B, self.h, self.w, 3), from01=True,
                                                  keep=True, precision=self.loss_conv_precision)
            self.perceptual.backward_nhwc(st, torch.full((B,), self.w_perc / B, device=dev), out=dpred[:B].view(B, self.h, self.w, 3))
            losses["loss_perceptual"] = d.mean() * self.w_perc
            total_loss = total_loss + losses["loss_perceptual"]
        if face is not None:
            if not self.face_loss:
                raise ValueError("pass face_loss=True to StageOneStep to use the face photometric term")
            lip = pred[:B].reshape(B, self.h, self.w, 3)
            holes = face.get("hole_noise")
            gt = _dev_f32(face["rgb_face_gt"], dev, "rgb_face_gt")
            args = (face["rgb_face_canonical"], face["mask_lip_canonical"], face["lip_lefttop_x"], face["lip_lefttop_y"], face["coord"])
            new, _ = m.composite_clip(lip, args[0], gt, args[1], args[2], args[3], args[4], hole_noise=holes)
            recon, saved = m.post_fusion_unet.forward_for_backward(new, precision=self.unet_precision)
            d_recon, floss = torch.empty_like(recon), _f(dev, 1)
            with torch.cuda.device(dev):
                ck(lib.s2l_mse(_ptr(recon), _ptr(gt), ctypes.c_float(self.lambda_rgb * self.w_post_fusion), _ptr(d_recon), _ptr(mwork),
                               _ptr(floss), recon.numel(), _stream()), "s2l_mse")
            if self.perceptual is not None:      # training.py:453-456 (use_face_perc_loss; its mask is all ones)
                wp = self.w_perc * self.w_post_fusion
                d, st = self.perceptual.distance_nhwc(recon, gt, from01=True, keep=True, precision=self.loss_conv_precision)
                self.perceptual.backward_nhwc(st, torch.full((B,), wp / B, device=dev), out=d_recon)
                losses["loss_perceptual"] = losses["loss_perceptual"] + d.mean() * wp
                total_loss = total_loss + d.mean() * wp
            d_new = m.post_fusion_unet.backward_to_input(saved, d_recon, unet_grads)
            d_lip = m.composite_backward_lip(d_new, args[0], args[1], args[2], args[3], args[4], self.h, self.w, hole_noise=holes)
            dpred[:B] += d_lip.reshape(B, P, 3)
            losses["loss_face"] = floss[0]
            total_loss = total_loss + floss[0]
        if S:
            lips = pred[B:].reshape(S * self.T, self.h, self.w, 3)
            sl, d_lips, window = self.chain.loss_and_dlip(lips, sync["rgb_face_canonical"], sync["rgb_face_gt"], sync["mask_lip_canonical"],
                                                          sync["lip_lefttop_x"], sync["lip_lefttop_y"], sync["coord_window"],
                                                          sync["canonical_face_bbox"], sync["mel"], sync["rgb_window_neg"])
            dpred[B:] = d_lips.reshape(S * self.T, P, 3)
            losses["loss_sync"] = sl
            losses["rgb_window"] = window
            total_loss = total_loss + sl
        g, aux = self.step.backward(dpred)
        if unet_grads:
            g.update({"post_fusion_unet." + k: v for k, v in unet_grads.items()})
        aux.update(losses)
        aux["pred"] = pred
        return total_loss, g, aux
