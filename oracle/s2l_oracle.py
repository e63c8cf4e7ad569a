"""CPU ORACLE for the Speech2Lip lip-render hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()`, `tools/make_goldens.py` and `bench.py`'s
`cpu_baseline` leg may import this module.  `speech2lip_amd/` never does: the product path
runs hand-written HIP kernels through the C-ABI in `include/s2l_hip.h` and fails loudly
when that library is missing.

This is our own functional restatement (PyTorch-CPU ops, fp32 by default, fp64 on request)
of the reference algorithm, written from the maths in SURVEY.md §3.3/§8a.  Each function
cites the reference lines it follows.  Parity is PINNED (except `lpips_alex`, which restates a third-party package
that is not available here and says so in its header): `tools/make_goldens.py` imports
the reference itself (in the build container, where /root/reference exists), checks every
function here against it, and commits the resulting vectors under `tests/golden/`;
`tests/test_oracle_golden.py` re-checks this file against those vectors everywhere.

State dicts are plain {name: tensor} with the reference's state-dict key names
(`tf_nerf.py:91-172`).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

PAD_MODE_MAY = 0      # 'may' / 'macron' / 'obama_adnerf' / 'obama2_face_crop' paths (tf_nerf.py:345-348)
PAD_MODE_DEFAULT = 1  # everything else (tf_nerf.py:349-350): paste origin one pixel up-left


def to_sd(np_sd, dtype=torch.float32) -> SD:
    return {k: torch.as_tensor(v).to(dtype) for k, v in np_sd.items()}


# --------------------------------------------------------------------------- A1
def get_coords(width: int, height: int, dtype=torch.float32, device=None) -> torch.Tensor:
    """Row-major (u,v) pixel grid in [0,1], endpoints inclusive -> [H*W, 2].
    Reference: src/face_simple/rendering.py:9-28 (linspace x, linspace y, meshgrid ij,
    stack [u, v])."""
    x = torch.linspace(0.0, 1.0, width, dtype=dtype, device=device)
    y = torch.linspace(0.0, 1.0, height, dtype=dtype, device=device)
    u = x.unsqueeze(0).expand(height, width)
    v = y.unsqueeze(1).expand(height, width)
    return torch.stack([u, v], dim=-1).reshape(-1, 2).contiguous()


# --------------------------------------------------------------------------- A2
def embed_uv(uv: torch.Tensor, multires: int = 10) -> torch.Tensor:
    """[N,d] -> [N, d + 2*L*d]: [x, sin(1x), cos(1x), sin(2x), cos(2x), ..., cos(2^(L-1) x)],
    every block carrying all d channels.  Reference: tf_nerf.py:391-425 (Embedder,
    include_input, log-sampled bands 2**linspace(0, L-1, L) which are exact powers of two)."""
    blocks = [uv]
    for k in range(multires):
        f = float(2 ** k)
        blocks.append(torch.sin(uv * f))
        blocks.append(torch.cos(uv * f))
    return torch.cat(blocks, dim=-1)


# --------------------------------------------------------------------------- A3
def time_div_term(out_dims: int = 20, device=None) -> torch.Tensor:
    """fp32 constant of PositionalEncodingTime.__init__ (tf_nerf.py:431-432)."""
    return torch.exp(torch.arange(0, out_dims, 2, dtype=torch.float32, device=device) * -(math.log(10000.0) / out_dims))


def time_pe(index: int, out_dims: int = 20, dtype=torch.float32, device=None) -> torch.Tensor:
    """1-D [out_dims] encoding of ONE frame index: pe[0::2]=sin(i*div), pe[1::2]=cos(i*div).
    Reference: tf_nerf.py:434-442 (uses position[0] only, returns a vector that fc_time
    broadcasts over all pixels)."""
    div = time_div_term(out_dims, device).to(dtype)
    pos = torch.tensor(float(int(index)), dtype=torch.float32, device=device).to(dtype)
    pe = torch.zeros(out_dims, dtype=dtype, device=device)
    pe[0::2] = torch.sin(pos * div)
    pe[1::2] = torch.cos(pos * div)
    return pe


# --------------------------------------------------------------------------- A4
def audio_encode(sd: SD, windows: torch.Tensor) -> torch.Tensor:
    """DeepSpeech windows [B,16,29] -> audio feature [B,64].
    Reference: tf_nerf.py:197-213 with layers :91-109 -- permute to [B,29,16], four
    Conv1d(k3,s2,p1)+LeakyReLU(0.02) (T 16->8->4->2->1), squeeze, Linear+LeakyReLU+Linear."""
    x = windows.permute(0, 2, 1)
    for i in (0, 2, 4, 6):
        x = F.conv1d(x, sd[f"encoder_conv.{i}.weight"], sd[f"encoder_conv.{i}.bias"], stride=2, padding=1)
        x = F.leaky_relu(x, 0.02)
    x = x.squeeze(-1)
    x = F.leaky_relu(F.linear(x, sd["encoder_fc1.0.weight"], sd["encoder_fc1.0.bias"]), 0.02)
    return F.linear(x, sd["encoder_fc1.2.weight"], sd["encoder_fc1.2.bias"])


# --------------------------------------------------------------------------- A5
def rgb_forward(sd: SD, uv_audio: torch.Tensor, time_index: int) -> torch.Tensor:
    """As-written v2 MLP: rows [N, 2+64] (+ one frame index) -> [N,3], no output activation.
    Reference: tf_nerf.py:225-285 under the May flags (audio_net, audio_not_embed, use_time):
    net = fc_uv(E(uv)) + fc_audio(a) + fc_time(PE(t)); 8x relu(Linear); after layer 4
    h = cat([skip, h]) with skip = fc_uv_skip + fc_audio_skip + fc_time_skip; output_linear."""
    dt = uv_audio.dtype
    e = embed_uv(uv_audio[:, :2])
    a = uv_audio[:, 2:]
    t = time_pe(time_index, 20, dt, uv_audio.device)
    lin = lambda n, x: F.linear(x, sd[n + ".weight"], sd[n + ".bias"])
    h = lin("fc_uv", e) + lin("fc_audio", a) + lin("fc_time", t)
    for i in range(8):
        h = F.relu(lin(f"pts_linears.{i}", h))
        if i == 4:
            skip = lin("fc_uv_skip", e) + lin("fc_audio_skip", a) + lin("fc_time_skip", t)
            h = torch.cat([skip, h], dim=-1)
    return lin("output_linear", h)


# --------------------------------------------------------------------------- A6
def render_frame_as_shipped(sd: SD, window: torch.Tensor, index: int, height: int, width: int) -> torch.Tensor:
    """One lip frame exactly as the shipped driver computes it: the audio window is tiled
    to H*W rows and the encoder runs on every copy.  Reference: inference.py:144-159.
    window [16,29] -> [H,W,3]."""
    hw = height * width
    audio = window.unsqueeze(0).expand(hw, 16, 29).contiguous()
    coords = get_coords(width, height, window.dtype, window.device)
    feat = audio_encode(sd, audio)
    out = rgb_forward(sd, torch.cat([coords, feat], dim=-1), index)
    return out[:, :3].reshape(height, width, 3)


def render_clip(sd: SD, windows: torch.Tensor, indices, height: int, width: int) -> torch.Tensor:
    """Batched/factored CPU variant (encoder once per frame).  Same function of the inputs
    as `render_frame_as_shipped`; used as the fair CPU baseline and the large-case oracle.
    windows [F,16,29] -> [F,H,W,3]."""
    coords = get_coords(width, height, windows.dtype, windows.device)
    feats = audio_encode(sd, windows)
    frames = []
    for f in range(windows.shape[0]):
        row = torch.cat([coords, feats[f].unsqueeze(0).expand(coords.shape[0], -1)], dim=-1)
        frames.append(rgb_forward(sd, row, int(indices[f])).reshape(height, width, 3))
    return torch.stack(frames)


# --------------------------------------------------------------------------- A7
def _grid_sample_bilinear_zeros(img: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """Own restatement of bilinear grid sampling, zero padding, align_corners=False.
    img [B,C,H,W], grid [B,Ho,Wo,2] (x,y in [-1,1]) -> [B,C,Ho,Wo].  Semantics of
    F.grid_sample as called at tf_nerf.py:366-367."""
    B, C, H, W = img.shape
    x = ((grid[..., 0] + 1) * W - 1) / 2
    y = ((grid[..., 1] + 1) * H - 1) / 2
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - x) * (y1 - y)
    w_ne = (x - x0) * (y1 - y)
    w_sw = (x1 - x) * (y - y0)
    w_se = (x - x0) * (y - y0)
    flat = img.reshape(B, C, H * W)

    def tap(xi, yi, w):
        ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(B, 1, -1).expand(B, C, -1)
        val = torch.gather(flat, 2, idx).reshape(B, C, *xi.shape[1:])
        return val * (w * ok.to(w.dtype)).unsqueeze(1)

    return tap(x0, y0, w_nw) + tap(x1, y0, w_ne) + tap(x0, y1, w_sw) + tap(x1, y1, w_se)


def composite(lip: torch.Tensor, face_canon: torch.Tensor, rgb_gt: torch.Tensor, mask: torch.Tensor,
              x0: int, y0: int, coord: torch.Tensor, pad_mode: int = PAD_MODE_MAY,
              expand_lip_mask: bool = True, pad_div: int = 5,
              use_builtin_grid_sample: bool = True, blackaug=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Paste + head-pose-warp composite up to (not including) the U-Net.
    lip [B,h,w,3]; face_canon/rgb_gt/mask [B,FH,FW,3]; coord [B,FH,FW,2]
    -> (rgb_merged_new [B,FH,FW,3], rgb_merged_canonical [B,FH,FW,3]).

    Reference: tf_nerf.py:320-389 -- (1) zero-pad the lip into the face frame at
    (x0,y0) [pad_mode may] or (x0-1,y0-1) [default] (:339-350); (2) lerp with the soft lip
    mask (:352); (3) rectangular expanded mask rows [y0-p, y0+h+2p), cols [x0-p, x0+w+p),
    p = w // pad_div (:354-364); (4) bilinear zero-padded grid_sample of image and mask,
    mask binarised by !=0 (:366-369); (5) training only, `blackaug` = (n1, n2): the two N(0,1) fields [B,FH,FW] that
    `add_black_hole` (:306-318) draws with torch.randn (channel 0 of a randn of the image shape), when the coin
    `random.random() > 0.5` of :371 came up -- holes are punched where a draw is < 1e-6 inside the warped canonical face
    (grid_sample of `face_canon > 0` equal to exactly 1): hole pixels of the merged image show the observed frame and
    hole pixels of the observed frame show the merged image (:371-384); (6) blend with the observed frame (:386)."""
    B, h, w, _ = lip.shape
    FH, FW = face_canon.shape[1:3]
    ox, oy = (x0, y0) if pad_mode == PAD_MODE_MAY else (x0 - 1, y0 - 1)
    # F.pad as the reference calls it (:343-350): a NEGATIVE amount crops, so a lip box that leaves the face frame is pasted
    # with its outside part cut off (and a box entirely outside raises, as there)
    lip_pad = F.pad(lip.permute(0, 3, 1, 2), (ox, FW - ox - w, oy, FH - oy - h), mode="constant", value=0).permute(0, 2, 3, 1)
    merged_c = mask * lip_pad + (1 - mask) * face_canon
    if expand_lip_mask:
        p = w // pad_div
        m = torch.zeros_like(mask)
        # python slicing as in :362 -- a negative start (x0 < p) WRAPS to the far side, which usually leaves an empty slice:
        # the rectangle vanishes and the output is the observed frame everywhere.  Reproduced, not repaired.
        m[:, y0 - p:y0 + h + 2 * p, x0 - p:x0 + w + p, :] = 1
    else:
        m = mask.clone()
    gs = (lambda im: F.grid_sample(im, coord, mode="bilinear", padding_mode="zeros", align_corners=False)) \
        if use_builtin_grid_sample else (lambda im: _grid_sample_bilinear_zeros(im, coord))
    warped = gs(merged_c.permute(0, 3, 1, 2))
    mw = gs(m.permute(0, 3, 1, 2))
    mw = (mw != 0).to(lip.dtype)
    gt = rgb_gt.permute(0, 3, 1, 2)
    if blackaug is not None:
        face_obs = gs((face_canon > 0).to(lip.dtype).permute(0, 3, 1, 2))
        face_obs = (face_obs == 1).to(lip.dtype)                                    # :374
        keep = []
        for n in blackaug:
            noise = (n.unsqueeze(1) >= 0.000001).to(lip.dtype)                      # :309-310: 0 = hole, 1 = keep
            keep.append(noise * face_obs + torch.ones_like(noise) * (1 - face_obs))  # :315 (then `!= 0 -> 1`, a no-op on 0/1)
        before = warped
        warped = keep[0] * before + (1 - keep[0]) * gt                              # :382
        gt = keep[1] * gt + (1 - keep[1]) * before                                  # :383
    merged_new = mw * warped + (1 - mw) * gt
    return merged_new.permute(0, 2, 3, 1).contiguous(), merged_c


# --------------------------------------------------------------------------- T1/T2
def predict_lip_image(sd: SD, coords: torch.Tensor, window: torch.Tensor, index: int,
                      height: int, width: int, eps_u01: float) -> torch.Tensor:
    """4-tap local-ensemble forward of training (one frame) -> [HW,3].
    Reference: src/face_simple/training.py:158-251.  `eps_u01` is the single U(0,1) draw
    of :200 (the reference draws it with torch.rand; callers pass it in for determinism).
    Taps at clamp(coords + (vx*0.5/W + eps, vy*0.5/H + eps), 0, 1); areas |du*dv|+1e-9,
    swapped 0<->3, 1<->2; area-weighted sum."""
    dt = coords.dtype
    feat = audio_encode(sd, window.unsqueeze(0))[0]
    rx, ry = 0.5 / width, 0.5 / height
    eps = torch.tensor(ry, dtype=torch.float32) * torch.tensor(eps_u01, dtype=torch.float32) / 2.0
    eps = eps.to(device=coords.device, dtype=dt)
    preds, areas = [], []
    for vx in (-1, 1):
        for vy in (-1, 1):
            c = coords.clone()
            c[:, 0] += vx * rx + eps
            c[:, 1] += vy * ry + eps
            c.clamp_(0, 1)
            row = torch.cat([c, feat.unsqueeze(0).expand(c.shape[0], -1)], dim=-1)
            preds.append(rgb_forward(sd, row, index))
            areas.append(torch.abs((c[:, 0] - coords[:, 0]) * (c[:, 1] - coords[:, 1])) + 1e-9)
    tot = torch.stack(areas).sum(dim=0)
    areas[0], areas[3] = areas[3], areas[0]
    areas[1], areas[2] = areas[2], areas[1]
    ret = 0
    for p, a in zip(preds, areas):
        ret = ret + p * (a / tot).unsqueeze(-1)
    return ret[:, :3]


def mse_loss(pred: torch.Tensor, target: torch.Tensor, weight: float = 1.0) -> torch.Tensor:
    """Photometric loss mean((pred-target)^2)*w.  Reference: training.py:605-619."""
    return ((pred - target) ** 2).mean() * weight


# --------------------------------------------------------------------------- §8f-1 U-Net
def unet_forward(sd: SD, x_nhwc: torch.Tensor, prefix: str = "post_fusion_unet.", eps: float = 1e-5, training: bool = False,
                 new_stats: Optional[dict] = None, momentum: float = 0.1) -> torch.Tensor:
    """Eval-mode `SimpleUnetLight` on NHWC input [B,H,W,3] -> [B,H,W,3].
    Reference: src/face_simple/models/SimpleUnetLight.py:16-111 (DoubleConv = (conv3x3 no bias -> BatchNorm
    -> ReLU) x2; Down = MaxPool2d(2) + DoubleConv; Up = bilinear x2 (align_corners=True), pad to the
    skip's size, cat([skip, up]) + DoubleConv(in, out, in//2); outc = conv1x1), called from
    tf_nerf.py:387 on `rgb_merged_new`.
    training=True: the network as the reference runs it until `it > 100000` (train.py:188-197: train mode): BatchNorm2d
    normalises with the statistics of the batch (biased variance) and, when `new_stats` is a dict, the running statistics
    it would hold afterwards are written there under the state-dict names (momentum 0.1, unbiased variance,
    num_batches_tracked + 1) -- nn.BatchNorm2d's documented update."""
    def cbr(x, name):
        head, idx = name.rsplit(".", 1)
        bn = f"{prefix}{head}.{int(idx) + 1}"
        y = F.conv2d(x, sd[f"{prefix}{name}.weight"], None, padding=1)
        if training:
            mean = y.mean(dim=(0, 2, 3))
            var = y.var(dim=(0, 2, 3), unbiased=False)
            if new_stats is not None:      # a dict that already holds statistics (an earlier call's) is continued from
                n = y.numel() // y.shape[1]
                old = lambda k: new_stats[k] if new_stats.get(k) is not None else sd.get(k)
                nbt = old(bn + ".num_batches_tracked")
                new_stats[bn + ".running_mean"] = ((1 - momentum) * old(bn + ".running_mean") + momentum * mean).detach()
                new_stats[bn + ".running_var"] = ((1 - momentum) * old(bn + ".running_var") + momentum * var * n / max(n - 1, 1)).detach()
                new_stats[bn + ".num_batches_tracked"] = nbt + 1 if nbt is not None else None
            y = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps) * sd[bn + ".weight"].view(1, -1, 1, 1) \
                + sd[bn + ".bias"].view(1, -1, 1, 1)
            return F.relu(y)
        scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + eps)
        y = (y - sd[bn + ".running_mean"].view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1) + sd[bn + ".bias"].view(1, -1, 1, 1)
        return F.relu(y)

    def double(x, stem):
        return cbr(cbr(x, stem + ".0"), stem + ".3")

    def up(xlow, skip, stem):
        xu = F.interpolate(xlow, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = skip.shape[2] - xu.shape[2], skip.shape[3] - xu.shape[3]
        xu = F.pad(xu, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return double(torch.cat([skip, xu], dim=1), stem)

    x = x_nhwc.permute(0, 3, 1, 2)
    x1 = double(x, "inc.double_conv")
    x2 = double(F.max_pool2d(x1, 2), "down1.maxpool_conv.1.double_conv")
    x3 = double(F.max_pool2d(x2, 2), "down2.maxpool_conv.1.double_conv")
    y = up(x3, x2, "up1.conv.double_conv")
    y = up(y, x1, "up2.conv.double_conv")
    y = F.conv2d(y, sd[prefix + "outc.conv.weight"], sd[prefix + "outc.conv.bias"])
    return y.permute(0, 2, 3, 1).contiguous()


# --------------------------------------------------------------------------- §8f-4: perceptual term (LPIPS, AlexNet)
# PARITY UNPINNED: `lpips` (requirement.txt:11, lpips==0.1.4) and torchvision are third-party packages that are neither in the
# reference repository nor installed here, so this restatement of their PUBLISHED forward pass (lpips/lpips.py LPIPS.forward,
# lpips/pretrained_networks.py alexnet, lpips/__init__.py normalize_tensor / spatial_average) cannot be checked against the
# package itself; it is anchored on the reference's call site (training.py:76, 655-674) and exercised with seeded weights.
LPIPS_ALEX = (("net.slice1.0", 4, 2, False), ("net.slice2.3", 1, 2, True), ("net.slice3.6", 1, 1, True), ("net.slice4.8", 1, 1, False),
              ("net.slice5.10", 1, 1, False))       # (conv, stride, padding, MaxPool2d(3, 2) in front)


def lpips_alex(sd: SD, in0: torch.Tensor, in1: torch.Tensor, eps: float = 1e-10) -> torch.Tensor:
    """lpips.LPIPS(net='alex', version='0.1', lpips=True, spatial=False)(in0, in1): NCHW images in [-1,1] -> [N,1,1,1]."""
    def feats(x):
        x = (x - sd["scaling_layer.shift"].to(x)) / sd["scaling_layer.scale"].to(x)
        outs = []
        for name, stride, pad, pool in LPIPS_ALEX:
            if pool:
                x = F.max_pool2d(x, kernel_size=3, stride=2)
            x = F.relu(F.conv2d(x, sd[name + ".weight"].to(x), sd[name + ".bias"].to(x), stride=stride, padding=pad))
            outs.append(x)
        return outs
    val = 0
    for kk, (f0, f1) in enumerate(zip(feats(in0), feats(in1))):
        n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + eps)
        n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True)) + eps)
        lin = F.conv2d((n0 - n1) ** 2, sd[f"lin{kk}.model.1.weight"].to(f0))
        val = val + lin.mean([2, 3], keepdim=True)
    return val


def perceptual_loss(sd: SD, prediction_nhwc: torch.Tensor, target_nhwc: torch.Tensor, weights: float = 1.0) -> torch.Tensor:
    """Trainer.add_perceptual_loss (training.py:655-674), mask = ones: (x - 0.5) * 2, LPIPS, mean over the batch, * weights."""
    recon_x = (prediction_nhwc.permute(0, 3, 1, 2) - 0.5) * 2
    x = (target_nhwc.permute(0, 3, 1, 2) - 0.5) * 2
    return lpips_alex(sd, recon_x, x).mean() * weights


# --------------------------------------------------------------------------- metrics
# --------------------------------------------------------------------------- §8f-3: pose -> warp grid
POSE_OBS2CAN = 0   # Tc . inv(T): the grid written to coords/*.npy (face_tracker.py:583-584, utils.py:54-58)
POSE_CAN2OBS = 1   # T . inv(Tc): training.py:263-268, utils.py:60-71
POSE_CAN2OBS_INV = 2   # inv(T . inv(Tc)): training.py:270-275 (the rel_pose of the depth photo loss)


def euler2rot(euler: torch.Tensor) -> torch.Tensor:
    """[B,3] (theta, phi, psi) -> Rx(theta) Ry(phi) Rz(psi) with the reference's sign layout (utils.py:8-34)."""
    th, ph, ps = euler[:, 0], euler[:, 1], euler[:, 2]
    o, z = torch.ones_like(th), torch.zeros_like(th)
    rx = torch.stack([o, z, z, z, th.cos(), -th.sin(), z, th.sin(), th.cos()], -1).reshape(-1, 3, 3)
    ry = torch.stack([ph.cos(), z, ph.sin(), z, o, z, -ph.sin(), z, ph.cos()], -1).reshape(-1, 3, 3)
    rz = torch.stack([ps.cos(), ps.sin(), z, -ps.sin(), ps.cos(), z, z, z, o], -1).reshape(-1, 3, 3)
    return rx @ ry @ rz


def prepare_transform_matrix(euler: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """[B,3],[B,3] -> [B,4,4] = [R(e0,-e1,-e2) | (t0,-t1,-t2)] (utils.py:36-52)."""
    sgn = torch.tensor([1.0, -1.0, -1.0], dtype=euler.dtype)
    T = torch.zeros(euler.shape[0], 4, 4, dtype=euler.dtype)
    T[:, :3, :3] = euler2rot(euler * sgn)
    T[:, :3, 3] = trans * sgn
    T[:, 3, 3] = 1
    return T


def rel_pose(canonical_euler, canonical_trans, euler, trans, mode: int) -> torch.Tensor:
    Tc = prepare_transform_matrix(canonical_euler.reshape(1, 3), canonical_trans.reshape(1, 3)).expand(euler.shape[0], 4, 4)
    T = prepare_transform_matrix(euler, trans)
    if mode == POSE_OBS2CAN:
        return Tc @ torch.inverse(T)
    out = T @ torch.inverse(Tc)
    return out if mode == POSE_CAN2OBS else torch.inverse(out)


def warp_grid(depth: torch.Tensor, T: torch.Tensor, focal: float, clamp: bool = False, eps: float = 1e-7):
    """depth [B or 1,H,W], T [B,4,4] -> (grid [B,H,W,2] in grid_sample units, z [B,H,W]).
    BackprojectDepth (utils.py:115-143): X = depth * pinv(K)[:3,:3] (x, y, 1), K = [[f,0,W/2],[0,f,H/2],[0,0,1]]
    (training.py:296-304); Project3D (utils.py:145-169): p = (K T)[:3] (X,1); pix = p.xy / (p.z + eps);
    pix.x /= W-1; pix.y /= H-1; grid = (pix - 0.5) * 2.  `clamp` = the [-1,1] clamp of face_tracker.py:606."""
    B = T.shape[0]
    H, W = depth.shape[-2:]
    dt = depth.dtype
    K = torch.tensor([[focal, 0, W / 2, 0], [0, focal, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=dt)
    inv_K = torch.linalg.pinv(K)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=dt)], 0)           # [3,HW]
    cam = depth.reshape(-1, 1, H * W).expand(B, 1, H * W) * (inv_K[:3, :3] @ pix)                  # [B,3,HW]
    cam = torch.cat([cam, torch.ones(B, 1, H * W, dtype=dt)], 1)
    P = (K @ T)[:, :3, :]
    p = P @ cam
    g = (p[:, :2] / (p[:, 2:3] + eps)).reshape(B, 2, H, W).permute(0, 2, 3, 1).clone()
    g[..., 0] /= W - 1
    g[..., 1] /= H - 1
    g = (g - 0.5) * 2
    if clamp:
        g = g.clamp(-1, 1)
    return g, p[:, 2].reshape(B, H, W)


def inverse_warping(depth: torch.Tensor, T: torch.Tensor, src_nhwc: torch.Tensor, focal: float):
    """training.py:296-314: grid from (depth, rel_pose), then border-padded bilinear sampling of src."""
    g, z = warp_grid(depth[None], T, focal)
    img = F.grid_sample(src_nhwc.permute(0, 3, 1, 2), g, mode="bilinear", padding_mode="border", align_corners=False)
    return img, z[:, None]


def depth_photo_loss(depth: torch.Tensor, T: torch.Tensor, src_nhwc: torch.Tensor, target_nhwc: torch.Tensor, mask, focal: float,
                     weights: float = 1.0) -> torch.Tensor:
    """Canonical-depth photometric loss (training.py:462-477): the observed frame warped into the canonical view by the depth
    map (Trainer.inverse_warping, :296-314) against the canonical face, masked mean of the squared error
    (add_loss_canonical_depth_photo, :621-634).  Differentiable w.r.t. `depth` (model.canonical_depth_head)."""
    pred, _ = inverse_warping(depth, T, src_nhwc, focal)                      # NCHW
    pred = pred.permute(0, 2, 3, 1)
    if mask is not None:
        return ((pred - target_nhwc) ** 2 * mask).sum() / (mask.sum() + 1e-6) * weights
    return ((pred - target_nhwc) ** 2).mean() * weights


# --------------------------------------------------------------------------- T3: lip-sync expert loss
def syncnet_encoder(sd: SD, x: torch.Tensor, prefix: str, blocks, eps: float = 1e-5, margins: Optional[list] = None) -> torch.Tensor:
    """One encoder of SyncNet_color (syncnet.py:11-54) in eval mode: per block conv -> BatchNorm(running stats) ->
    (+ x if residual) -> ReLU (conv.py:5-19).  `blocks` = speech2lip_amd.weights.SYNCNET_FACE / SYNCNET_AUDIO.
    margins (a list, test aid): per block the [B] tensor min |pre-ReLU value| / rms of that sample's pre-ReLU map -- how far the
    sample's closest ReLU decision is from a tie (another fp32 evaluation order may resolve a tie the other way, which changes
    the GRADIENT inside that unit's receptive field by O(1), not by rounding)."""
    for i, (cin, cout, k, stride, pad, res) in enumerate(blocks):
        p = f"{prefix}.{i}.conv_block."
        y = F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride=stride, padding=pad)
        y = F.batch_norm(y, sd[p + "1.running_mean"], sd[p + "1.running_var"], sd[p + "1.weight"], sd[p + "1.bias"],
                         training=False, eps=eps)
        z = y + x if res else y
        if margins is not None:
            zd = z.detach().reshape(z.shape[0], -1)
            margins.append(zd.abs().min(dim=1).values / zd.pow(2).mean(dim=1).sqrt().clamp_min(1e-30))
        x = F.relu(z)
    return x


def syncnet_forward(sd: SD, audio_sequences: torch.Tensor, face_sequences: torch.Tensor, blocks_face, blocks_audio,
                    face_margins: Optional[list] = None):
    """SyncNet_color.forward (syncnet.py:57-67): ([B,1,80,16], [B,15,48,96]) -> L2-normalised ([B,512], [B,512])."""
    f = syncnet_encoder(sd, face_sequences, "face_encoder", blocks_face, margins=face_margins)
    a = syncnet_encoder(sd, audio_sequences, "audio_encoder", blocks_audio)
    a, f = a.reshape(a.shape[0], -1), f.reshape(f.shape[0], -1)
    return F.normalize(a, p=2, dim=1), F.normalize(f, p=2, dim=1)


def sync_window(g_rgb: torch.Tensor, syncnet_T: int = 5) -> torch.Tensor:
    """[B,3,T,H,W] RGB -> [B,3T,H-H//2,W]: BGR, lower half rows, frames stacked on channels (training.py:588-590)."""
    g = g_rgb[:, [2, 1, 0]]
    g = g[:, :, :, g.size(3) // 2:]
    return torch.cat([g[:, :, i] for i in range(syncnet_T)], dim=1)


def cosine_loss(a: torch.Tensor, v: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """training.py:576-579."""
    d = F.cosine_similarity(a, v)
    return F.binary_cross_entropy(d.unsqueeze(1), y)


def sync_contrastive_loss(sd: SD, mel, g_rgb_pos, g_rgb_neg, blocks_face, blocks_audio, syncnet_T: int = 5,
                          pos_margins: Optional[list] = None):
    """Trainer.get_sync_contrastive_loss (training.py:581-603): BCE(cos, 1) on the generated window + BCE(cos, 0) on the
    negative window.  pos_margins: see syncnet_encoder (the face encoder on the generated window, the pass gradients flow through)."""
    B = mel.shape[0]
    a, v = syncnet_forward(sd, mel, sync_window(g_rgb_pos, syncnet_T), blocks_face, blocks_audio, face_margins=pos_margins)
    pos = cosine_loss(a, v, torch.ones(B, 1, dtype=mel.dtype, device=mel.device))
    a, v = syncnet_forward(sd, mel, sync_window(g_rgb_neg, syncnet_T), blocks_face, blocks_audio)
    neg = cosine_loss(a, v, torch.zeros(B, 1, dtype=mel.dtype, device=mel.device))
    return pos + neg


# --------------------------------------------------------------------------- §8f-4 (light half): sync loss into the MLP
def crop_resize(img_nhwc: torch.Tensor, bbox, size=(96, 96)) -> torch.Tensor:
    """training.py:541-544: `rgb_merged[:, y:y2, x:x2, :]`, then `transforms.Resize([96, 96])` on the NCHW tensor.
    torchvision 0.9.0 (requirement.txt:34) resizes TENSORS with functional_tensor.resize ->
    torch.nn.functional.interpolate(img, size=[h, w], mode='bilinear', align_corners=False) (no antialiasing in 0.9).
    torchvision is not installed in this image: this function is the restatement of that one call.
    bbox = (x, y, x2, y2) of data['canonical_face_bbox'][0].  [B,FH,FW,3] -> [B,96,96,3]."""
    x, y, x2, y2 = (int(v) for v in bbox[:4])
    crop = img_nhwc[:, y:y2, x:x2, :].permute(0, 3, 1, 2)
    return F.interpolate(crop, size=list(size), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)


def sync_chain_window(sd: SD, unet_sd: SD, coords, audio_window, index: int, total_frame: int, eps_u01, face_canon, rgb_gt,
                      mask, x0: int, y0: int, coord_window, bbox, height: int, width: int, pad_mode: int = PAD_MODE_MAY,
                      pad_div: int = 5, unet_training: bool = False, new_stats: Optional[dict] = None) -> torch.Tensor:
    """The generated 5-frame window of the sync loss for ONE sample (training.py:491-548): for t in 0..T-1 the 4-tap
    ensemble render of audio_window[t] at frame index min(index + t, total_frame - 1) (:515-518), pasted and warped with
    coord_window[t] against the MAIN frame's observed image (:527-536), through the frozen U-Net (the first return
    value of post_fusion2_onlylip is rgb_recon, tf_nerf.py:387-389), cropped to the canonical face box and resized to
    96x96 (:541-544).  BatchNorm mode of the frozen net: train.py:188-197 calls post_fusion_unet.eval() ONCE when the
    sync loss becomes active, but Trainer.train_step (training.py:150) calls self.model.train() on every step, which puts
    the sub-module back into train mode -- so a reference run that goes through train_step normalises each one-frame call
    with that frame's batch statistics (and keeps updating the running statistics) although the parameters are frozen:
    `unet_training=True` (`new_stats` collects the running statistics, call after call).  `unet_training=False` is
    train_stage1 called with the sub-module left in eval mode (golden G11).
    audio_window [T,16,29]; eps_u01: one U(0,1) draw per frame (each predict_lip_image call draws its own, :200);
    face_canon/rgb_gt/mask [1,FH,FW,3]; coord_window [T,FH,FW,2] -> rgb_window [1,3,T,96,96] (:547-548)."""
    frames = []
    for t in range(audio_window.shape[0]):
        idx = index + t if index + t < total_frame else total_frame - 1
        lip = predict_lip_image(sd, coords, audio_window[t], idx, height, width, eps_u01[t]).reshape(1, height, width, 3)
        new, _ = composite(lip, face_canon, rgb_gt, mask, x0, y0, coord_window[t:t + 1], pad_mode=pad_mode, pad_div=pad_div)
        recon = unet_forward(unet_sd, new, training=unet_training, new_stats=new_stats)
        frames.append(crop_resize(recon, bbox))
    win = torch.stack(frames, 0)                       # T,B,H,W,C
    return win.permute(1, 4, 0, 2, 3)                  # B,C,T,H,W


def stage_one_losses(sd: SD, unet_sd: SD, sync_sd: SD, blocks_face, blocks_audio, data: dict, eps_u01, hole_noise, height: int,
                     width: int, lambda_rgb: float = 1.0, w_post_fusion: float = 1.0, w_syncloss: float = 0.01,
                     pad_mode: int = PAD_MODE_MAY, pad_div: int = 5, unet_training: bool = False, with_sync: bool = True,
                     new_stats: Optional[dict] = None) -> dict:
    """The loss of ONE reference optimisation step after `it > 100000` (Trainer.train_stage1, training.py:347-574) under the
    May flags with the LPIPS and canonical-depth terms switched off:
        loss = lambda_rgb * MSE(predict_lip_image, rgb)                                              (:414-418)
             + lambda_rgb * w_post_fusion * MSE(post_fusion2_onlylip(lip, blackaug=True)[0], rgb_face_ori)   (:436-459)
             + w_syncloss * get_sync_contrastive_loss(mel, rgb_window, rgb_window_neg)               (:491-557)
    `data` holds the reference's batch dict entries (batch 1): audio [1,16,29], rgb [1,h,w,3], index, total_frame,
    rgb_face_zero, rgb_face_ori, mask_lip_canonical [1,FH,FW,3], lip_lefttop_x/y, coord [1,FH,FW,2], audio_window
    [1,T,16,29], coord_window [1,T,FH,FW,2], canonical_face_bbox [[x,y,x2,y2,score]], mel [1,1,80,16], rgb_window_neg
    [1,3,T,96,96].  eps_u01: the 1 + T draws of torch.rand in the order the step makes them (main frame first);
    hole_noise: None when the coin of tf_nerf.py:371 came up tails, else the two randn fields [1,FH,FW].
    unet_training / with_sync=False: the step BEFORE `it > 100000` -- the post-fusion net in train mode (BatchNorm batch
    statistics, its parameters trained too) and no sync term (:491 is false).
    unet_training / with_sync=True: the step after `it > 100000` as Trainer.train_step runs it (self.model.train() at
    training.py:150 undoes train.py's post_fusion_unet.eval()): every one-frame U-Net call -- the main frame first, then
    the T window frames -- uses its own batch statistics; `new_stats` receives the running statistics after all 1 + T calls.
    Differentiable w.r.t. `sd` and `unet_sd` (build them with requires_grad tensors)."""
    coords = get_coords(width, height, device=data["audio"].device)
    idx = int(data["index"])
    x0, y0 = int(data["lip_lefttop_x"]), int(data["lip_lefttop_y"])
    pred = predict_lip_image(sd, coords, data["audio"][0], idx, height, width, eps_u01[0])
    loss_rgb = mse_loss(pred, data["rgb"].reshape(-1, 3), lambda_rgb)
    lip = pred.reshape(1, height, width, 3)
    new, _ = composite(lip, data["rgb_face_zero"], data["rgb_face_ori"], data["mask_lip_canonical"], x0, y0, data["coord"],
                       pad_mode=pad_mode, pad_div=pad_div, blackaug=hole_noise)
    recon = unet_forward(unet_sd, new, training=unet_training, new_stats=new_stats)
    loss_face = mse_loss(recon, data["rgb_face_ori"], lambda_rgb * w_post_fusion)
    if not with_sync:
        return {"loss": loss_rgb + loss_face, "loss_rgb": loss_rgb, "loss_face": loss_face, "pred": pred, "rgb_face_recon": recon}
    window = sync_chain_window(sd, unet_sd, coords, data["audio_window"][0], idx, int(data["total_frame"]), eps_u01[1:],
                               data["rgb_face_zero"], data["rgb_face_ori"], data["mask_lip_canonical"], x0, y0,
                               data["coord_window"][0], data["canonical_face_bbox"][0], height, width, pad_mode, pad_div,
                               unet_training=unet_training, new_stats=new_stats)
    loss_sync = sync_contrastive_loss(sync_sd, data["mel"], window, data["rgb_window_neg"], blocks_face, blocks_audio,
                                      window.shape[2]) * w_syncloss
    return {"loss": loss_rgb + loss_face + loss_sync, "loss_rgb": loss_rgb, "loss_face": loss_face, "loss_sync": loss_sync,
            "pred": pred, "rgb_face_recon": recon, "rgb_window": window}


def psnr(a: torch.Tensor, b: torch.Tensor, peak: float = 1.0) -> float:
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)


def rmse(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(((a.double() - b.double()) ** 2).mean().sqrt())
