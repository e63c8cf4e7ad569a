"""Deterministic synthetic weights ("G0" of SURVEY.md §8c) for the lip-render hot path.

There is no network on the build or GPU boxes, so no real Speech2Lip checkpoint exists
here.  Parity tests, goldens and the bench all use weights produced by this generator:
a counter-based splitmix64 stream (pure integer arithmetic -> bit-identical on every
host, independent of torch/numpy RNG versions) mapped to U(-b, b).

The tensor names and shapes are the state-dict keys the reference model creates under the
May flag set (`/root/reference/src/face_simple/models/tf_nerf.py:91-109` audio encoder,
`:131-135` dead `coord_linears`, `:144` output layer, `:146-172` v2 MLP).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

# (name, shape) in the order the packer consumes them.  Linear weights are [out, in],
# Conv1d weights are [out, in, k] exactly as torch stores them.
HOT_PATH_TENSORS = [
    ("encoder_conv.0.weight", (32, 29, 3)), ("encoder_conv.0.bias", (32,)),
    ("encoder_conv.2.weight", (32, 32, 3)), ("encoder_conv.2.bias", (32,)),
    ("encoder_conv.4.weight", (64, 32, 3)), ("encoder_conv.4.bias", (64,)),
    ("encoder_conv.6.weight", (64, 64, 3)), ("encoder_conv.6.bias", (64,)),
    ("encoder_fc1.0.weight", (64, 64)), ("encoder_fc1.0.bias", (64,)),
    ("encoder_fc1.2.weight", (64, 64)), ("encoder_fc1.2.bias", (64,)),
    ("fc_uv.weight", (256, 42)), ("fc_uv.bias", (256,)),
    ("fc_audio.weight", (256, 64)), ("fc_audio.bias", (256,)),
    ("fc_time.weight", (256, 20)), ("fc_time.bias", (256,)),
    ("fc_uv_skip.weight", (256, 42)), ("fc_uv_skip.bias", (256,)),
    ("fc_audio_skip.weight", (256, 64)), ("fc_audio_skip.bias", (256,)),
    ("fc_time_skip.weight", (256, 20)), ("fc_time_skip.bias", (256,)),
    ("pts_linears.0.weight", (256, 256)), ("pts_linears.0.bias", (256,)),
    ("pts_linears.1.weight", (256, 256)), ("pts_linears.1.bias", (256,)),
    ("pts_linears.2.weight", (256, 256)), ("pts_linears.2.bias", (256,)),
    ("pts_linears.3.weight", (256, 256)), ("pts_linears.3.bias", (256,)),
    ("pts_linears.4.weight", (256, 256)), ("pts_linears.4.bias", (256,)),
    ("pts_linears.5.weight", (256, 512)), ("pts_linears.5.bias", (256,)),
    ("pts_linears.6.weight", (256, 256)), ("pts_linears.6.bias", (256,)),
    ("pts_linears.7.weight", (256, 256)), ("pts_linears.7.bias", (256,)),
    ("output_linear.weight", (3, 256)), ("output_linear.bias", (3,)),
]

# Present in reference checkpoints, never read by any forward (tf_nerf.py:131-135).
DEAD_TENSORS = [
    ("coord_linears.0.weight", (256, 2)), ("coord_linears.0.bias", (256,)),
    ("coord_linears.1.weight", (256, 256)), ("coord_linears.1.bias", (256,)),
    ("coord_linears.2.weight", (256, 256)), ("coord_linears.2.bias", (256,)),
    ("coord_linears.3.weight", (256, 256)), ("coord_linears.3.bias", (256,)),
    ("coord_linears.4.weight", (64, 256)), ("coord_linears.4.bias", (64,)),
]

HOT_PATH_PARAM_COUNT = sum(int(np.prod(s)) for _, s in HOT_PATH_TENSORS)  # 691,491

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over a uint64 counter array."""
    with np.errstate(over="ignore"):
        z = (counter + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform01(n: int, stream: int) -> np.ndarray:
    """n float64 values in [0,1) from stream `stream` (53 mantissa bits each)."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([stream], dtype=np.uint64))[0]
        ctr = base + np.arange(n, dtype=np.uint64)
    return (splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _stream_id(name: str, seed: int) -> int:
    return ((zlib.crc32(name.encode()) & 0xFFFFFFFF) << 20) ^ (seed & 0xFFFFF)


def make_state_dict(seed: int = 0, gain: str = "he", include_dead: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Seeded fp32 weights for every hot-path tensor.

    gain="he":     weights U(+-sqrt(6/fan_in)) -- keeps hidden activations O(1) through the 8
                   ReLU layers; `output_linear.weight` is scaled by a further 1/8 so the RGB
                   output has RMS ~0.5 like a trained model, which makes the absolute 1e-4
                   RMSE / 50 dB PSNR bars meaningful (with "torch" gain the output RMS is
                   0.03 and any kernel passes an absolute bar).
    gain="torch":  weights U(+-1/sqrt(fan_in)), torch.nn.Linear's default scale.
    Biases are U(+-1/sqrt(fan_in)) in both modes.
    """
    if gain not in ("he", "torch"):
        raise ValueError(f"unknown gain {gain!r}")
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    fan_in = {}
    specs = HOT_PATH_TENSORS + (DEAD_TENSORS if include_dead else [])
    for name, shape in specs:
        if name.endswith(".weight"):
            fan_in[name[: -len(".weight")]] = int(np.prod(shape[1:]))
    for name, shape in specs:
        layer = name.rsplit(".", 1)[0]
        fi = fan_in[layer]
        if name.endswith(".weight"):
            bound = np.sqrt(6.0 / fi) if gain == "he" else 1.0 / np.sqrt(fi)
            if gain == "he" and layer == "output_linear":
                bound /= 8.0
        else:
            bound = 1.0 / np.sqrt(fi)
        u = uniform01(int(np.prod(shape)), _stream_id(name, seed))
        out[name] = ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)
    return out


def synthetic_audio(n_frames: int, seed: int = 1) -> np.ndarray:
    """Surrogate `audio.npy`: float64 [N,16,29] log-softmax of seeded normals, with the
    clip-end zero padding of the DeepSpeech windowing
    (`/root/reference/preprocess/deepspeech_features/deepspeech_features.py:65-75`:
    8 zero feature rows each side, 16-row windows at stride 2).  SURVEY.md §8d.
    """
    n_feat = 2 * n_frames
    u1 = uniform01(n_feat * 29, _stream_id("audio.u1", seed))
    u2 = uniform01(n_feat * 29, _stream_id("audio.u2", seed))
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)  # Box-Muller
    z = z.reshape(n_feat, 29)
    logits = z - z.max(axis=1, keepdims=True)
    logp = logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))
    padded = np.concatenate([np.zeros((8, 29)), logp, np.zeros((8, 29))], axis=0)
    idx = (2 * np.arange(n_frames))[:, None] + np.arange(16)[None, :]
    return padded[idx]  # [N,16,29] float64


# ---- post-fusion U-Net (SimpleUnetLight) ---------------------------------------
# (prefix, cin, cout) of the ten 3x3 convolutions, in execution order; each is followed by a BatchNorm.
UNET_CONVS = [
    ("inc.double_conv.0", 3, 64), ("inc.double_conv.3", 64, 64),
    ("down1.maxpool_conv.1.double_conv.0", 64, 128), ("down1.maxpool_conv.1.double_conv.3", 128, 128),
    ("down2.maxpool_conv.1.double_conv.0", 128, 128), ("down2.maxpool_conv.1.double_conv.3", 128, 128),
    ("up1.conv.double_conv.0", 256, 128), ("up1.conv.double_conv.3", 128, 64),
    ("up2.conv.double_conv.0", 128, 64), ("up2.conv.double_conv.3", 64, 64),
]


def _bn_name(conv_name: str) -> str:
    head, idx = conv_name.rsplit(".", 1)
    return f"{head}.{int(idx) + 1}"


def make_unet_state_dict(seed: int = 0, prefix: str = "post_fusion_unet.") -> "OrderedDict[str, np.ndarray]":
    """Seeded weights for `SimpleUnetLight` (reference: src/face_simple/models/SimpleUnetLight.py:82-111),
    state-dict keys as in a reference checkpoint.  He-uniform convolutions, BatchNorm affine and
    running statistics drawn away from the identity so that the eval-mode fold is exercised."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def u(name, n, lo, hi):
        return (lo + (hi - lo) * uniform01(n, _stream_id("unet." + name, seed))).astype(np.float32)

    for name, cin, cout in UNET_CONVS:
        b = np.sqrt(6.0 / (cin * 9))
        out[f"{prefix}{name}.weight"] = u(name + ".w", cout * cin * 9, -b, b).reshape(cout, cin, 3, 3)
        bn = _bn_name(name)
        out[f"{prefix}{bn}.weight"] = u(bn + ".g", cout, 0.5, 1.5)
        out[f"{prefix}{bn}.bias"] = u(bn + ".b", cout, -0.1, 0.1)
        out[f"{prefix}{bn}.running_mean"] = u(bn + ".m", cout, -0.1, 0.1)
        out[f"{prefix}{bn}.running_var"] = u(bn + ".v", cout, 0.5, 1.5)
        out[f"{prefix}{bn}.num_batches_tracked"] = np.array(100, dtype=np.int64)
    out[f"{prefix}outc.conv.weight"] = u("outc.w", 3 * 64, -0.15, 0.15).reshape(3, 64, 1, 1)
    out[f"{prefix}outc.conv.bias"] = u("outc.b", 3, -0.1, 0.1)
    return out


# ---- SyncNet_color (the lip-sync expert of the T3 loss) ---------------------------
# (cin, cout, (kh, kw), (sy, sx), (py, px), residual) per block, as src/face_simple/models/syncnet.py:11-54.
SYNCNET_FACE = [
    (15, 32, (7, 7), (1, 1), (3, 3), False), (32, 64, (5, 5), (1, 2), (1, 1), False),
    (64, 64, (3, 3), (1, 1), (1, 1), True), (64, 64, (3, 3), (1, 1), (1, 1), True),
    (64, 128, (3, 3), (2, 2), (1, 1), False), (128, 128, (3, 3), (1, 1), (1, 1), True),
    (128, 128, (3, 3), (1, 1), (1, 1), True), (128, 128, (3, 3), (1, 1), (1, 1), True),
    (128, 256, (3, 3), (2, 2), (1, 1), False), (256, 256, (3, 3), (1, 1), (1, 1), True),
    (256, 256, (3, 3), (1, 1), (1, 1), True), (256, 512, (3, 3), (2, 2), (1, 1), False),
    (512, 512, (3, 3), (1, 1), (1, 1), True), (512, 512, (3, 3), (1, 1), (1, 1), True),
    (512, 512, (3, 3), (2, 2), (1, 1), False), (512, 512, (3, 3), (1, 1), (0, 0), False),
    (512, 512, (1, 1), (1, 1), (0, 0), False),
]
SYNCNET_AUDIO = [
    (1, 32, (3, 3), (1, 1), (1, 1), False), (32, 32, (3, 3), (1, 1), (1, 1), True), (32, 32, (3, 3), (1, 1), (1, 1), True),
    (32, 64, (3, 3), (3, 1), (1, 1), False), (64, 64, (3, 3), (1, 1), (1, 1), True), (64, 64, (3, 3), (1, 1), (1, 1), True),
    (64, 128, (3, 3), (3, 3), (1, 1), False), (128, 128, (3, 3), (1, 1), (1, 1), True), (128, 128, (3, 3), (1, 1), (1, 1), True),
    (128, 256, (3, 3), (3, 2), (1, 1), False), (256, 256, (3, 3), (1, 1), (1, 1), True), (256, 256, (3, 3), (1, 1), (1, 1), True),
    (256, 512, (3, 3), (1, 1), (0, 0), False), (512, 512, (1, 1), (1, 1), (0, 0), False),
]
SYNCNET_BLOCKS = [("face_encoder", i, s) for i, s in enumerate(SYNCNET_FACE)] + \
                 [("audio_encoder", i, s) for i, s in enumerate(SYNCNET_AUDIO)]
SYNCNET_TENSORS = ("conv_block.0.weight", "conv_block.0.bias", "conv_block.1.weight", "conv_block.1.bias",
                   "conv_block.1.running_mean", "conv_block.1.running_var")


def make_syncnet_state_dict(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded SyncNet_color weights with the reference's state-dict keys.  The real `lipsync_expert.pth` is not in the
    reference repository (training.py:88), so parity of this net is structural: same generator on both sides.
    He-uniform convolutions (halved on residual blocks so the trunk does not explode), BatchNorm affine and running
    statistics away from the identity so that the eval-mode fold is exercised."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def u(name, n, lo, hi):
        return (lo + (hi - lo) * uniform01(n, _stream_id("syncnet." + name, seed))).astype(np.float32)

    for enc, i, (cin, cout, (kh, kw), _, _, res) in SYNCNET_BLOCKS:
        p = f"{enc}.{i}."
        b = np.sqrt(6.0 / (cin * kh * kw)) * (0.5 if res else 1.0)
        out[p + "conv_block.0.weight"] = u(p + "w", cout * cin * kh * kw, -b, b).reshape(cout, cin, kh, kw)
        out[p + "conv_block.0.bias"] = u(p + "cb", cout, -0.05, 0.05)
        out[p + "conv_block.1.weight"] = u(p + "g", cout, 0.7, 1.3)
        out[p + "conv_block.1.bias"] = u(p + "b", cout, -0.1, 0.1)
        out[p + "conv_block.1.running_mean"] = u(p + "m", cout, -0.1, 0.1)
        out[p + "conv_block.1.running_var"] = u(p + "v", cout, 0.7, 1.3)
        out[p + "conv_block.1.num_batches_tracked"] = np.array(100, dtype=np.int64)
    return out


LPIPS_ALEX_CONVS = (("net.slice1.0", 3, 64, 11), ("net.slice2.3", 64, 192, 5), ("net.slice3.6", 192, 384, 3),
                    ("net.slice4.8", 384, 256, 3), ("net.slice5.10", 256, 256, 3))


def make_lpips_state_dict(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded weights with the state-dict keys of lpips.LPIPS(net='alex', version='0.1') (lpips==0.1.4, requirement.txt:11).
    The real ones -- torchvision's AlexNet features and the package's alex.pth linear heads -- are not in the reference
    repository, so parity of this net is structural: same generator on both sides.  He-uniform convolutions, small biases,
    non-negative linear heads (the package clamps them at >= 0 in training), the package's fixed shift / scale."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def u(name, n, lo, hi):
        return (lo + (hi - lo) * uniform01(n, _stream_id("lpips." + name, seed))).astype(np.float32)

    out["scaling_layer.shift"] = np.array([-.030, -.088, -.188], np.float32).reshape(1, 3, 1, 1)
    out["scaling_layer.scale"] = np.array([.458, .448, .450], np.float32).reshape(1, 3, 1, 1)
    for name, cin, cout, k in LPIPS_ALEX_CONVS:
        b = np.sqrt(6.0 / (cin * k * k))
        out[name + ".weight"] = u(name + ".w", cout * cin * k * k, -b, b).reshape(cout, cin, k, k)
        out[name + ".bias"] = u(name + ".b", cout, -0.05, 0.1)
    for i, (_, _, cout, _) in enumerate(LPIPS_ALEX_CONVS):
        w = u(f"lin{i}", cout, 0.0, 2.0 / cout).reshape(1, cout, 1, 1)
        out[f"lin{i}.model.1.weight"] = w
        out[f"lins.{i}.model.1.weight"] = w
    return out


def synthetic_sync_batch(batch: int, seed: int = 0, frames_t: int = 5, height: int = 96, width: int = 96):
    """(mel [B,1,80,16], rgb_window_pos [B,3,T,H,W], rgb_window_neg [B,3,T,H,W]) in the layouts of
    someones_lip_dataset.py:331 / training.py:548-553.  Smooth images plus noise, so that the two windows are
    neither identical nor orthogonal in embedding space."""
    def u(name, n):
        return uniform01(n, _stream_id("syncbatch." + name, seed)).astype(np.float32)
    mel = (u("mel", batch * 80 * 16).reshape(batch, 1, 80, 16) * 4.0 - 2.0).astype(np.float32)
    yy, xx = np.meshgrid(np.linspace(0, 1, height, dtype=np.float32), np.linspace(0, 1, width, dtype=np.float32), indexing="ij")
    base = 0.5 + 0.3 * np.sin(6.0 * xx + 3.0 * yy)
    pos = np.clip(base[None, None, None] + 0.4 * (u("pos", batch * 3 * frames_t * height * width).reshape(batch, 3, frames_t, height, width) - 0.5), 0, 1)
    neg = np.clip(base[None, None, None, ::-1] + 0.4 * (u("neg", batch * 3 * frames_t * height * width).reshape(batch, 3, frames_t, height, width) - 0.5), 0, 1)
    return mel, pos.astype(np.float32), np.ascontiguousarray(neg).astype(np.float32)


def synthetic_warp_coords(n_frames: int, face_h: int = 500, face_w: int = 500, seed: int = 4) -> np.ndarray:
    """Surrogate `coords/%05d.npy`: float32 [N,FH,FW,2] = the identity grid of
    F.grid_sample(align_corners=False) + a per-frame rigid perturbation (rotation <= 3 deg, shift <= 0.02) + N(0,1e-3)
    jitter, clamped to [-1,1] as /root/reference/preprocess/face_tracker.py:606 clamps the real grids."""
    def u(name, n):
        return uniform01(n, _stream_id("warp." + name, seed))
    ys, xs = np.meshgrid(np.arange(face_h), np.arange(face_w), indexing="ij")
    ident = np.stack([(2 * xs + 1) / face_w - 1, (2 * ys + 1) / face_h - 1], -1)          # [FH,FW,2] (x, y)
    ang = (u("angle", n_frames) - 0.5) * (6.0 * np.pi / 180.0)
    rot = np.stack([np.stack([np.cos(ang), -np.sin(ang)], -1), np.stack([np.sin(ang), np.cos(ang)], -1)], -2)   # [N,2,2]
    shift = (u("shift", n_frames * 2).reshape(n_frames, 1, 1, 2) - 0.5) * 0.04
    n = n_frames * face_h * face_w * 2
    jitter = np.sqrt(-2.0 * np.log(1.0 - u("j1", n))) * np.cos(2.0 * np.pi * u("j2", n)) * 1e-3
    grid = np.einsum("hwk,fjk->fhwj", ident, rot) + shift + jitter.reshape(n_frames, face_h, face_w, 2)
    return np.clip(grid, -1.0, 1.0).astype(np.float32)


def synthetic_image(shape, seed: int, name: str = "img") -> np.ndarray:
    """uniform(0,1) float32 image(s) of `shape` from the same counter-based generator (seeds 2, 3)."""
    n = int(np.prod(shape))
    return uniform01(n, _stream_id("image." + name, seed)).astype(np.float32).reshape(shape)
