"""Helpers for the bf16 training-mode tests: decode the device layouts (csrc/s2l_bf16.h) and emulate the kernels'
arithmetic on the CPU (bf16-rounded operands, fp32 accumulation), so that the kernels can be checked tightly."""
import numpy as np
import torch


def bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def tiles_to_rows(img_i16: torch.Tensor, n_layers: int, np_rows: int, feats: int = 256) -> torch.Tensor:
    """int16 view of the bf16 image [L][np/32][feats/32][half 2][lane = n + 32 hh][4 (a & 1) + c] (feature 32R + 8a + 4hh + c of
    row n, a = 2 half + (a & 1); csrc/s2l_bf16.h) -> fp32 [L][np][feats]."""
    nb = feats // 32
    t = img_i16.cpu().view(torch.bfloat16).to(torch.float32).reshape(n_layers, np_rows // 32, nb, 2, 2, 32, 2, 4)   # L,G,R,half,hh,n,a1,c
    t = t.permute(0, 1, 5, 2, 3, 6, 4, 7)                                                                          # L,G,n,R,half,a1,hh,c
    return t.reshape(n_layers, np_rows, feats)


def masks_to_rows(masks_i64: torch.Tensor, np_rows: int) -> torch.Tensor:
    """ReLU mask dwords [8 layers][np/32 groups][4 stages][64 lanes = n + 32 hh] (csrc/train_bf16.hip: bit 15 - d = low half of
    the lane's dword d of the stage, bit 31 - d = its high half, d = 8 which + 2 a + p, halves c = 2p, 2p + 1) -> bool
    [8][np][256], feature 32 (2 q + which) + 8 a + 4 hh + c of row 32 group + n."""
    m = masks_i64.cpu().numpy().view(np.uint32).reshape(8, np_rows // 32, 4, 2, 32)            # L, G, q, hh, n
    out = np.zeros((8, np_rows // 32, 32, 256), dtype=bool)
    for q in range(4):
        for which in range(2):
            for a in range(4):
                for p in range(2):
                    d = 8 * which + 2 * a + p
                    for half in range(2):
                        bit = ((m[:, :, q] >> np.uint32((31 if half else 15) - d)) & np.uint32(1)).astype(bool)   # L, G, hh, n
                        for hh in range(2):
                            out[:, :, :, 32 * (2 * q + which) + 8 * a + 4 * hh + 2 * p + half] = bit[:, :, hh, :]
    return torch.from_numpy(out.reshape(8, np_rows, 256))


def blob_offsets():
    """Offsets (in floats) of the fp32 blob sections, evaluated from csrc/s2l_layout.h."""
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech2lip_amd", "csrc", "s2l_layout.h")
    env = {"align4": lambda v: (v + 3) & ~3}
    for m in re.finditer(r"constexpr\s+(?:int64_t|int)\s+(\w+)\s*=\s*([^;]+);", open(path).read()):
        env[m.group(1)] = eval(m.group(2).replace("int64_t(", "(").replace("/", "//"), {}, env)
    return env


def folded_from_blob(packed: torch.Tensor):
    """G0, c0, G5, c5 exactly as the device folded them (fp32 blob of s2l_pack_weights)."""
    o = blob_offsets()
    p = packed.cpu()
    G0 = p[o["OFF_G0"]:o["OFF_G0"] + 256 * 128].reshape(256, 128)[:, :126]
    G5 = p[o["OFF_G5"]:o["OFF_G5"] + 256 * 128].reshape(256, 128)[:, :126]
    return G0, p[o["OFF_BG0"]:o["OFF_BG0"] + 256], G5, p[o["OFF_BG5"]:o["OFF_BG5"] + 256]


def folded(sd):
    """G0, c0, G5, c5 and the hidden matrices from a state dict of fp32 tensors (what pack_fold_general builds)."""
    def fold(w_first, names):
        C = torch.cat([sd[f"{n}.weight"] for n in names], dim=1)                     # [256,126]
        bsum = sum(sd[f"{n}.bias"] for n in names)
        return w_first @ C, w_first @ bsum
    W0, W5 = sd["pts_linears.0.weight"], sd["pts_linears.5.weight"]
    G0, c0 = fold(W0, ("fc_uv", "fc_audio", "fc_time"))
    G5, c5 = fold(W5[:, :256], ("fc_uv_skip", "fc_audio_skip", "fc_time_skip"))
    return G0, c0 + sd["pts_linears.0.bias"], G5, c5 + sd["pts_linears.5.bias"]


def forward_emu(sd, x):
    """x fp32 [N,128] (126 used) -> (rgb [N,3], h [8,N,256] as stored (bf16 values), pre-activation list)."""
    G0, c0, G5, c5 = folded(sd)
    xb = bf(x[:, :126])
    hs, zs = [], []
    z = xb @ bf(G0).t() + c0
    for L in range(8):
        if L > 0:
            W = sd[f"pts_linears.{L}.weight"]
            if L == 5:
                z = hs[-1] @ bf(W[:, 256:]).t() + xb @ bf(G5).t() + c5
            else:
                z = hs[-1] @ bf(W).t() + sd[f"pts_linears.{L}.bias"]
        zs.append(z)
        hs.append(bf(torch.relu(z)))
    rgb = hs[-1] @ bf(sd["output_linear.weight"]).t() + sd["output_linear.bias"]
    return rgb, torch.stack(hs), zs


def forward_teacher_forced(sd, x, h_dev, fold=None):
    """Per-layer expectation given the DEVICE's previous-layer activations (so that bf16 rounding flips do not cascade):
    returns expected h [8,N,256].  fold = folded_from_blob(packed) uses the device's own folded first/skip matrices."""
    G0, c0, G5, c5 = fold if fold is not None else folded(sd)
    xb = bf(x[:, :126])
    out = [bf(torch.relu(xb @ bf(G0).t() + c0))]
    for L in range(1, 8):
        W = sd[f"pts_linears.{L}.weight"]
        if L == 5:
            z = h_dev[L - 1] @ bf(W[:, 256:]).t() + xb @ bf(G5).t() + c5
        else:
            z = h_dev[L - 1] @ bf(W).t() + sd[f"pts_linears.{L}.bias"]
        out.append(bf(torch.relu(z)))
    return torch.stack(out)


def assert_bf16_close(got, exp, what, frac=3e-3):
    """Equal up to fp32 summation noise, except that a value within that noise of a bf16 rounding tie (or of the ReLU
    threshold) may land on the neighbouring bf16 value: a small fraction, by at most one bf16 ulp."""
    d = (got - exp).abs()
    off = d > 1e-5 * (1 + exp.abs())
    assert float(off.float().mean()) < frac, (what, float(off.float().mean()))
    assert bool((d <= 2.0 ** -7 * exp.abs() * 1.01 + 2e-5).all()), (what, float(d.max()))


def backward_teacher_forced(sd, drgb, masks, g_dev, fold):
    """Expected dz tiles g_7..g_0 [8,N,256] (index = layer) and d audio [N,64], each step from the DEVICE's previous
    gradient (bf16 values) so that rounding flips do not cascade."""
    G0, _, G5, _ = fold
    m = masks.to(torch.float32)
    exp = [None] * 8
    exp[7] = bf((bf(drgb) @ bf(sd["output_linear.weight"])) * m[7])
    for l in range(7, 0, -1):
        W = sd[f"pts_linears.{l}.weight"]
        W = W[:, 256:] if l == 5 else W
        exp[l - 1] = bf((g_dev[l] @ bf(W)) * m[l - 1])
    dxa = g_dev[5] @ bf(G5[:, 42:106]) + g_dev[0] @ bf(G0[:, 42:106])
    return torch.stack(exp), dxa
