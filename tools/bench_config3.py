"""BASELINE config 3 end to end on one MI355X: 128x128 lip crop rendered for N frames
(audio encoder -> frame vectors -> fused MLP) + paste/head-pose-warp composite into 500x500 faces.
Inputs are synthetic and resident in HBM (SURVEY.md §8d recipe); frames are processed in batches so
that the pose grids (2 MB/frame) and observed frames (3 MB/frame) fit comfortably.
    python tools/bench_config3.py [frames=5000] [batch=500]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 500
UNET = "--unet" in sys.argv     # also run the post-fusion U-Net: the reference's full inference output (inference.py:167-178)
h = w = 128; FH = FW = 500; x0, y0 = 186, 300
m = s2l.TalkingFace(dev, s2l.may_config(h, w)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
g = torch.Generator(device=dev).manual_seed(2)
audio = torch.from_numpy(W.synthetic_audio(N, 1).astype(np.float32)).to(dev)
face = torch.rand(1, FH, FW, 3, device=dev, generator=g)
mask = torch.zeros(1, FH, FW, 3, device=dev); mask[:, y0:y0 + h, x0:x0 + w] = 1
# one batch worth of pose grids / observed frames, reused for every batch (content does not affect timing)
ys, xs = torch.meshgrid(torch.arange(FH, device=dev), torch.arange(FW, device=dev), indexing="ij")
ident = torch.stack([(2 * xs + 1) / FW - 1, (2 * ys + 1) / FH - 1], -1).float()
ang = (torch.rand(BATCH, device=dev, generator=g) - 0.5) * (6 * np.pi / 180)
rot = torch.stack([torch.stack([ang.cos(), -ang.sin()], -1), torch.stack([ang.sin(), ang.cos()], -1)], -2)
shift = (torch.rand(BATCH, 1, 1, 2, device=dev, generator=g) - 0.5) * 0.04
coord = (torch.einsum("hwk,fjk->fhwj", ident, rot) + shift + torch.randn(BATCH, FH, FW, 2, device=dev, generator=g) * 1e-3).clamp(-1, 1).contiguous()
gt = torch.rand(BATCH, FH, FW, 3, device=dev, generator=g)
lip = torch.empty(BATCH, h, w, 3, device=dev)
out = torch.empty(BATCH, FH, FW, 3, device=dev)
def run():
    for s in range(0, N, BATCH):
        n = min(BATCH, N - s)
        m.render_clip(audio[s:s + n], torch.arange(s, s + n, device=dev), h, w, out=lip[:n])
        m.composite_clip(lip[:n], face, gt[:n], mask, x0, y0, coord[:n], out=out[:n])
        if UNET:
            m.post_fusion_unet.forward_nhwc(out[:n], out=recon[:n])
recon = torch.empty(BATCH, FH, FW, 3, device=dev) if UNET else None
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
# parity: last batch, frame 0 of it, against the CPU oracle
from oracle import s2l_oracle as O
s = (N - 1) // BATCH * BATCH
sd = O.to_sd(W.make_state_dict(0, "he"))
with torch.no_grad():
    ref_lip = O.render_clip(sd, audio[s:s + 1].cpu(), [s], h, w)
    ref_new, _ = O.composite(ref_lip, face.cpu(), gt[:1].cpu(), mask.cpu(), x0, y0, coord[:1].cpu())
extra = {}
if UNET:
    with torch.no_grad():
        ref_recon = O.unet_forward(O.to_sd(W.make_unet_state_dict(0)), ref_new)
    extra = {"unet_rmse": float(f"{O.rmse(recon[0].cpu(), ref_recon[0]):.3e}"), "unet_psnr_db": round(O.psnr(recon[0].cpu(), ref_recon[0]), 1)}
print(json.dumps({"config": f"config 3: {N} frames, 128x128 lip + composite into 500x500" + (" + post-fusion U-Net" if UNET else "") + f", batches of {BATCH}",
                  "seconds": round(dt, 3), "frames_per_s": round(N / dt, 1),
                  "lip_flops_per_frame_g": 15.058, "lip_tflops": round(15.058e9 * N / dt / 1e12, 1),
                  "parity": {"lip_rmse": float(f"{O.rmse(lip[0].cpu(), ref_lip[0]):.3e}"),
                             "composite_rmse": float(f"{O.rmse(out[0].cpu(), ref_new[0]):.3e}"),
                             "composite_psnr_db": round(O.psnr(out[0].cpu(), ref_new[0]), 1), **extra}}))
