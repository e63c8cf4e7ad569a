"""BASELINE config 5 timing: one training step = 4-tap ensemble forward + MSE + full backward for a
batch of frames at 96x96 (fp32 exact-parity mode).  FLOPs per step (as-written
model, fwd + dgrad + wgrad, 4 taps): 3 * 4 * 2 * 644,864 * HW * frames.
    python tools/bench_train.py [frames=64]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = Wd = 96
m = s2l.TalkingFace(dev, s2l.may_config(H, Wd)).train()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
opt = torch.optim.Adam([p for n, p in m.named_parameters() if not n.startswith("coord_linears")], lr=1e-4)
audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
target = torch.rand(B, H * Wd, 3, device=dev)
step = s2l.LipTrainStep(m, H, Wd)
u01 = [0.5] * B
def one():
    loss, g, _ = step.loss_and_grads(audio, list(range(B)), target, u01)
    s2l.training.apply_grads(m, g)
    opt.step()
    return loss
l0 = float(one()); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 3
for _ in range(n):
    l = one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = 3 * 4 * 2 * 644_864 * H * Wd * B
print(json.dumps({"config": f"training step, {B} frames 96x96, fp32 parity mode, Adam", "ms_per_step": round(dt * 1e3, 2),
                  "frames_per_s": round(B / dt, 1), "as_written_tflop_per_step": round(flops / 1e12, 3),
                  "tflops": round(flops / dt / 1e12, 1), "loss_first": l0, "loss_last": float(l),
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
