"""BASELINE config 5 timing: one training step = 4-tap ensemble forward + MSE + full backward for a
batch of frames at 96x96 (fp32 exact-parity mode).  FLOPs per step (as-written
model, fwd + dgrad + wgrad, 4 taps): 3 * 4 * 2 * 644,864 * HW * frames.
    python tools/bench_train.py [frames=64] [fp32|bf16] [--profile]
bf16 = the precision BASELINE config 5 names (bf16 MFMA operands and saved state, fp32 accumulation and master weights)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 64
PREC = args[1] if len(args) > 1 else "fp32"
H = Wd = 96
m = s2l.TalkingFace(dev, s2l.may_config(H, Wd)).train()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
opt = torch.optim.Adam([p for n, p in m.named_parameters() if not n.startswith("coord_linears")], lr=1e-4)
audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
target = torch.rand(B, H * Wd, 3, device=dev)
step = s2l.LipTrainStep(m, H, Wd, precision=PREC)
u01 = [0.5] * B
def one():
    loss, g, _ = step.loss_and_grads(audio, list(range(B)), target, u01)
    s2l.training.apply_grads(m, g)
    opt.step()
    return loss
l0 = float(one()); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    l = one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = 3 * 4 * 2 * 644_864 * H * Wd * B
print(json.dumps({"config": f"training step, {B} frames 96x96, {PREC}" + (" parity mode" if PREC == "fp32" else " MFMA, fp32 accumulate + master weights") + ", Adam", "ms_per_step": round(dt * 1e3, 2),
                  "frames_per_s": round(B / dt, 1), "as_written_tflop_per_step": round(flops / 1e12, 3),
                  "tflops": round(flops / dt / 1e12, 1), "loss_first": l0, "loss_last": float(l),
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))

if "--profile" in sys.argv:       # per-phase wall times of one step (synchronising after each C-ABI call family)
    import collections
    from speech2lip_amd import _abi
    lib = _abi.load()
    acc = collections.OrderedDict()
    def wrap(name):
        fn = getattr(lib, name)
        def timed(*a):
            torch.cuda.synchronize(); t = time.perf_counter(); r = fn(*a); torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t); return r
        return timed
    class Proxy:
        def __getattr__(self, name):
            return wrap(name)
    step.lib = Proxy()
    torch.cuda.synchronize(); t = time.perf_counter(); one(); torch.cuda.synchronize(); tot = time.perf_counter() - t
    print(json.dumps({"profiled_step_ms": round(tot * 1e3, 2), "phases_ms": {k: round(v * 1e3, 3) for k, v in acc.items()}}))
