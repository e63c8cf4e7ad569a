"""BASELINE config 5 timing: one training step = 4-tap ensemble forward + MSE + full backward for a
batch of frames at 96x96 (fp32 exact-parity mode).  FLOPs per step (as-written
model, fwd + dgrad + wgrad, 4 taps): 3 * 4 * 2 * 644,864 * HW * frames.
    python tools/bench_train.py [frames=64] [fp32|bf16] [--profile] [--sync=S [--trainbn] [--group=frames per U-Net launch]] [--full [--trainbn | --early]]
bf16 = the precision BASELINE config 5 names (bf16 MFMA operands and saved state, fp32 accumulation and master weights)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 64
PREC = args[1] if len(args) > 1 else "fp32"
if "--full" in sys.argv:      # every term of the reference's stage-1 iteration (LPIPS, face, sync) on each of the B samples
    print(json.dumps(benchlib.bench_stage1_full(torch.device("cuda:0"), B, PREC, unet_train_mode="--trainbn" in sys.argv, early="--early" in sys.argv)))
    sys.exit(0)
sync = [a for a in sys.argv[1:] if a.startswith("--sync")]
if sync:      # --sync=S: the step with the lipsync_expert loss attached to S of the B samples
    print(json.dumps(benchlib.bench_train_sync(torch.device("cuda:0"), B, int(sync[0].split("=")[1]) if "=" in sync[0] else 8, PREC,
                                               unet_train_mode="--trainbn" in sys.argv, half_width_tensors="--fp32-tensors" not in sys.argv,
                                               frames_per_group=next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--group=")), None))))
    sys.exit(0)
res = benchlib.bench_train(torch.device("cuda:0"), B, PREC)
step, one = res.pop("_step"), res.pop("_one")
print(json.dumps(res))

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
