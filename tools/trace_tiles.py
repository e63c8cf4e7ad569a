"""Experiment: per-tile phase timestamps of the render kernel.
Needs an experiment build of the renderer with timestamps in its assembly body:
    S2L_RENDER_TRACE=1 python speech2lip_amd/csrc/gen_render_body.py /tmp/inc/render_body.inc
    tools/build_variant.sh render.hip ab/trace.so -I/tmp/inc -DS2L_EXP_TRACE
    python tools/trace_tiles.py ab/trace.so [frames]      (S2L_CUS=n: run on n CUs only)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2L_LIB"] = os.path.abspath(sys.argv[1])
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
H = Wd = 96; F = int(sys.argv[2]) if len(sys.argv) > 2 else 1008
m = s2l.TalkingFace(dev, s2l.may_config(H, Wd)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
audio = torch.from_numpy(W.synthetic_audio(F, 1).astype(np.float32)).to(dev)
for _ in range(3):
    m.render_clip(audio, list(range(F)), H, Wd)
torch.cuda.synchronize()
ntiles = (H * Wd // 16) * ((F + 11) // 12)
trace = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)
lib = ctypes.CDLL(os.environ["S2L_LIB"])
lib.s2l_debug_set_trace.argtypes = [ctypes.c_void_p]
assert lib.s2l_debug_set_trace(trace.data_ptr()) == 0
if os.environ.get("S2L_CUS"):      # run on fewer CUs (one workgroup each): is the clock, not the schedule, the limit?
    lib.s2l_set_render_cus.argtypes = [ctypes.c_int]
    assert lib.s2l_set_render_cus(int(os.environ["S2L_CUS"])) == 0
    from speech2lip_amd import _abi
    _abi.load().s2l_set_render_cus(int(os.environ["S2L_CUS"]))
ev = []
m.render_clip(audio, list(range(F)), H, Wd, _events=ev); torch.cuda.synchronize()
ms = ev[0][0].elapsed_time(ev[0][1])
t = trace.cpu().numpy().reshape(ntiles, 16)
names = ["q0+p0 -> h0", "layer0", "layer1", "layer2", "layer3", "layer4 (+q5,p5)", "layer5", "layer6", "out layer+store"]
d = np.diff(t[:, :10], axis=1)
print(f"tiles {ntiles}, kernel {ms:.3f} ms; per-phase s_memtime ticks: median / p10 / p90")
for i, n in enumerate(names):
    print(f"  {n:22s} {np.median(d[:, i]):10.0f} {np.percentile(d[:, i], 10):10.0f} {np.percentile(d[:, i], 90):10.0f}")
tot = t[:, 9] - t[:, 0]
per_wg = ntiles / float(os.environ.get('S2L_CUS', 256))
print(f"  tile total {np.median(tot):.0f} ticks; tiles per workgroup {per_wg:.2f}; => {np.median(tot) * per_wg / (ms * 1e-3) / 1e9:.3f} G ticks/s if tiles were back to back")
first = t[:256, 0]
print(f"  ideal MFMA cycles per tile: {(7 * 3072 + 192) * 32}")
