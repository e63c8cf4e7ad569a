"""Experiment: does the render kernel's speed depend on the DATA (weights/activations)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
H = Wd = 96; F = 500
audio = torch.from_numpy(W.synthetic_audio(F, 1).astype(np.float32)).to(dev)
def run(name, mutate):
    m = s2l.TalkingFace(dev, s2l.may_config(H, Wd)).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
    with torch.no_grad():
        mutate(m)
    out = torch.empty(F, H, Wd, 3, device=dev)
    for _ in range(2):
        m.render_clip(audio, list(range(F)), H, Wd, out=out)
    torch.cuda.synchronize()
    ev = []
    for _ in range(3):
        m.render_clip(audio, list(range(F)), H, Wd, out=out, _events=ev)
    torch.cuda.synchronize()
    ms = np.mean([a.elapsed_time(b) for a, b in ev])
    print(f"{name:34s} kernel {ms:8.3f} ms  -> {F / ms * 1e3:9.1f} frames/s   out rms {float(out.pow(2).mean().sqrt()):.3g}")
run("he weights (bench)", lambda m: None)
run("all weights zero", lambda m: [p.zero_() for p in m.parameters()])
run("hidden weights = 1e-3 const", lambda m: [p.fill_(1e-3) for n, p in m.named_parameters() if n.startswith("pts_linears")])
run("hidden weights sign-only +-0.05", lambda m: [p.copy_(torch.sign(p) * 0.05) for n, p in m.named_parameters() if n.startswith("pts_linears") and n.endswith("weight")])
run("torch-gain weights", lambda m: m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "torch", include_dead=True).items()}))
