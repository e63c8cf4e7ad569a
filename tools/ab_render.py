"""A/B aid for kernel experiments on the renderer: sha256 of a 1000-frame 96x96 clip and its render time.
    S2L_LIB=ab/libs2l_base.so python tools/ab_render.py     # the other build of the same ABI
Two builds whose renderers perform the same arithmetic in the same order must print the same digest."""
import hashlib
import sys

import torch

from benchlib import W, make_model, _median_ms
import numpy as np

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
m = make_model(dev, 96, 96)
audio = torch.from_numpy(W.synthetic_audio(F, 1).astype(np.float32)).to(dev)
idx = torch.arange(F, device=dev)
with torch.no_grad():
    clip = m.render_clip(audio, idx, 96, 96)
    torch.cuda.synchronize()
    digest = hashlib.sha256(clip.cpu().numpy().tobytes()).hexdigest()
    ms = _median_ms(lambda: m.render_clip(audio, idx, 96, 96), reps=5, inner=3)
print({"frames": F, "sha256": digest[:16], "ms": round(ms, 3), "fps": round(F / ms * 1e3, 1)})
