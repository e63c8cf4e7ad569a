import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from tools.cmp_convh import CONVS, p, bf16_bits
dev = torch.device("cuda:0")
lib = _abi.load()
u = s2l.SimpleUnetLight().to(dev).train()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
tensors = u._tensors()
raw, raw16 = u._raw_blobs(tensors, u._table(tensors), True)
layer, tr, F, H, Wd, gate = [int(v) for v in sys.argv[1:7]]
cin, cout = CONVS[layer]
if tr: cin, cout = cout, cin
g = torch.Generator().manual_seed(1)
a = torch.randn(F, H, Wd, cin, generator=g).to(torch.bfloat16).to(dev)
gt = torch.randn(F, H, Wd, cout, generator=g).clamp_min(0).to(torch.bfloat16).to(dev) if gate else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = torch.zeros(F, H, Wd, cout, device=dev)
_abi.check(lib.s2l_debug_conv_layer_f32(p(raw), p(raw16), layer, tr, p(a.float()), cin, p(None), 0, p(gt.float() if gate else None), p(ref), H, Wd, F, st), "ref")
out = torch.full((F, H, Wd, cout), -1, dtype=torch.int16, device=dev)
_abi.check(lib.s2l_convh_layer(p(raw16), layer, tr, p(a), cin, p(None), 0, p(gt), p(out), H, Wd, F, st), "convh")
torch.cuda.synchronize()
bad = (out != bf16_bits(ref))
unw = (out == -1).all(dim=-1)
for f in range(F):
    print("frame", f, "mismatching pixels (X), unwritten (U):")
    m = bad[f].any(dim=-1).cpu(); uw = unw[f].cpu()
    for y in range(H):
        print("%3d " % y + "".join("U" if uw[y, x] else "X" if m[y, x] else "." for x in range(Wd)))
    cm = bad[f].any(dim=0).any(dim=0).cpu()
    print("channels:", "".join("X" if c else "." for c in cm))
