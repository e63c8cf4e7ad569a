import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.unet import nhwc_to_c32, c32_to_nhwc
dev = torch.device("cuda:0")
lib = _abi.load()
u = s2l.SimpleUnetLight().to(dev).train()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
tensors = u._tensors()
raw, raw16 = u._raw_blobs(tensors, u._table(tensors), True)
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
H, Wd, F, layer, cin, cout = 32, 16, 1, 1, 64, 64
g = torch.Generator().manual_seed(3)
ah = nhwc_to_c32(torch.randn(F, H, Wd, cin, generator=g).to(torch.bfloat16).to(dev))
out = torch.full((F, cout // 32, H, Wd, 32), -1, dtype=torch.int16, device=dev)
stat = torch.full((F * 1024 * 2 * cout,), float("nan"), device=dev)
blocks = ctypes.c_int(0)
_abi.check(lib.s2l_debug_convh_layer_stats(p(raw16), layer, p(ah), cin, None, 0, p(out), p(stat), ctypes.byref(blocks), H, Wd, F, st), "x")
torch.cuda.synchronize()
z = c32_to_nhwc(out).view(torch.bfloat16).double()[0]      # [H,W,C]
got = stat[:2 * cout].reshape(2, cout).double().cpu()
ref_s, ref_q = z.sum((0, 1)).cpu(), (z * z).sum((0, 1)).cpu()
print("blocks", blocks.value)
print("got s ", got[0][:8].tolist())
print("ref s ", ref_s[:8].tolist())
print("got q ", got[1][:8].tolist())
print("ref q ", ref_q[:8].tolist())
# candidates
cands = {"even cols": z[:, 0::2].sum((0, 1)), "odd cols": z[:, 1::2].sum((0, 1)), "even rows": z[0::2].sum((0, 1)), "odd rows": z[1::2].sum((0, 1)),
         "rows 0-15": z[:16].sum((0, 1)), "first 2 rows of each 4": torch.cat([z[r:r + 2] for r in range(0, 32, 4)]).sum((0, 1)),
         "last 2 rows of each 4": torch.cat([z[r + 2:r + 4] for r in range(0, 32, 4)]).sum((0, 1))}
for k, v in cands.items():
    print(f"{k:26s} max|got - cand| {float((got[0] - v.cpu()).abs().max()):.4f}")
# which channel of ref does got[c] match best (by sum of squares)?
perm = [int((ref_q - got[1][c]).abs().argmin()) for c in range(cout)]
print("q-match channel map", perm[:16], "...", perm[32:40])
