"""A/B: the assembly bf16 forward (kind 0, the default) against the C++ kernel (kind 1) on the same inputs (s2l_set_bf16_forward_kernel), per layer.
    python tools/cmp_bf16_forward.py [rows=200000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.talking_face import _ptr, _stream
import bf16_util as U
dev = torch.device("cuda:0")
m = s2l.TalkingFace(dev, s2l.may_config(96, 96)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
lib = _abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
Np = int(lib.s2l_bf16_rows_padded(N)); lay = Np * 256
torch.manual_seed(0)
x = torch.randn(N, 128, device=dev) * 0.5
xT = torch.zeros(Np * 128, dtype=torch.int16, device=dev)
lib.s2l_rows_to_tiles_bf16(_ptr(x), 128, _ptr(xT), N, _stream())
pb, pf = m.packed_weights_bf16(), m.packed_weights()
outs = []
for kind in (0, 1):
    assert lib.s2l_set_bf16_forward_kernel(kind) == 0
    hT = torch.zeros(8 * lay, dtype=torch.int16, device=dev)
    masks = torch.zeros(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
    rgb = torch.zeros(N, 3, device=dev)
    rc = lib.s2l_train_forward_bf16(_ptr(pb), _ptr(pf), _ptr(xT), _ptr(hT), _ptr(masks), _ptr(rgb), N, _stream())
    torch.cuda.synchronize()
    outs.append((rc, U.tiles_to_rows(hT, 8, Np), U.masks_to_rows(masks, Np), rgb.cpu()))
lib.s2l_set_bf16_forward_kernel(0)
(ra, ha, ma, ga), (rb, hb, mb, gb) = outs
print("rc", ra, rb)
bad = 0
for l in range(8):
    d = (ha[l] != hb[l])
    bad += int(d.sum()) + int((ma[l] != mb[l]).sum())
    print("layer", l, "h mismatches:", int(d.sum()), "of", d.numel(), "rows:", int(d.any(dim=1).sum()),
          "first rows", d.any(dim=1).nonzero()[:6].flatten().tolist(), "blocks", sorted(set((d.any(dim=0).nonzero().flatten() // 32).tolist()))[:8],
          "| masks differ:", int((ma[l] != mb[l]).sum()))
print("rgb max diff", float((ga - gb).abs().max()), "IDENTICAL" if bad == 0 and torch.equal(ga, gb) else "DIFFERENT")
