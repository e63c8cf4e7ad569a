#!/usr/bin/env python3
"""A/B of the two bf16 backward kernels (C++ bwd_bf16_kernel vs the generated-assembly bwd_asm_bf16_kernel): dz images must be
bit-identical, the assembly kernel's per-tile audio sums must equal the per-row dxa of the C++ kernel summed per tile.
    python tools/cmp_bf16_backward.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.talking_face import _ptr, _stream

dev = torch.device("cuda:0")
m = s2l.TalkingFace(dev, s2l.may_config(96, 96)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
lib = _abi.load()
pb, pf = m.packed_weights_bf16(), m.packed_weights()
ok = True
for N in [int(a) for a in sys.argv[1:]] or [256, 200, 256 * 3, 256 * 700 + 77]:
    Np = int(lib.s2l_bf16_rows_padded(N)); lay = Np * 256
    torch.manual_seed(N)
    x = torch.randn(N, 128, device=dev) * 0.5
    xT = torch.empty(Np * 128, dtype=torch.int16, device=dev)
    lib.s2l_rows_to_tiles_bf16(_ptr(x), 128, _ptr(xT), N, _stream())
    hT = torch.empty(8 * lay, dtype=torch.int16, device=dev)
    masks = torch.empty(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
    rgb, drgb = torch.empty(N, 3, device=dev), torch.randn(N, 3, device=dev)
    _abi.check(lib.s2l_train_forward_bf16(_ptr(pb), _ptr(pf), _ptr(xT), _ptr(hT), _ptr(masks), _ptr(rgb), N, _stream()), "fwd")
    dz_c, dz_a = torch.zeros(8 * lay, dtype=torch.int16, device=dev), torch.full((8 * lay,), 0x7fc0, dtype=torch.int16, device=dev)
    dxa = torch.zeros(Np, 64, device=dev)
    tiles = torch.full((Np // 256, 64), float("nan"), device=dev)
    _abi.check(lib.s2l_train_backward_bf16(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dz_c), _ptr(dxa), N, _stream()), "bwd")
    _abi.check(lib.s2l_train_backward_bf16_tiles(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dz_a), _ptr(tiles), N, _stream()), "bwd asm")
    torch.cuda.synchronize()
    same = torch.equal(dz_c, dz_a)
    ref = dxa.view(Np // 256, 256, 64).double().sum(1)
    err = float((tiles.double() - ref).abs().max() / ref.abs().max())
    bad_layers = [l for l in range(8) if not torch.equal(dz_c[l * lay:(l + 1) * lay], dz_a[l * lay:(l + 1) * lay])]
    print(f"rows {N}: dz images identical: {same} (layers that differ: {bad_layers}); tile sums rel err {err:.2e}")
    ok = ok and same and err < 1e-5
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
