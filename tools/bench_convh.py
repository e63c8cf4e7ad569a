"""Per-layer time of the half-width convolution kernel (csrc/convh.hip) at the training chain's shapes.
    python tools/bench_convh.py [frames=20] [size=500] [kernel: 0 eight waves interleaved | 1 four waves | 2 eight waves in alternating roles] [--nogate: the train-mode chain's launches] [--bstats: also time the input-gradient launches that leave stage 1 of the BatchNorm backward]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128), (256, 128), (128, 64), (128, 64), (64, 64)]
LVL = [0, 0, 1, 1, 2, 2, 1, 1, 0, 0]
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
dev = torch.device("cuda:0")
NOGATE = "--nogate" in sys.argv
BSTATS = "--bstats" in sys.argv
sys.argv = [a for a in sys.argv if not a.startswith("--")]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 500
if len(sys.argv) > 3 and int(sys.argv[3]) != 0 and "S2L_LIB" not in os.environ:
    _abi.reference_kernels().__enter__()      # forms 1 and 2 live in libs2l_hip_ref.so
lib = _abi.load()
if len(sys.argv) > 3:
    _abi.check(lib.s2l_set_unet_half_kernel(int(sys.argv[3])), "s2l_set_unet_half_kernel")      # 0: eight waves, 1: four
u = s2l.SimpleUnetLight().to(dev).train()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
tensors = u._tensors()
raw, raw16 = u._raw_blobs(tensors, u._table(tensors), True)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = 0.0
for tr in (0, 1):
    for l in range(1, 10):
        cin, cout = CONVS[l]
        if tr:
            cin, cout = cout, cin
        h = S >> LVL[l]
        cat = l in (6, 8) and not tr
        CA, CB = (cin // 2, cin // 2) if cat else (cin, 0)
        # (one frame of slack behind the tensors: the access-pattern pricing builds, S2L_CH_EXP & 16384, read up to 40 KiB past a tile's origin)
        a = torch.randn(F + 1, h, h, CA, device=dev).to(torch.bfloat16)[:F]
        b = torch.randn(F + 1, h, h, CB, device=dev).to(torch.bfloat16)[:F] if CB else None
        gate = tr and l in (1, 3, 5, 7, 9) and not NOGATE
        gt = torch.randn(F, h, h, cout, device=dev).clamp_min(0).to(torch.bfloat16) if gate else None
        out = torch.empty(F, h, h, cout, dtype=torch.int16, device=dev)
        call = lambda: _abi.check(lib.s2l_convh_layer(p(raw16), l, tr, p(a), CA, p(b), CB, p(gt), p(out), h, h, F, st), "convh")
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        fl = 2 * 9 * cin * cout * h * h * F
        tot += us
        print(f"L{l} {'dgrad' if tr else 'fwd  '} {cin:3d}->{cout:3d} @{h:3d} gate={int(bool(gate))}: {us:7.0f} us  {fl / us / 1e6:6.0f} TFLOP/s", flush=True)
        if BSTATS and tr and l in (1, 3, 5, 7, 9) and not gate:
            z = torch.randn(F, h, h, cout, device=dev).to(torch.bfloat16)
            rows = torch.randn(F, 512, device=dev)
            stat = torch.empty(F * 1024 * 2 * cout, device=dev)
            blocks = ctypes.c_int(0)
            call = lambda: _abi.check(lib.s2l_debug_convh_layer_bstats(p(raw16), l, p(a), p(z), p(rows), p(out), p(stat), ctypes.byref(blocks), h, h, F, st), "bstats")
            call(); torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                call()
            e1.record(); torch.cuda.synchronize()
            us2 = e0.elapsed_time(e1) / 5 * 1e3
            print(f"   + backward statistics ({blocks.value} tiles per frame): {us2:7.0f} us  (+{us2 - us:.0f} us; z is {F * h * h * cout * 2 / 1e6:.0f} MB)", flush=True)
print(f"total {tot / 1e3:.2f} ms per {F} frames")
