#!/usr/bin/env python3
"""Time the half-width 3x3 weight-gradient kernel alone (s2l_debug_conv_wgrad_h) at the U-Net's layer shapes, 8 frames of 500 x 500.
    python tools/dev/bench_conv_wgrad.py [frames=8]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from speech2lip_amd import _abi

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
lib = _abi.load()
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
part = torch.empty(64 * 256 * 128 * 9, dtype=torch.float32, device=dev)
for CA, CB, cout, H, W in ((64, 0, 64, 500, 500), (64, 64, 64, 500, 500), (64, 0, 128, 250, 250), (128, 128, 128, 250, 250), (128, 0, 128, 125, 125)):
    cin = CA + CB
    dz = torch.randint(-3000, 3000, (F * (cout // 32) * H * W * 32,), dtype=torch.int16, device=dev)
    a = torch.randint(-3000, 3000, (F * (CA // 32) * H * W * 32,), dtype=torch.int16, device=dev)
    b = torch.randint(-3000, 3000, (F * (CB // 32) * H * W * 32,), dtype=torch.int16, device=dev) if CB else None
    dw = torch.empty(cout * cin * 9, dtype=torch.float32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: _abi.check(lib.s2l_debug_conv_wgrad_h(p(dz), p(a), CA, p(b), CB, cout, p(part), p(dw), H, W, F, st), "wgrad")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * F * H * W * 9 * cin * cout
    print(f"{cin:3d} -> {cout:3d} @ {H}x{W} x {F}: {us:7.1f} us (kernel + partial-sum pass), {fl / us / 1e6:6.0f} TFLOP/s")
