"""What the MI355X sustains on dense bf16 MFMAs alone (v_mfma_f32_32x32x16_bf16 on registers, one or two waves per SIMD): the
denominator that the convolution / training kernels' TFLOP/s should be read against next to the 2.4-GHz datasheet figure.
    python tools/ubench_mfma.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2lip_amd import _abi
lib = _abi.load()
dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
for waves in (4, 8):
    for iters in (20000, 200000):
        _abi.check(lib.s2l_debug_bf16_mfma_rate(1000, waves, ctypes.c_void_p(sink.data_ptr()), st), "warm")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _abi.check(lib.s2l_debug_bf16_mfma_rate(iters, waves, ctypes.c_void_p(sink.data_ptr()), st), "rate")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flop = n_cu * waves * iters * 8 * 32768.0
        cyc = iters * 8 * 32 * (waves / 4)      # MFMA-pipe cycles per SIMD
        print(f"{waves} waves/CU, {iters} x 8 MFMAs per wave: {ms:8.2f} ms  {flop / ms / 1e9:7.0f} TFLOP/s  implied MFMA clock {cyc / ms / 1e6:.2f} GHz", flush=True)
