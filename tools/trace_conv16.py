"""Experiment: where a chunk of the U-Net's split-bf16 convolution kernel spends its cycles (wave 0 of every workgroup, s_memtime).
    tools/build_variant.sh unet.hip ab/trace_conv16.so -DS2L_EXP_TRACE
    python tools/trace_conv16.py ab/trace_conv16.so [frames=2]
Phases per chunk: 1 issue of the next chunk's loads (+ LDS-DMA), 2 the chunk's 108 MFMAs + operand reads, 3 barrier "done reading",
4 commit (hi / lo conversion + LDS writes + vmcnt), 5 barrier "published"; 0 = prologue."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2L_LIB"] = os.path.abspath(sys.argv[1])
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
PREC = sys.argv[3] if len(sys.argv) > 3 else "split"      # "bf16": the plain-bf16 form of the same kernels
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(F, 500, 500, 3, device=dev)
for _ in range(2):
    u.forward_nhwc(x, precision=PREC)
torch.cuda.synchronize()
trace = torch.zeros(12 * 8192 * 24, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.s2l_debug_set_conv_trace.argtypes = [ctypes.c_void_p]
lib.s2l_debug_set_conv_trace(trace.data_ptr())
u.forward_nhwc(x, precision=PREC)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(12, 8192, 24)
for launch in range(9):
    tl = t[launch]
    tl = tl[tl[:, 0] > 0]
    if not len(tl):
        continue
    nch = tl[:, 1]
    life = tl[:, 20] - tl[:, 0]
    ph = tl[:, 2:8] / np.maximum(nch[:, None], 1)
    names = ["prologue/chunk", "issue", "mfma", "barrier1", "commit", "barrier2"]
    if len(tl) <= 256:      # the persistent kernel: one record per workgroup, phases summed over all its chunks
        names = ["prologue+epilogues/chunk", "mfma+requests+commit", "barrier", "weights wait", "epilogue stores", "advance"]
    inloop = tl[:, 2:8].sum(axis=1)
    span = tl[:, 20].max() - tl[:, 0].min()
    print(f"layer {launch + 1}: {len(tl)} workgroups x {int(np.median(nch))} chunks, lifetime {np.median(life):.0f} cycles = {np.median(life / np.maximum(nch, 1)):.0f} per chunk; "
          + ", ".join(f"{n} {np.median(ph[:, k]):.0f}" for k, n in enumerate(names))
          + f"; per workgroup: prologue {np.median(tl[:, 2]):.0f}, epilogue {np.median(life - inloop):.0f}; launch span {span} cycles = "
          f"{span / (len(tl) / 256):.0f} per workgroup slot ({np.median(life) / (span / (len(tl) / 256)):.2f} of it inside a workgroup)")
