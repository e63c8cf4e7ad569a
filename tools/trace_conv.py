"""Experiment: per-workgroup timestamps of the U-Net's 3x3 convolution kernel (one eval forward, 500x500).
    tools/build_variant.sh unet.hip ab/trace_conv.so -DS2L_EXP_TRACE
    python tools/trace_conv.py ab/trace_conv.so [frames=2 (<= 4)] [launch=1]
Findings (round 2): a 16-channel chunk takes 38.7 k cycles when two workgroups share a CU (36.9 k = the MFMAs of both) and
19.8-20.9 k alone (18.4 k); prologue 6.2 k and epilogue 6.5 k per workgroup; staggering the two slots of a CU changes nothing."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2L_LIB"] = os.path.abspath(sys.argv[1])
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(F, 500, 500, 3, device=dev)
for _ in range(2):
    u.forward_nhwc(x)
torch.cuda.synchronize()
trace = torch.zeros(12 * 8192 * 24, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.s2l_debug_set_conv_trace.argtypes = [ctypes.c_void_p]
lib.s2l_debug_set_conv_trace(trace.data_ptr())
u.forward_nhwc(x)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(12, 8192, 24)[which]
t = t[t[:, 0] > 0]
n = len(t)
nch = int((t[0, 2:18] > 0).sum())
print(f"launch {which}: {n} workgroups, {nch} chunks each")
life = t[:, 20] - t[:, 0]
pro = t[:, 1] - t[:, 0]
chunks = np.diff(t[:, 1:2 + nch], axis=1)
epi = t[:, 20] - t[:, 1 + nch]
print(f"workgroup lifetime median {np.median(life):.0f} (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
print(f"  prologue (first fetch + commit + barrier) median {np.median(pro):.0f}; per chunk median {np.median(chunks):.0f} (p10 {np.percentile(chunks, 10):.0f}, p90 {np.percentile(chunks, 90):.0f}; ideal alone 18432, sharing a SIMD 36864); epilogue {np.median(epi):.0f}")
