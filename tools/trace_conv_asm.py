"""Experiment: per-tile timestamps of the assembly 3x3 convolution (csrc/gen_conv_body.py) in one U-Net eval forward at 500x500.
    mkdir -p /tmp/cinc && S2L_CONV_TRACE=1 python speech2lip_amd/csrc/gen_conv_body.py /tmp/cinc
    tools/build_variant.sh unet.hip ab/trace_conv_asm.so -I/tmp/cinc -DS2L_EXP_TRACE
    python tools/trace_conv_asm.py ab/trace_conv_asm.so [frames=4 (tiles of a launch <= 8192)]
Slots per tile: 0 start, 1 accumulators initialised, 2 + c end of chunk c (after its barrier), 20 end of the epilogue."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2L_LIB"] = os.path.abspath(sys.argv[1])
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
tr = trace.cpu().numpy().reshape(12, 8192, 24)
names = ["inc.2", "down1.1", "down1.2", "down2.1", "down2.2", "up1.1", "up1.2", "up2.1", "up2.2"]
for which in range(9):
    t = tr[which]
    t = t[t[:, 0] > 0]
    if not len(t) or t[0, 20] == 0:
        print(f"{names[which]:8s}: no assembly-kernel timestamps"); continue
    nch = int((t[0, 2:18] > 0).sum())
    init = t[:, 1] - t[:, 0]
    chunks = np.diff(t[:, 1:2 + nch], axis=1)
    epi = t[:, 20] - t[:, 1 + nch]
    tot = t[:, 20] - t[:, 0]
    med = lambda a: f"{np.median(a):7.0f} (p10 {np.percentile(a, 10):6.0f}, p90 {np.percentile(a, 90):6.0f})"
    print(f"{names[which]:8s}: {len(t):5d} tiles x {nch:2d} chunks; tile {med(tot)} cycles = {np.median(tot) / (nch * 184.32):.1f} % of its MFMA cycles; "
          f"init {np.median(init):.0f}; first chunk {med(chunks[:, 0])}; middle {med(chunks[:, 1:-1]) if nch > 2 else '-'}; "
          f"last {med(chunks[:, -1])}; epilogue {med(epi)}")
print("(s_memtime counts shader cycles; a chunk's MFMAs = 18 432 cycles)")
if os.environ.get("S2L_TRACE_DETAIL"):
    which = int(os.environ["S2L_TRACE_DETAIL"])
    t = tr[which]; t = t[t[:, 0] > 0]
    nch = int((t[0, 2:18] > 0).sum())
    chunks = np.diff(t[:, 1:2 + nch], axis=1)
    print(f"{names[which]} per chunk index: median", [int(np.median(chunks[:, c])) for c in range(nch)])
    print("   p10", [int(np.percentile(chunks[:, c], 10)) for c in range(nch)])
    print("   gap between a tile's end and the next one's start (same workgroup):", int(np.median(t[1:64, 0] - t[:63, 20])))
