"""Experiment: per-stage phase timestamps of the assembly bf16 forward (csrc/gen_fwd16_body.py), wave 0 of every workgroup.
    mkdir -p /tmp/finc_tr && S2L_FWD_TRACE=1 python speech2lip_amd/csrc/gen_fwd16_body.py /tmp/finc_tr
    tools/build_variant.sh train_bf16.hip ab/tf_trace.so -I/tmp/finc_tr -DS2L_EXP_TRACE
    python tools/trace_fwd16.py ab/tf_trace.so [frames=64]
Slots per (tile, stage): 0 k-loop starts, 1 k-loop done, 2 vmcnt(0) / lgkmcnt(0) passed, 3 barrier passed, 4 epilogue done."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2L_LIB"] = os.path.abspath(sys.argv[1])
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.talking_face import _ptr, _stream
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
m = s2l.TalkingFace(dev, s2l.may_config(96, 96)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
lib = _abi.load()
N = 4 * 96 * 96 * B
Np = int(lib.s2l_bf16_rows_padded(N)); lay = Np * 256
torch.manual_seed(0)
x = torch.randn(N, 128, device=dev) * 0.5
xT = torch.empty(Np * 128, dtype=torch.int16, device=dev)
lib.s2l_rows_to_tiles_bf16(_ptr(x), 128, _ptr(xT), N, _stream())
hT = torch.empty(8 * lay, dtype=torch.int16, device=dev)
masks = torch.empty(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
rgb = torch.empty(N, 3, device=dev)
pb, pf = m.packed_weights_bf16(), m.packed_weights()
assert lib.s2l_set_bf16_forward_kernel(0) == 0
fwd = lambda: lib.s2l_train_forward_bf16(_ptr(pb), _ptr(pf), _ptr(xT), _ptr(hT), _ptr(masks), _ptr(rgb), N, _stream())
for _ in range(2):
    fwd()
torch.cuda.synchronize()
ntiles = Np // 256
trace = torch.zeros(ntiles * 32 * 8, dtype=torch.int64, device=dev)
raw = ctypes.CDLL(os.environ["S2L_LIB"])
raw.s2l_debug_set_fwd_trace.argtypes = [ctypes.c_void_p]
assert raw.s2l_debug_set_fwd_trace(trace.data_ptr()) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); fwd(); b.record(); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(ntiles, 32, 8)
t = t[t[:, 0, 0] > 0]
print(f"{len(t)} tiles traced, kernel {a.elapsed_time(b):.3f} ms (with tracing)")
names = ["k-loop", "wait vmcnt/lgkm", "barrier", "epilogue", "to next stage's k-loop"]
d = np.diff(t[:, :, :5], axis=2)                                    # [tile, stage, 4]
nxt = np.concatenate([t[:, 1:, 0] - t[:, :-1, 4], np.zeros((len(t), 1), dtype=np.int64)], axis=1)
for kind, stages in (("layer 0 (x only: 32 MFMAs)", range(0, 4)), ("layer 5 (x + h: 96 MFMAs)", range(20, 24)),
                     ("other layers (64 MFMAs = 2048 cycles)", [s for s in range(4, 32) if not 20 <= s < 24])):
    sel = list(stages)
    print(kind)
    for i, n in enumerate(names[:4]):
        v = d[:, sel, i].ravel()
        print(f"   {n:24s} median {np.median(v):7.0f}  p10 {np.percentile(v, 10):7.0f}  p90 {np.percentile(v, 90):7.0f}")
    v = nxt[:, [s for s in sel if s < 31]].ravel()
    print(f"   {names[4]:24s} median {np.median(v):7.0f}  p10 {np.percentile(v, 10):7.0f}  p90 {np.percentile(v, 90):7.0f}")
tot = t[:, 31, 4] - t[:, 0, 0]
print(f"tile (32 stages) median {np.median(tot):.0f} cycles = {np.median(tot) / 32:.0f} per stage")
if os.environ.get("S2L_TRACE_DETAIL"):
    print("per stage: median k-loop / wait / epilogue, and p90 of the k-loop")
    for s_ in range(32):
        print(f"  stage {s_:2d} (layer {s_ >> 2} q {s_ & 3}): {np.median(d[:, s_, 0]):6.0f} {np.median(d[:, s_, 1]):6.0f} {np.median(d[:, s_, 3]):6.0f}   p90 {np.percentile(d[:, s_, 0], 90):6.0f}"
              f"   tile-order effect: first tile of a workgroup {np.median(d[:256, s_, 0]):6.0f}, later {np.median(d[256:, s_, 0]):6.0f}")
