"""Where the end-to-end inference driver's time goes on the host: loader alone, writer alone (frames/s each), by mode / thread count."""
import os, sys, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import speech2lip_amd as s2l
from tools import benchlib
dev = torch.device("cuda:0")
root = benchlib._dataset_tmp("may_face_crop_lip")
benchlib.write_synthetic_dataset(root, 640, train=False)
ds = s2l.SomeonesLipClip(root, "val", s2l.may_config(96, 96, data_path=root))
n = len(ds)
for mode in ("thread", "process", "process"):
    for workers in (8, 32, 64):
        st = s2l.ClipStreamer(ds, dev, 100, mode=mode, workers=workers)
        torch.cuda.synchronize(); t = time.perf_counter()
        for clip in st:
            pass
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        st.close()
        print(f"loader {mode:8s} workers {workers:3d}: {n / dt:7.0f} frames/s", flush=True)
frames = torch.randint(0, 255, (100, 500, 500, 3), dtype=torch.uint8, device=dev)
names = ["%05d" % k for k in range(100)]
out = os.path.join(os.path.dirname(root), "o")
for workers in (8, 32, 64):
    wr = s2l.FrameWriter(out, workers=workers)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(6):
        wr.submit(frames, names)
    wr.close()
    dt = time.perf_counter() - t
    print(f"writer threads {workers:3d}: {600 / dt:7.0f} frames/s (random-noise frames: worst case for the JPEG encoder)", flush=True)
smooth = s2l.to8b(ds.load(dev, 0, 100).rgb_face_ori)
for workers in (8, 32, 64):
    wr = s2l.FrameWriter(out, workers=workers)
    t = time.perf_counter()
    for _ in range(6):
        wr.submit(smooth, names)
    wr.close()
    print(f"writer threads {workers:3d}: {600 / (time.perf_counter() - t):7.0f} frames/s (smooth frames)", flush=True)
shutil.rmtree(os.path.dirname(root), ignore_errors=True)
