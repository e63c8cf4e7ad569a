"""Reader worker of `data.FramePrefetcher(mode="process")`: a STAND-ALONE SCRIPT run as a child process.  It builds its own
`SomeonesLipClip` (the reference's `SomeonesLipDataset.__init__`, someones_lip_dataset.py:43-164) from the folder / mode / cfg the parent
sends as its first line, then answers one request per line {"index": i, "shm": name}: `load_one_frame(i)` (:242-399), every tensor of the
dictionary written into the shared-memory slab `name` back to back (64-byte aligned), one JSON manifest line back
({key: [offset, shape, dtype]} for tensors, {key: value} for the python scalars).  Why a process: the reader's PIL / numpy work (JPEG
decode, the 8-bit resize of the negative window; 10 - 18 ms per frame) holds the interpreter lock, so loader THREADS top out near one core.
torch is imported for the dictionary's tensors only; the HIP library is never loaded here."""
import json
import os
import sys

import numpy as np

_blocks = {}


def _attach(name):
    from multiprocessing import shared_memory
    b = _blocks.get(name)
    if b is None:
        b = _blocks[name] = shared_memory.SharedMemory(name=name)
        try:      # (the parent owns the block)
            from multiprocessing import resource_tracker
            resource_tracker.unregister(b._name, "shared_memory")
        except Exception:
            pass
    return b


def main():
    pkg = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(pkg))
    import importlib.util
    import torch      # noqa: F401  (data.py needs it; CPU only here)
    spec = importlib.util.spec_from_file_location("_s2l_data_standalone", os.path.join(pkg, "data.py"))
    D = importlib.util.module_from_spec(spec)      # data.py alone: not the package (whose __init__ imports the kernels' wrappers)
    sys.modules[spec.name] = D
    spec.loader.exec_module(D)
    head = json.loads(sys.stdin.readline())
    ds = D.SomeonesLipClip(head["folder"], head["mode"], cfg=head["cfg"], img_ext=head.get("img_ext", ".jpg"))
    ds.load_sync_fields = bool(head.get("sync_fields", True))
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            req = json.loads(line)
            d = ds.load_one_frame(int(req["index"]))
            buf = _attach(req["shm"]).buf
            man, off = {}, 0
            for k, v in d.items():
                if hasattr(v, "numpy") and hasattr(v, "dtype"):      # torch tensor
                    a = np.ascontiguousarray(v.numpy())
                elif isinstance(v, np.ndarray):
                    a = np.ascontiguousarray(v)
                else:
                    man[k] = {"value": v if isinstance(v, (int, float, str, bool)) else float(v)}
                    continue
                if off + a.nbytes > len(buf):
                    raise RuntimeError(f"shared-memory slab too small for {k}")
                np.ndarray(a.shape, a.dtype, buffer=buf, offset=off)[...] = a
                man[k] = {"off": off, "shape": list(a.shape), "dtype": str(a.dtype), "tensor": hasattr(v, "numpy") and not isinstance(v, np.ndarray)}
                off = (off + a.nbytes + 63) // 64 * 64
            sys.stdout.write(json.dumps(man) + "\n")
        except Exception as e:
            sys.stdout.write(json.dumps({"__error__": repr(e)}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
