"""Decode worker of `data.ClipStreamer(mode="process")`: a STAND-ALONE SCRIPT run as a child process (`python _io_worker.py`; imports
numpy and PIL only, never torch or the HIP library), because PIL's JPEG decoder keeps the interpreter lock -- 8 threads decode no
faster than one (measured: 634 vs 560 images/s on 8 cores), which capped the inference driver at ~800 frames/s whatever the GPU did.
One request per line on stdin (JSON): decode one observed frame (`ori_images_face/%05d.jpg`, someones_lip_dataset.py:272-275) and load one
pose grid (`coords/%05d.npy`, :251-262) straight into the parent's shared-memory staging blocks; one "ok" / "err ..." line back.
A request {"detach": [names]} (sent by `ClipStreamer.close()`) unmaps those blocks here; as a backstop the cache of mappings is capped
(least recently used first), so a parent that never closes its streamers cannot grow this process without bound."""
import json
import sys

import numpy as np

_blocks = {}          # name -> SharedMemory, in order of last use (dicts keep insertion order)
_MAX_BLOCKS = 16      # two streamers' worth (3 slots x {frames, coords} each) plus slack


def _attach(name):
    from multiprocessing import shared_memory
    b = _blocks.pop(name, None)
    if b is None:
        # (the parent owns the block: keep this process's resource tracker from unlinking it at exit)
        b = shared_memory.SharedMemory(name=name)
        try:
            from multiprocessing import resource_tracker
            resource_tracker.unregister(b._name, "shared_memory")
        except Exception:
            pass
        while len(_blocks) >= _MAX_BLOCKS:
            _detach([next(iter(_blocks))])
    _blocks[name] = b
    return b


def _detach(names):
    for name in names:
        b = _blocks.pop(name, None)
        if b is not None:
            try:
                b.close()
            except Exception:      # (a view still exported: the mapping goes with the next collection)
                pass


def decode_into(frames_shm, frames_shape, coords_shm, coords_shape, j, jpeg_path, npy_path):
    if jpeg_path is not None:
        from PIL import Image
        dst = np.ndarray(tuple(frames_shape), dtype=np.uint8, buffer=_attach(frames_shm).buf)
        with Image.open(jpeg_path) as im:
            dst[j] = np.asarray(im.convert("RGB"))
        del dst                     # (no exported view may outlive the request: `close()` of the mapping would refuse)
    if npy_path is not None:
        dst = np.ndarray(tuple(coords_shape), dtype=np.float32, buffer=_attach(coords_shm).buf)
        dst[j] = np.load(npy_path)
        del dst
    return j


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            req = json.loads(line)
            if isinstance(req, dict):
                _detach(req.get("detach", []))
            else:
                decode_into(*req)
            sys.stdout.write("ok\n")
        except Exception as e:      # the parent raises with this text
            sys.stdout.write("err " + repr(e).replace("\n", " ") + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
