#!/usr/bin/env python3
"""Golden G17: the composite's EDGE geometry, from the reference itself (tf_nerf.py:339-364).

`F.pad` with a negative amount crops, so a lip box that leaves the face frame is pasted with its outside part cut off; the
expanded rectangle is a python slice, so a negative start wraps around (and usually leaves an empty slice: no warped pixel at
all).  Runs only in the build container, like tools/make_goldens.py (whose import machinery it reuses); separate so that the
other fixtures are not regenerated.  Checks oracle == reference on every case before writing tests/golden/g17_composite_edges.npz.

    python tools/make_golden_edges.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens as MG  # noqa: E402
from oracle import s2l_oracle as O  # noqa: E402

# name, data path (pad mode / rectangle rule), lip h x w, (x0, y0)
CASES = [
    ("may_right_bottom", "dataset/may_face_crop_lip", 16, 24, 48, 52),        # box 48..72 x 52..68 in a 64x64 frame: cropped right + bottom
    ("may_left_top", "dataset/may_face_crop_lip", 16, 24, -5, -3),            # negative origin: cropped left + top; rectangle start wraps
    ("default_origin0", "dataset/someone_else", 16, 24, 0, 0),                # default mode pastes at (x0-1, y0-1) = (-1, -1)
    ("default_rect_wraps", "dataset/someone_else", 16, 24, 3, 30),            # box inside, but x0 - p = 3 - 4 < 0: the rectangle's column slice wraps -> empty
    ("obama2_rect_inside", "dataset/obama2_face_crop_lip", 16, 24, 3, 30),    # same origin with p = w // 12 = 2: the rectangle survives
    ("may_rect_clipped_far", "dataset/may_face_crop_lip", 16, 24, 38, 40),    # rectangle end beyond the frame: clipped by the slice, box inside
]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_config, _, TalkingFace, _, _, _ = MG.import_reference()
    rng = np.random.default_rng(4321)
    FH = FW = 64
    face = torch.from_numpy(rng.random((1, FH, FW, 3), dtype=np.float32))
    gt = torch.from_numpy(rng.random((1, FH, FW, 3), dtype=np.float32))
    m = torch.from_numpy((0.25 + 0.75 * rng.random((1, FH, FW, 1), dtype=np.float32))).expand(-1, -1, -1, 3).contiguous()
    ys, xs = torch.meshgrid(torch.arange(FH), torch.arange(FW), indexing="ij")
    ident = torch.stack([(2 * xs + 1) / FW - 1, (2 * ys + 1) / FH - 1], -1).float()
    coord = (ident * 1.04 + torch.tensor([0.01, -0.02]))[None] + torch.from_numpy(rng.standard_normal((1, FH, FW, 2)).astype(np.float32)) * 2e-3
    coord[:, :2] = ident[None, :2]
    coord = coord.contiguous()
    out = dict(face=face.numpy(), gt=gt.numpy(), mask=m.numpy(), coord=coord.numpy(), names=np.array([c[0] for c in CASES]))
    with torch.no_grad():
        for name, path, lh, lw, x0, y0 in CASES:
            lip = torch.from_numpy(rng.random((1, lh, lw, 3), dtype=np.float32))
            model, _ = MG.ref_model(ref_config, TalkingFace, lh, lw, data_path=path)
            _, new_ref, can_ref = model.post_fusion2_onlylip(lip, face, gt, m, x0, y0, coord)
            mode = O.PAD_MODE_MAY if ("may" in path or "obama2" in path) else O.PAD_MODE_DEFAULT
            new_o, can_o = O.composite(lip, face, gt, m, x0, y0, coord, pad_mode=mode, pad_div=12 if "obama2" in path else 5)
            e_new, e_can = MG.maxerr(new_ref, new_o), MG.maxerr(can_ref, can_o)
            shown = float((new_ref != gt).any(-1).float().mean())
            print(f"  {name:22s} oracle vs reference: new {e_new:.1e} canonical {e_can:.1e}; warped pixels shown: {shown:.3f}")
            assert e_new == 0.0 and e_can == 0.0, name
            out.update({f"{name}/lip": lip.numpy(), f"{name}/x0": np.array(x0), f"{name}/y0": np.array(y0), f"{name}/path": np.array(path),
                        f"{name}/merged_new": new_ref.numpy(), f"{name}/merged_canonical": can_ref.numpy(), f"{name}/shown": np.array(shown)})
        # a box ENTIRELY outside the frame: F.pad raises in the reference
        model, _ = MG.ref_model(ref_config, TalkingFace, 16, 24, data_path="dataset/may_face_crop_lip")
        try:
            model.post_fusion2_onlylip(torch.zeros(1, 16, 24, 3), face, gt, m, 70, 10, coord)
            raised = False
        except RuntimeError:
            raised = True
        print("  box entirely outside -> the reference raises:", raised)
        out["outside_raises"] = np.array(raised)
    np.savez_compressed(os.path.join(MG.GOLD, "g17_composite_edges.npz"), **out)
    print("wrote g17_composite_edges.npz")


if __name__ == "__main__":
    main()
