"""Build libs2l_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libs2l_hip.so")
REF_LIB = os.path.join(PKG, "libs2l_hip_ref.so")      # the same ABI + the non-default kernel forms (-DS2L_WITH_REFERENCE_KERNELS): tests / tools only
RESOURCES = os.path.join(PKG, "kernel_resources.json")      # per-kernel registers / spills / scratch / LDS of the last build
SOURCES = ["pack.hip", "frontend.hip", "rows.hip", "render.hip", "ensemble.hip", "train.hip", "composite.hip", "unet.hip", "warp.hip", "syncnet.hip", "lpips.hip", "syncchain.hip", "train_bf16.hip", "quant.hip", "render16.hip", "convh.hip"]
# translation units that differ in the reference library: the split convolution's generated-assembly form (conv16.hip, its dispatch in
# unet.hip) and the four-wave / alternating-roles forms of the half-width convolution (convh.hip).  Every other object is shared.
REF_SOURCES = ["unet.hip", "convh.hip", "conv16.hip"]
# -ffp-contract=off: parity needs the reference's separate roundings (x*y then +z); FMAs are explicit fmaf()
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]
_REMARK = re.compile(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|"
                     r"Occupancy \[waves/SIMD\]):\s*(\S+)")


def _parse_resources(text: str) -> dict:
    """{kernel symbol: {vgprs, agprs, scratch, sgpr_spill, vgpr_spill, lds, occupancy}} from -Rpass-analysis=kernel-resource-usage.
    Several kernels wait with COUNTED vmcnt on requests the compiler does not know as memory operations (render, bf16 forward /
    backward bodies, conv3x3_split_kernel): a register spill to scratch would be an uncounted vector-memory operation inside their
    loops, so tests/test_abi_and_host.py checks this table for zero scratch."""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "SGPRs Spill": "sgpr_spill",
            "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds", "Occupancy [waves/SIMD]": "occupancy"}
    for m in _REMARK.finditer(text):
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None:
            cur[keys[k]] = int(v)
    return out


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libs2l_hip.so cannot be built (set HIPCC=...)")


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(REF_LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, "..", "include", "s2l_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip translation unit and link the shared library next to this file."""
    if not force and not needs_build() and os.path.exists(RESOURCES):
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    # the renderer's kernel body is generated assembly text (csrc/gen_render_body.py), included by render.hip
    sys.path.insert(0, CSRC)
    try:
        import gen_render_body
        gen_render_body.main_all(objdir)       # render_body.inc (long shape), render_body_wide.inc, render_body_single.inc
        import gen_render_fs_body
        gen_render_fs_body.main(os.path.join(objdir, "render_fs_body.inc"))      # the feature-split tile (one frame per call)
        import gen_rows_fs_body
        gen_rows_fs_body.main(os.path.join(objdir, "rows_fs_body.inc"))          # ... and the general-row MLP's (rgb_forward on a few thousand rows)
        import gen_render16_body
        gen_render16_body.main_all(objdir)     # render16_body_{long,wide,single}.inc: the split-bf16 speed mode (render16.hip)
        import gen_conv16_body
        gen_conv16_body.main(objdir)           # conv16_body.inc: the split-bf16 3x3 convolution of the U-Net's speed mode (conv16.hip)
        import gen_convh_body
        gen_convh_body.main(objdir)            # convh_body.inc: the 3x3 convolution on bf16 tensors of the half-width training chain (convh.hip)
        import gen_convh8_body
        gen_convh8_body.main(objdir)           # convh8_body.inc: its eight-wave form (two waves per SIMD)
        import gen_convhx_body
        gen_convhx_body.main(objdir)           # convhx_body.inc: the eight waves in alternating roles (reference library only; the default is convh8)
        import gen_conv_body               # ... and so are the U-Net's fp32 3x3 convolutions (unet.hip)
        gen_conv_body.main(objdir)
        import gen_fwd16_body              # ... and the bf16 training forward (train_bf16.hip)
        gen_fwd16_body.main(objdir)
        import gen_bwd16_body              # ... and its backward
        gen_bwd16_body.main(objdir)
    finally:
        sys.path.pop(0)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-I", objdir, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, False, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src in REF_SOURCES:      # the reference library's own objects, in the same pass
        obj = os.path.join(objdir, "ref_" + src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-DS2L_WITH_REFERENCE_KERNELS", "-I", objdir, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, True, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs, ref_objs, resources, ref_resources = [], [], {}, {}
    for src, obj, is_ref, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}{' (reference build)' if is_ref else ''}:\n{out}")
        (ref_resources if is_ref else resources).update(_parse_resources(out))
        rest = "\n".join(l for l in out.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l)
        if verbose and rest.strip():
            print(rest)
        (ref_objs if is_ref else objs).append(obj)
    ref_names = {s.replace(".hip", ".o") for s in REF_SOURCES}
    for out_lib, parts in ((LIB, objs), (REF_LIB, [o for o in objs if os.path.basename(o) not in ref_names] + ref_objs)):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_lib + ".tmp", *parts]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link of {os.path.basename(out_lib)} failed:\n{r.stdout}")
        os.replace(out_lib + ".tmp", out_lib)
    # per-kernel resources of the product library; the kernels only the reference library holds under "reference_only"
    resources["reference_only"] = {k: v for k, v in ref_resources.items() if k not in resources}
    with open(RESOURCES, "w") as f:
        json.dump(resources, f, indent=0, sort_keys=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
