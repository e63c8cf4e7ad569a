"""CPU: the C-ABI library loads and exports every symbol include/s2l_hip.h declares; host-side
logic (config, module surface, loud failure without a GPU).  No compute calls."""
import ctypes
import os
import re

import numpy as np

import pytest
import torch

import speech2lip_amd as s2l
from speech2lip_amd import _abi, build, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _abi.load()


def test_header_symbols_are_all_exported(lib):
    header = open(os.path.join(ROOT, "include", "s2l_hip.h")).read()
    declared = set(re.findall(r"\b(s2l_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    raw = ctypes.CDLL(build.LIB)
    for name in declared:
        assert hasattr(raw, name), name
    ref = ctypes.CDLL(build.REF_LIB)      # the test-side build with the non-default kernel forms: the same ABI
    for name in declared:
        assert hasattr(ref, name), name
    assert lib.s2l_version().decode().startswith("s2l_hip") and b"gfx950" in lib.s2l_version()
    assert lib.s2l_packed_floats() > W.HOT_PATH_PARAM_COUNT  # blob holds every weight at least once


def test_product_never_touches_the_reference_library_or_the_oracle():
    """libs2l_hip_ref.so (the non-default kernel forms, -DS2L_WITH_REFERENCE_KERNELS) and oracle/ are test infrastructure: no module of
    the product package loads, names or imports either -- `_abi.load_reference` / `reference_kernels` are DEFINED in _abi.py and
    called from tests/ and tools/ only."""
    pkg = os.path.join(ROOT, "speech2lip_amd")
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        text = open(os.path.join(pkg, fn)).read()
        assert "oracle" not in text.replace("oracle's", "").replace("the oracle", "") or fn in ("build.py",), fn
        if fn not in ("_abi.py", "build.py"):
            assert "load_reference" not in text and "reference_kernels" not in text and "libs2l_hip_ref" not in text, fn
    abi = open(os.path.join(pkg, "_abi.py")).read()
    body = abi[abi.index("def load()"):abi.index("_ref_lib = None")]
    assert "ref" not in body.lower().replace("prefer", "")      # load() itself knows nothing of the second library


def test_tensor_order_matches_header_enum():
    header = open(os.path.join(ROOT, "include", "s2l_hip.h")).read()
    body = header[header.index("S2L_T_CONV0_W"):header.index("S2L_NUM_TENSORS")]
    enum = re.findall(r"S2L_T_([A-Z0-9_]+)", body)
    assert len(enum) == len(_abi.TENSOR_ORDER) == 42
    canon = lambda k: k.replace("encoder_conv.", "conv").replace("encoder_fc1.", "fc1_").replace("pts_linears.", "pts") \
        .replace("output_linear", "out").replace(".weight", "_w").replace(".bias", "_b").upper()
    assert [canon(k) for k in _abi.TENSOR_ORDER] == enum
    assert [n for n, _ in W.HOT_PATH_TENSORS] == _abi.TENSOR_ORDER


def test_argument_errors_do_not_need_a_gpu(lib):
    null = ctypes.c_void_p(0)
    assert lib.s2l_audio_encode(null, null, null, 4, null) == -1           # S2L_E_NULL
    assert lib.s2l_render_lip(null, null, null, null, null, null, 16, 1, null) == -1
    one = ctypes.c_void_p(16)
    assert lib.s2l_audio_encode(one, one, one, -1, null) == -2              # S2L_E_SIZE
    assert lib.s2l_audio_encode(null, null, null, 0, null) == 0             # empty batch: pointers may be null
    assert lib.s2l_audio_encode(one, one, one, 0, null) == 0                # empty batch is a no-op
    assert lib.s2l_rgb_forward(one, one, 0, one, one, 0, null) == 0
    assert lib.s2l_render_lip(one, one, one, one, one, one, 16, 0, null) == 0
    odd = ctypes.c_void_p(20)
    assert lib.s2l_render_lip(one, odd, one, one, one, one, 16, 1, null) == -3   # S2L_E_ALIGN
    # composite geometry: a lip box ENTIRELY outside the face frame is an error, as in the reference (F.pad raises; a box that
    # only leaves the frame partly is cropped, golden G17 -- those calls go on to launch and need a GPU)
    assert lib.s2l_composite(one, one, 0, one, 0, one, one, one, null, null, 16, 24, 64, 64, 70, 30, 0, 4, 1, null) == -4
    assert lib.s2l_composite(one, one, 0, one, 0, one, one, one, null, null, 16, 24, 64, 64, -25, 30, 0, 4, 1, null) == -4
    assert lib.s2l_composite(one, one, 0, one, 0, one, one, one, null, null, 16, 24, 64, 64, 20, 65, 0, 4, 1, null) == -4
    assert lib.s2l_composite(one, one, 0, one, 0, one, one, one, null, null, 16, 24, 64, 64, 20, 64, 0, 4, 1, null) != -4    # touching from outside: cropped to nothing
    assert lib.s2l_composite(one, one, 0, one, 0, one, one, one, null, null, 16, 24, 64, 64, 0, 0, 1, 4, 1, null) != -4     # default mode: origin (-1, -1), cropped
    assert lib.s2l_composite(one, one, 7, one, 0, one, one, one, null, null, 16, 24, 64, 64, 20, 30, 0, 4, 1, null) == -2
    # round-3 entry points: the pair pass of the lip-sync expert, the frozen / bf16 train-mode U-Net, the kernel selectors
    assert lib.s2l_syncnet_forward_pair(one, one, one, one, one, one, 5, 4, null) == -2          # more mel windows than face windows
    assert lib.s2l_syncnet_forward_pair(one, one, one, one, one, one, 0, 0, null) == 0
    assert lib.s2l_syncnet_forward_pair(one, null, one, one, null, one, 2, 4, null) == -1        # mel / audio_emb needed when audio_batch > 0
    assert lib.s2l_syncnet_face_backward_prefix(one, one, one, one, one, 5, 4, null) == -2       # more gradients than forwarded windows
    assert lib.s2l_syncnet_face_backward_prefix(one, one, one, one, one, 0, 4, null) == 0
    tbl = (ctypes.c_void_p * 52)(*([16] * 52))
    assert lib.s2l_unet_train_backward(one, tbl, one, one, one, one, null, null, 8, 8, 1, null) == -1       # neither d_x nor grads
    assert lib.s2l_unet_train_backward_bf16(one, null, tbl, one, one, one, one, one, one, 8, 8, 1, null) == -1   # no bf16 blob
    assert lib.s2l_unet_train_forward_bf16(one, null, tbl, 1e-5, 0.1, 0, one, one, one, one, 8, 8, 1, null) == -1
    assert lib.s2l_unet_train_forward_bf16(one, odd, tbl, 1e-5, 0.1, 0, one, one, one, one, 8, 8, 1, null) == -3
    assert lib.s2l_unet_train_forward(one, tbl, 1e-5, 0.1, 0, one, one, one, one, 3, 8, 1, null) == -2             # H < 4
    assert lib.s2l_set_unet_split_kernel(3) == -2 and lib.s2l_set_unet_split_kernel(0) == 0
    assert lib.s2l_set_unet_split_kernel(2) == -5 and lib.s2l_set_unet_split_kernel(1) == 0 and lib.s2l_set_unet_split_kernel(0) == 0   # form 2: reference library only
    # round-4 entries: the half-width chains (bf16 tensors between the kernels), their selector, one layer on its own
    assert lib.s2l_unet_train_forward_frames_h(one, null, tbl, 1e-5, 0.1, 0, one, one, one, one, 8, 8, 1, null) == -1        # the bf16 blob is required
    assert lib.s2l_unet_train_forward_frames_h(one, one, tbl, 1e-5, 0.1, 0, one, one, one, one, 8, 8, 9000, null) == -2      # frames x planes > 65535
    assert lib.s2l_unet_train_forward_frames_h(one, odd, tbl, 1e-5, 0.1, 0, one, one, one, one, 8, 8, 1, null) == -3
    assert lib.s2l_unet_train_backward_frames_h(one, one, tbl, one, one, one, null, 8, 8, 1, null) == -1                     # the input gradient is the output
    assert lib.s2l_unet_train_backward_frames_h_grads(one, one, tbl, one, one, one, one, one, null, 8, 8, 1, null) == -1     # ... here the parameter gradients
    assert lib.s2l_unet_train_backward_frames_h_grads(one, one, tbl, null, one, one, one, one, one, 8, 8, 1, null) == -1     # x is needed for the first layer's dW
    assert lib.s2l_unet_train_backward_frames_grads(one, null, tbl, one, one, one, one, one, null, 8, 8, 1, null) == -1
    assert lib.s2l_unet_forward_saved_h(one, one, one, one, one, 8, 8, 8, 8, 2, 0, 1, null) == -4                            # window origin not a multiple of 4
    assert lib.s2l_unet_forward_saved_h(one, null, one, one, one, 8, 8, 8, 8, 0, 0, 1, null) == -1
    assert lib.s2l_unet_forward_saved_h(one, one, one, one, one, 8, 8, 8, 8, 0, 0, 0, null) == 0                             # no frames: nothing to do
    assert lib.s2l_unet_backward_h(one, one, one, one, one, null, 8, 8, 8, 8, 0, 0, 1, null) == -1
    assert lib.s2l_convh_layer(one, 0, 0, one, 64, null, 0, null, one, 8, 8, 1, null) == -2                                  # layer 0 is the fp32-input convolution
    assert lib.s2l_convh_layer(one, 1, 0, one, 32, null, 0, null, one, 8, 8, 1, null) == -2                                  # channel count of the layer
    assert lib.s2l_set_unet_half_kernel(3) == -2 and lib.s2l_set_unet_half_kernel(0) == 0
    assert lib.s2l_set_unet_half_kernel(1) == -5 and lib.s2l_set_unet_half_kernel(2) == -5                                  # S2L_E_UNSUPPORTED in the product library
    assert lib.s2l_unet_train_frames_h_saved_halves(8, 8, 0) == 0 and lib.s2l_unet_train_frames_h_saved_halves(500, 500, 1) > 2 * 10 ** 8
    assert lib.s2l_unet_saved_h_halves(500, 500, 1) == lib.s2l_unet_saved_floats(500, 500, 1)
    # F one-frame calls in one set of launches (every frame its own statistics group)
    assert lib.s2l_unet_train_backward_frames(one, null, tbl, one, one, one, one, null, 8, 8, 2, null) == -1     # input gradient is the output
    assert lib.s2l_unet_train_forward_frames(one, null, tbl, 1e-5, 0.1, 0, one, one, one, one, 8, 8, 0, null) == -2
    assert lib.s2l_unet_train_frames_saved_floats(8, 8, 3) == lib.s2l_unet_train_saved_floats(8, 8, 3) + 2 * 10 * 512
    assert lib.s2l_unet_train_frames_scratch_floats(3) == 3 * 262144 and lib.s2l_unet_train_frames_scratch_floats(0) == 0
    assert lib.s2l_unet_train_frames_work_floats(8, 8, 1) == lib.s2l_unet_train_work_floats(8, 8, 1)


def test_config_inherit_and_merge(tmp_path):
    (tmp_path / "base.yaml").write_text("model:\n  a: 1\n  b: {x: 1, y: 2}\ndata:\n  path: p\n")
    (tmp_path / "mid.yaml").write_text("inherit_from: base.yaml\nmodel:\n  b: {y: 3}\n  c: 4\n")
    (tmp_path / "leaf.yaml").write_text("inherit_from: mid.yaml\nmodel:\n  a: 9\ntraining:\n  k: [1, 2]\n")
    (tmp_path / "default.yaml").write_text("zzz: 1\n")
    cfg = s2l.load_config("leaf.yaml", "default.yaml", abs_path=str(tmp_path))
    assert cfg["model"] == {"a": 9, "b": {"x": 1, "y": 3}, "c": 4}
    assert cfg["data"]["path"] == "p" and cfg["training"]["k"] == [1, 2] and cfg["zzz"] == 1
    solo = s2l.load_config(str(tmp_path / "default.yaml"))
    assert solo == {"zzz": 1}


def test_may_config_surface():
    cfg = s2l.may_config(64, 48)
    assert cfg["data"]["height"] == 64 and cfg["data"]["width"] == 48 and cfg["training"]["batch_rays"] == 64 * 48
    assert cfg["model"]["MLP_version"] == "v2" and "may" in cfg["data"]["path"]


def test_module_surface_and_state_dict_keys():
    m = s2l.TalkingFace(torch.device("cpu"), s2l.may_config(16, 16), mode="eval")
    keys = set(m.state_dict().keys())
    assert set(n for n, _ in W.HOT_PATH_TENSORS) <= keys
    assert set(n for n, _ in W.DEAD_TENSORS) <= keys
    assert m.audio_dims == 64 and m.uv_embedder.out_dims == 42 and m.time_embedder_new.out_dims == 20
    sd = {k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()}
    sd.update({k: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})   # post_fusion_unet.* (§8f-1)
    sd["canonical_depth_head"] = torch.zeros(500, 500)                                    # out-of-path keys are ignored
    res = m.load_state_dict(sd)
    assert not res.missing_keys and res.unexpected_keys == ["canonical_depth_head"]
    assert set(W.make_unet_state_dict(0)) <= keys
    assert torch.equal(m.fc_uv.weight.detach(), sd["fc_uv.weight"])


def test_unsupported_flags_raise():
    cfg = s2l.may_config()
    cfg["model"]["use_head_pose"] = True
    with pytest.raises(NotImplementedError):
        s2l.TalkingFace(torch.device("cpu"), cfg)
    cfg = s2l.may_config()
    cfg["model"]["MLP_version"] = "v1"
    with pytest.raises(NotImplementedError):
        s2l.TalkingFace(torch.device("cpu"), cfg)


def test_no_cpu_fallback():
    """The product path refuses to run without the GPU instead of silently computing on CPU."""
    m = s2l.TalkingFace(torch.device("cpu"), s2l.may_config(16, 16), mode="eval")
    with pytest.raises(_abi.S2LError):
        m.audio_merge_forward(torch.zeros(2, 16, 29))
    with pytest.raises(_abi.S2LError):
        m.rgb_forward(torch.zeros(4, 66), time_pts=torch.tensor([0]))
    with pytest.raises(_abi.S2LError):
        m.render_clip(torch.zeros(1, 16, 29), [0], 16, 16)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "speech2lip_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "s2l_oracle" not in text, f


def test_get_coords_matches_reference_grid(golden):
    g = golden("g1_embed.npz")
    for key in g:
        if key.startswith("coords_"):
            w, h = map(int, key[len("coords_"):].split("x"))
            assert torch.equal(s2l.get_coords(w, h, torch.device("cpu")), torch.from_numpy(g[key])), key


def test_bench_refuses_to_report_fewer_gpus_than_requested():
    """`bench.py --gpus N` outside torchrun spawns its own ranks; with fewer visible GPUs than N it must fail loudly instead
    of printing a line whose n_gpus differs from --gpus (this container has no GPU at all)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 2 requested but only" in (r.stderr + r.stdout)
    assert "n_gpus" not in r.stdout
    # under a launcher whose world size disagrees with --gpus the line is refused as well
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_synthetic_generators_are_deterministic():
    a, b = W.synthetic_warp_coords(2, 20, 24, seed=4), W.synthetic_warp_coords(2, 20, 24, seed=4)
    assert a.shape == (2, 20, 24, 2) and a.dtype == "float32" and (a == b).all() and abs(a).max() <= 1.0
    assert not (W.synthetic_warp_coords(2, 20, 24, seed=5) == a).all()
    # near the identity grid of grid_sample(align_corners=False): centre of pixel (y,x) -> ((2x+1)/W-1, (2y+1)/H-1)
    assert abs(a[0, 10, 12, 0] - (25 / 24 - 1)) < 0.08 and abs(a[0, 10, 12, 1] - (21 / 20 - 1)) < 0.08
    img = W.synthetic_image((3, 4, 3), 2)
    assert img.min() >= 0 and img.max() < 1 and (img == W.synthetic_image((3, 4, 3), 2)).all()


def test_canonical_depth_head_parameter(tmp_path):
    """tf_nerf.py:174-195: the parameter exists under the reference's state-dict key when model.use_canonical_depth is on, built
    from the init depth + head mask the reference reads (holes filled with the mean depth inside the mask, zero outside)."""
    from PIL import Image
    cfg = s2l.may_config(16, 16, str(tmp_path / "may_face"))
    (tmp_path / "may_face").mkdir()
    init = np.zeros((12, 10), np.float32)
    init[2:9, 3:8] = 9.0 + np.arange(35, dtype=np.float32).reshape(7, 5) * 0.01
    init[4, 5] = 0.0                                                     # a hole inside the head
    np.save(tmp_path / "depth.npy", init)
    mask = np.zeros((12, 10, 3), np.uint8)
    mask[1:10, 2:9] = 255
    Image.fromarray(mask).save(tmp_path / "may_face" / "canonical_head_mask.png")
    (tmp_path / "may_face" / "canonical_head_mask.png").rename(tmp_path / "may_face" / "canonical_head_mask.jpg")
    cfg["model"].update(use_canonical_depth=True, canonical_depth_init_path=str(tmp_path / "depth.npy"), canonical_depth_height=12,
                        canonical_depth_width=10)
    m = s2l.TalkingFace(torch.device("cpu"), cfg)
    d = m.canonical_depth_head.detach()
    assert "canonical_depth_head" in m.state_dict() and d.shape == (12, 10) and m.canonical_depth_head.requires_grad
    mean = float(init[init > 0].mean())
    assert abs(float(d[4, 5]) - mean) < 1e-5 and float(d[0, 0]) == 0.0 and abs(float(d[1, 2]) - mean) < 1e-5
    assert torch.equal(d[2:9, 3:8][init[2:9, 3:8] > 0], torch.from_numpy(init[2:9, 3:8][init[2:9, 3:8] > 0]))
    cfg["model"].pop("canonical_depth_init_path")
    assert s2l.TalkingFace(torch.device("cpu"), cfg).canonical_depth_head.shape == (12, 10)


def test_render_body_generator_is_deterministic_and_complete(tmp_path):
    """csrc/gen_render_body.py writes the renderer's assembly body at build time.  Its own assertions check the LDS wait
    bookkeeping at every loop edge; here: two runs give the same text, and the text holds exactly the MFMAs of the schedule --
    16 slabs x 192 in the layer loop, 192 in the output layer, 12 in each of the four out-of-line table refills."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_render_body", os.path.join(ROOT, "speech2lip_amd", "csrc", "gen_render_body.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    a, b = str(tmp_path / "a.inc"), str(tmp_path / "b.inc")
    gen.main(a)
    gen.main(b)
    text = open(a).read()
    assert text == open(b).read()
    assert text.count("v_mfma_f32_16x16x4_f32") == 16 * 192 + 192 + 4 * 12
    assert text.count("s_barrier") == 1 + 16 + 2 + 2 + 1          # prime, slabs, q0/p0, q5/p5, output layer
    assert text.count("global_load_lds_dwordx4") >= 9 * 4


def test_conv_body_generator_is_deterministic_and_complete(tmp_path):
    """csrc/gen_conv_body.py writes the U-Net 3x3 convolution's assembly bodies.  Two runs give the same text; each variant holds
    the chunk's 9 taps x 64 MFMAs four times (first chunk of a tile, later even chunks, odd chunks with / without a fetch), 15
    LDS-DMA instructions per fetching copy plus the prime, one barrier per chunk copy plus the prime, and the tile's 16 (+ 8
    pooled) stores twice: behind the next tile's first chunk, and after the loop for the last tile.  The variants differ in
    their tile ends only: pooled copy, fused 1x1 output convolution, or no bias / ReLU with an optional gate."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_conv_body", os.path.join(ROOT, "speech2lip_amd", "csrc", "gen_conv_body.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(), b.mkdir()
    gen.main(str(a))
    gen.main(str(b))
    for variant in ("fwd", "fwd_pool", "fwd_out", "lin"):
        text = open(a / f"conv_body_{variant}.inc").read()
        assert text == open(b / f"conv_body_{variant}.inc").read()
        assert text.count("v_mfma_f32_16x16x4_f32") == 4 * 9 * 64
        assert text.count("global_load_lds_dwordx4") == 4 * (6 + 9) + (variant == "fwd_out")      # + the output weights
        assert text.count("s_barrier") == 1 + 4
        assert text.count("global_store_dwordx4") == 2 * (16 + (8 if variant == "fwd_pool" else 0))
        assert text.count("global_store_dwordx3") == (4 if variant == "fwd_out" else 0)              # 12-byte output pixels, per tile row
        assert text.count("v_fma_f32") == (3 * 4 * 16 if variant == "fwd_out" else 0)                 # 64 -> 3 on 4 rows, 16 channels per lane
        assert text.count("global_load_dwordx4 v[") == (2 * 16 if variant == "lin" else 0)           # gate quads, behind both last-chunk copies
        assert text.count("global_load_dwordx4 a[") == (0 if variant == "lin" else 2 * 4)            # bias quads (prime + per tile)
        body = text.split("asm volatile")[1].split(": [karg]")[0]
        assert "s32" not in body and "s33" not in body           # s32 / s33 stay the compiler's
        assert '"v"(' not in text                                # every VGPR is the body's: no vector operand


def test_fwd16_body_generator_is_deterministic_and_complete(tmp_path):
    """csrc/gen_fwd16_body.py writes the assembly form of the bf16 training forward (64 rows per wave).  Two runs give the same
    text; it holds the MFMAs of its four layer bodies -- layer 0: 4 stages x 32, layers of kind B / C: 4 x 64 each, layer 5:
    4 x 96 -- plus the output layer's 32, one barrier per stage (+ the prime, + one behind the output layer), the stage DMA of the parts the next stage needs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_fwd16_body", os.path.join(ROOT, "speech2lip_amd", "csrc", "gen_fwd16_body.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(), b.mkdir()
    gen.main(str(a))
    gen.main(str(b))
    text = open(a / "fwd16_body.inc").read()
    assert text == open(b / "fwd16_body.inc").read()
    assert text.count("v_mfma_f32_32x32x16_bf16") == 4 * 32 + 2 * 4 * 64 + 4 * 96 + 32
    assert text.count("s_barrier") == 16 + 1 + 1                      # per stage, the prime, after the output layer
    # stage DMA: only the parts the next stage needs (x: 4 instructions, h: 8) -- layer 0: 3 x 4 + 8, layer 5: 3 x 12 + 8,
    # kind C: 3 x 8 + (4 + 8), kind B: 2 x 8 + (4 + 8) + (4 + 8); + the prime's 4
    assert text.count("global_load_lds_dwordx4") == 20 + 44 + 36 + 40 + 4
    assert text.count("global_store_dwordx4") == 16 * 8 + 8            # a stage's images behind the next stage's MFMAs, + the last
    assert '"v"(' not in text


def test_lpips_module_has_the_package_state_dict_and_oracle_properties():
    """speech2lip_amd.LPIPS carries the state-dict keys of lpips.LPIPS(net='alex') (lpips==0.1.4), frozen; the oracle's
    restatement is zero on identical images, symmetric, and positive otherwise."""
    import speech2lip_amd as s2l
    from oracle import s2l_oracle as O
    from speech2lip_amd import weights as W
    m = s2l.LPIPS(net="alex", version="0.1")
    want = {"scaling_layer.shift", "scaling_layer.scale"}
    for name in ("net.slice1.0", "net.slice2.3", "net.slice3.6", "net.slice4.8", "net.slice5.10"):
        want |= {name + ".weight", name + ".bias"}
    for i in range(5):
        want |= {f"lin{i}.model.1.weight", f"lins.{i}.model.1.weight"}
    assert set(m.state_dict().keys()) == want
    assert tuple(m.state_dict()["lin2.model.1.weight"].shape) == (1, 384, 1, 1)
    assert tuple(m.state_dict()["net.slice1.0.weight"].shape) == (64, 3, 11, 11)
    assert not any(p.requires_grad for p in m.parameters()) and not m.training
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_lpips_state_dict(0).items()}, strict=True)
    with pytest.raises(Exception, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))
    sd = O.to_sd(W.make_lpips_state_dict(0))
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(2, 3, 64, 80, generator=g) * 2 - 1, torch.rand(2, 3, 64, 80, generator=g) * 2 - 1
    dab, dba = O.lpips_alex(sd, a, b), O.lpips_alex(sd, b, a)
    assert tuple(dab.shape) == (2, 1, 1, 1) and float(dab.min()) > 0
    assert torch.allclose(dab, dba, rtol=1e-6) and float(O.lpips_alex(sd, a, a).abs().max()) == 0.0


def test_lpips_pretrained_never_runs_on_random_weights(tmp_path):
    """LPIPS(pretrained=True) (training.py:76) must not train silently against random features: the package would load
    torchvision's AlexNet + its own linear heads; here a missing part warns at construction, records it in weights_loaded and
    refuses to run, and load_trunk / load_lins accept the torchvision / package key layouts with strict key checks."""
    import warnings
    import speech2lip_amd as s2l
    from speech2lip_amd import _abi, weights as W
    with pytest.warns(RuntimeWarning, match="RANDOM"):
        m = s2l.LPIPS(net="alex", version="0.1", model_path="models/lpips_weights_v0.1/alex.pth")      # the reference's call
    assert m.weights_loaded == {"trunk": False, "lins": False}
    with pytest.raises(_abi.S2LError, match="random features"):
        m._require_weights()
    sd = {k: torch.from_numpy(v) for k, v in W.make_lpips_state_dict(0).items()}
    tv = {f"features.{k.split('.')[2]}.{k.split('.')[3]}": v for k, v in sd.items() if k.startswith("net.")}      # torchvision names
    assert set(tv) == {f"features.{i}.{p}" for i in (0, 3, 6, 8, 10) for p in ("weight", "bias")}
    heads = {k: v for k, v in sd.items() if k.startswith("lin") and not k.startswith("lins.")}
    torch.save(tv, tmp_path / "alexnet.pth")
    torch.save(heads, tmp_path / "alex.pth")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m2 = s2l.LPIPS(net="alex", version="0.1", model_path=str(tmp_path / "alex.pth"), trunk_path=str(tmp_path / "alexnet.pth"))
    assert m2.weights_loaded == {"trunk": True, "lins": True}
    m2._require_weights()
    for k, v in sd.items():
        assert torch.equal(m2.state_dict()[k], v), k
    with pytest.raises(KeyError):
        m.load_trunk({k: v for k, v in tv.items() if not k.startswith("features.8")})
    with pytest.raises(KeyError):
        m.load_lins({**heads, "lin5.model.1.weight": heads["lin0.model.1.weight"]})
    m.load_trunk(tv)
    assert m.weights_loaded == {"trunk": True, "lins": False}
    m.load_lins(heads)
    m._require_weights()
    s2l.LPIPS(pretrained=False)._require_weights()            # explicit random init (pnet_rand-style structural use) is allowed


def test_config_and_trainer_read_the_reference_keys():
    """lambda_rgb lives under cfg['model'] (src/face_simple/config.py:41, may.yaml:11); may_config(train_flags=True) carries
    may.yaml's loss switches (:43-54)."""
    import speech2lip_amd as s2l
    cfg = s2l.may_config(96, 96, train_flags=True)
    assert cfg["model"]["lambda_rgb"] == 1.0 and cfg["model"]["use_canonical_depth"] is True
    t = cfg["training"]
    assert t["use_perceptual_loss"] and t["use_syncloss"] and t["use_canonical_depth_loss_photo_v2"] and t["stage"] == "stage1"
    assert (t["w_post_fusion"], t["w_perceptual_loss"], t["w_syncloss"]) == (1.0, 0.01, 0.01)
    cfg0 = s2l.may_config(96, 96)
    assert not cfg0["training"]["use_perceptual_loss"] and not cfg0["training"]["use_syncloss"]
    from speech2lip_amd.training import Trainer

    class _M:      # Trainer only reads these from the model at construction
        audio_dims, device = 64, "cpu"
    cfg0["model"]["lambda_rgb"] = 0.25
    cfg0["training"]["lambda_rgb"] = 9.0
    _M.cfg = cfg0
    assert Trainer(_M()).w_photometric_loss == 0.25
    del cfg0["model"]["lambda_rgb"]
    assert Trainer(_M()).w_photometric_loss == 9.0
    assert hasattr(Trainer, "train_step")


def test_sync_chain_unet_window_geometry():
    """SyncChain.unet_window: the canonical-face box dilated by the U-Net's dependency radius, origins on the 4-pixel grid of the
    two pooling levels, sizes multiples of 4 unless the crop ends at the frame edge (what s2l_unet_forward_saved_window accepts)."""
    import speech2lip_amd as s2l
    ch = s2l.SyncChain.__new__(s2l.SyncChain)
    ch.window = True
    r = s2l.SyncChain.UNET_RADIUS
    assert r >= 32
    for bbox, (FH, FW) in (([110, 90, 390, 420, 1.0], (500, 500)), ([0, 0, 500, 500, 1.0], (500, 500)), ([3, 5, 97, 121, 0.9], (130, 101)),
                           ([200, 260, 500, 500, 1.0], (500, 500)), ([37, 41, 77, 83, 1.0], (501, 503))):
        x0, y0, x1, y1 = ch.unet_window(bbox, FH, FW)
        assert 0 <= x0 <= max(0, bbox[0] - r) and 0 <= y0 <= max(0, bbox[1] - r)
        assert min(FW, bbox[2] + r) <= x1 <= FW and min(FH, bbox[3] + r) <= y1 <= FH
        assert x0 % 4 == 0 and y0 % 4 == 0
        assert (x1 - x0) % 4 == 0 or x1 == FW
        assert (y1 - y0) % 4 == 0 or y1 == FH
        assert x0 >= bbox[0] - r - 3 and y0 >= bbox[1] - r - 3          # no larger than needed
    ch.window = False
    assert ch.unet_window([110, 90, 390, 420, 1.0], 500, 500) == (0, 0, 500, 500)


def test_no_kernel_spills_to_scratch():
    """Several kernels wait with COUNTED vmcnt on requests issued as assembly text (the renderer, the bf16 forward / backward bodies,
    conv3x3_split_kernel): a VGPR spilled to scratch would be a vector-memory operation the count does not know, inside their loops.
    The build records hipcc's per-kernel resource table; no kernel of the library may use scratch or spill a VGPR."""
    import json
    from speech2lip_amd import build
    build.build_library()
    if not os.path.exists(build.RESOURCES):
        build.build_library(force=True)
    table = json.load(open(build.RESOURCES))
    assert len(table) >= 100
    table = {**{k: v for k, v in table.items() if k != "reference_only"}, **table.get("reference_only", {})}      # both libraries' kernels
    bad = {k: v for k, v in table.items() if v.get("scratch", 0) != 0 or v.get("vgpr_spill", 0) != 0}
    assert not bad, bad
    counted = [k for k in table if any(s in k for s in ("conv3x3_split_kernel", "render_tiles_kernel", "fwd_asm_bf16_kernel",
                                                        "bwd_asm_bf16_kernel", "conv3x3_asm_kernel"))]
    assert len(counted) >= 4 + 3 + 1 + 1 + 4, counted
    for k in counted:
        assert table[k]["vgprs"] + table[k].get("agprs", 0) <= 512
    # the 8-wave half-width conv kernels (csrc/convh.hip) are sized for two waves per SIMD: 512-thread workgroups only fit a CU's
    # register file at <= 128 VGPR + 128 AGPR, and the launch would fail (not slow down) beyond that
    half = [k for k in table if "convh8_asm_kernel" in k or "convh8_relu_asm_kernel" in k]
    assert len(half) == 2, half
    for k in half:
        assert table[k]["occupancy"] >= 2 and table[k]["vgprs"] <= 128 and table[k].get("agprs", 0) <= 128, (k, table[k])


def test_staging_registers_of_the_persistent_conv_kernel_are_never_copied(tmp_path):
    """conv3x3_split_kernel requests halo values with assembly-text loads into tied operands and retires them with a hand-counted
    vmcnt (csrc/unet.hip, fetch_one: three rules).  If the register allocator ever copies one of those registers (v_mov /
    v_accvgpr) the copy may be taken before the data has landed -- a rare, timing-dependent corruption that an earlier form of the
    kernel had (tools/soak_conv_kernels.py found it).  Check the compiler's own assembly: in every instantiation, no move reads a
    register that an assembly-text load writes, and all instantiations stage in registers only (no scratch)."""
    import shutil
    import subprocess
    from speech2lip_amd import build
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    build.build_library()                                      # (generates the .inc files unet.hip includes)
    out = tmp_path / "unet.s"
    flags = [f for f in build.FLAGS if not f.startswith("-Rpass")]
    subprocess.run([hipcc, *flags, "-I", os.path.join(build.PKG, "build"), "-S", "--cuda-device-only",
                    os.path.join(build.CSRC, "unet.hip"), "-o", str(out)], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    txt = out.read_text().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^_ZN3s2l20conv3x3_split_kernel\w+:", l)]
    assert len(starts) == 4
    for st in starts:
        end = next(i for i in range(st, len(txt)) if "s_endpgm" in txt[i])
        body = txt[st:end]
        regs, inasm = set(), False
        for l in body:
            if "ASMSTART" in l:
                inasm = True
            elif "ASMEND" in l:
                inasm = False
            elif inasm:
                m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
                if m:
                    regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        assert len(regs) == 40, (txt[st], sorted(regs))       # two staging sets of five 16-byte registers, the same ones at every site
        for l in body:
            l2 = l.strip()
            ops = l2.split(None, 1)
            if len(ops) < 2 or not ops[0].startswith(("v_mov", "v_accvgpr", "v_swap")):
                continue
            used = set()
            for src in ops[1].split(",")[1:]:
                for m in re.finditer(r"v\[(\d+):(\d+)\]", src):
                    used.update(range(int(m.group(1)), int(m.group(2)) + 1))
                for m in re.finditer(r"\bv(\d+)\b", src):
                    used.add(int(m.group(1)))
            assert not (used & regs), (txt[st], l2)
        assert not any("scratch_" in l for l in body)


def _reference_style_config(h=80, w=120):
    """A config dictionary with every key the reference's factory indexes (src/face_simple/config.py:33-73) -- the merged form
    of its default.yaml + face_simple default.yaml + may.yaml, typed here key by key (values: may.yaml where it sets them, the
    defaults otherwise)."""
    import speech2lip_amd as s2l
    cfg = s2l.may_config(h, w, train_flags=True)
    cfg["model"].update(use_canonical_depth=False, post_fusion_warping="backward")
    cfg["training"].update(
        lindisp=False, perturb=True, raw_noise_std=1, n_sample_points_fine=64, local_rank=0, use_canonical_loss=False,
        use_temp_consist=False, use_loss_bg=False, use_loss_face=False, use_loss_facewoaudio=False, use_loss_lip=False,
        use_c_lip=False, use_perceptual_loss_mask=False, use_low_resolution=False, use_perceptual_loss=False, use_syncloss=False,
        out_dir="log/face_simple/may")
    cfg["test"] = {"threshold": 0.5}
    return cfg


def test_factories_build_model_and_trainer_like_the_reference():
    """src/config.py:67-95 + src/face_simple/config.py:13-94: `get_model(cfg, device, len_dataset, config)` and
    `get_trainer(model, optimizer, cfg, device)` through `method_dict['face_simple'].config`, and the reference's POSITIONAL
    constructor order `Trainer(model, optimizer, device, out_dir, cfg=...)` (training.py:21-23)."""
    import speech2lip_amd as s2l
    from speech2lip_amd import config as C
    cfg = _reference_style_config()
    assert cfg["training"]["use_sync_contrastive_loss"] is True            # may.yaml:47
    assert s2l.may_config(96, 96)["training"]["use_sync_contrastive_loss"] is False
    model = s2l.get_model(cfg, device="cpu", len_dataset=123, config=None)
    assert isinstance(model, s2l.TalkingFace) and model.audio_dims == 64 and hasattr(model, "post_fusion_unet")
    assert C.method_dict["face_simple"].config.get_model is not None and set(C.method_dict) == {"face_simple"}
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    tr = s2l.get_trainer(model, opt, cfg, torch.device("cpu"))
    assert isinstance(tr, s2l.Trainer) and tr.optimizer is opt and tr.model is model and tr.cfg is cfg
    assert tr.out_dir == "log/face_simple/may" and tr.device == torch.device("cpu")
    assert tr.batch_rays == 80 * 120 and (tr.height, tr.width) == (80, 120)
    assert tr.multi_gpu is True and tr.w_photometric_loss == 1.0 and tr.w_post_fusion == 1.0
    assert tr.use_post_fusion and tr.fusion_lip_only and not tr.use_syncloss and not tr.use_perceptual_loss
    assert (tr.threshold, tr.n_sample_points, tr.n_sample_points_fine, tr.raw_noise_std) == (0.5, 16, 64, 1)
    # positional binding as in the reference: (model, optimizer, device, out_dir, cfg=...)
    tr2 = s2l.Trainer(model, opt, torch.device("cpu"), "some/dir", cfg=cfg, batch_rays=9600, lambda_rgb=0.5, use_time=True)
    assert tr2.device == torch.device("cpu") and tr2.out_dir == "some/dir" and tr2.cfg is cfg and tr2.w_photometric_loss == 0.5
    with pytest.raises(TypeError):
        s2l.Trainer(model, opt, cfg)                       # round 3's order (cfg third) must not bind silently
    with pytest.raises(NotImplementedError):
        s2l.Trainer(model, opt, "cpu", "d", cfg=cfg, use_head_pose=True)
    with pytest.raises(TypeError):
        s2l.Trainer(model, opt, "cpu", "d", cfg=cfg, no_such_flag=1)
    # fix_post_net (training.py:121-129): parameters frozen, sub-module in eval mode
    cfg2 = _reference_style_config()
    cfg2["training"]["fix_post_net"] = True
    m2 = s2l.get_model(cfg2, device="cpu")
    s2l.get_trainer(m2, None, cfg2, "cpu")
    assert not any(p.requires_grad for p in m2.post_fusion_unet.parameters()) and not m2.post_fusion_unet.training
    with pytest.raises(KeyError):
        s2l.get_model({**cfg, "method": "nerf"}, device="cpu")


@pytest.mark.skipif(not os.path.exists("/root/reference/configs/face_simple_configs/may/may.yaml"),
                    reason="the reference checkout only exists in the build container")
def test_factories_accept_the_reference_may_yaml():
    """The reference's own may.yaml over its defaults, loaded by our load_config, goes through get_model / get_trainer unchanged
    (except the two switches that need files outside the repository: the 3DMM depth init and the expert / LPIPS weights)."""
    import speech2lip_amd as s2l
    cfg = s2l.load_config("configs/face_simple_configs/may/may.yaml", "configs/default.yaml", abs_path="/root/reference")
    assert cfg["method"] == "face_simple" and cfg["training"]["use_sync_contrastive_loss"] is True
    cfg["model"]["use_canonical_depth"] = False
    cfg["training"].update(use_perceptual_loss=False, use_syncloss=False)
    model = s2l.get_model(cfg, device="cpu", len_dataset=None, config=None)
    tr = s2l.get_trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), cfg, "cpu")
    assert (tr.height, tr.width, tr.batch_rays) == (80, 120, 9600) and tr.multi_gpu is True
    assert tr.w_perceptual_loss == 0.01 and tr.w_syncloss == 0.01 and tr.use_post_fusion


def test_bench_line_stays_inside_the_drivers_tail():
    """bench.compact: the `extra` object that goes into the ONE JSON line keeps numbers only (descriptive strings live in
    docs/BENCH_LEGEND.md, the uncut object in gpurun_out/bench_extra_full.json) -- a realistic full object shrinks below 6 KB."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    leg = {"config": "x" * 300, "ms_per_step": 41.4321987, "tflops": 642.81234, "loss_first": 0.7, "loss_last": 0.1, "peak_mem_gb": 39.6,
           "roofline": {"bound": "hbm", "achieved": 5505.123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.688140432},
           "parity": {"rmse_vs_cpu": 1.418e-06, "psnr_db_vs_cpu": 117.0}, "error": "RuntimeError: " + "y" * 60}
    full = {f"leg{i}": dict(leg) for i in range(20)}
    small = bench.compact(full)
    assert len(json.dumps(small, separators=(",", ":"))) < 6000 < len(json.dumps(full))
    assert small["leg0"]["ms_per_step"] == 41.43 and small["leg0"]["roofline"] == {"achieved": 5505.0, "frac": 0.6881}
    assert "config" not in small["leg0"] and "loss_last" not in small["leg0"] and small["leg0"]["error"].startswith("RuntimeError")
    legend = open(os.path.join(ROOT, "docs", "BENCH_LEGEND.md")).read()
    for name in ("composite", "render_split", "small_clips", "config4_rank_block", "config3", "unet_fp32", "train_bf16", "dropin_trainer", "infer_clip_end_to_end",
                 "eager_torch_gpu", "eager_torch_gpu_train", "stage1_early_iteration_bf16"):
        assert f"`{name}" in legend, name


def test_module_tree_cache_follows_the_tree():
    """speech2lip_amd._modcache: the cached views equal named_parameters() / state_dict() / train() and notice re-assigned
    parameters, replaced children (at any depth) and newly registered buffers."""
    import torch.nn as nn
    from speech2lip_amd._modcache import param_map, set_training, state_tensors

    class Leaf(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = nn.Linear(3, 2)
            self.bn = nn.BatchNorm1d(2)
            self.register_buffer("scratch", torch.zeros(1), persistent=False)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = Leaf(), nn.Sequential(Leaf(), nn.ReLU())
            self.shared = self.a.fc      # the same module under a second name

    def same(net):
        ref = dict(net.named_parameters())
        got = param_map(net)
        assert list(got) == list(ref) and all(got[k] is ref[k] for k in ref)
        sd = net.state_dict()
        st = dict(state_tensors(net))
        assert set(st) <= set(sd) and all(st[k].data_ptr() == sd[k].data_ptr() for k in st)
        assert {v.data_ptr() for v in sd.values()} == {v.data_ptr() for v in st.values()}      # (a shared module's tensors appear once)

    net = Net()
    same(net)
    net.a.fc.weight = nn.Parameter(torch.ones(2, 3))          # re-assigned parameter
    same(net)
    net.b[0] = Leaf()                                        # replaced child two levels down
    same(net)
    net.b[0].register_buffer("extra", torch.ones(2))          # new persistent buffer
    same(net)
    net.eval()
    net.b[0].bn.train()
    assert set_training(net, True) is net and all(m.training for m in net.modules())
    set_training(net, False)
    assert not any(m.training for m in net.modules())

    class Frozen(nn.Module):      # a module that overrides train(): the override is honoured
        def train(self, mode=True):
            return super().train(False)

    net.c = Frozen()
    set_training(net, True)
    assert net.training and net.a.training and not net.c.training


def test_hot_tensor_slots_follow_nested_module_replacement():
    """`TalkingFace._hot_tensors` / `SimpleUnetLight._tensors` (what `packed_weights()` keys on) must see a sub-module that is
    replaced BELOW the top level -- `model.audio_net.encoder_conv[0] = ...`, `unet.inc.double_conv[1] = ...`,
    `nn.SyncBatchNorm.convert_sync_batchnorm(unet)`, `add_module`, `del` + re-register -- not only a top-level `__setattr__`
    (the reference resolves its parameters through the live module tree on every forward, tf_nerf.py:197-285)."""
    import torch.nn as nn
    from speech2lip_amd import _abi
    m = s2l.TalkingFace(torch.device("cpu"), s2l.may_config(16, 16), mode="eval")
    named = dict(m.named_parameters())
    assert all(a is named[n] for a, n in zip(m._hot_tensors(), _abi.TENSOR_ORDER))
    # a nested replacement (index into a Sequential two levels down)
    name0 = next(n for n in _abi.TENSOR_ORDER if n.count(".") >= 2)
    path, _, attr = name0.rpartition(".")
    parent_path, _, key = path.rpartition(".")
    parent = m.get_submodule(parent_path)
    old = getattr(parent, key) if not key.isdigit() else parent[int(key)]
    new = type(old)(*([old.in_channels, old.out_channels, old.kernel_size[0], old.stride[0], old.padding[0]] if isinstance(old, nn.Conv1d)
                      else [old.in_features, old.out_features]))
    parent.add_module(key, new)
    named = dict(m.named_parameters())
    got = dict(zip(_abi.TENSOR_ORDER, m._hot_tensors()))
    assert got[name0] is named[name0] and got[name0] is getattr(new, attr)
    assert all(got[n] is named[n] for n in _abi.TENSOR_ORDER)
    # a re-assigned parameter inside an unchanged module
    m.fc_uv.weight = nn.Parameter(torch.zeros_like(m.fc_uv.weight))
    assert dict(zip(_abi.TENSOR_ORDER, m._hot_tensors()))["fc_uv.weight"] is m.fc_uv.weight
    # a top-level replacement still works
    m.fc_time = nn.Linear(m.fc_time.in_features, m.fc_time.out_features)
    assert dict(zip(_abi.TENSOR_ORDER, m._hot_tensors()))["fc_time.weight"] is m.fc_time.weight

    u = s2l.SimpleUnetLight()
    from speech2lip_amd.unet import _TENSOR_NAMES
    state = dict(u.state_dict(keep_vars=True))
    assert all(t is state[n] for t, n in zip(u._tensors(), _TENSOR_NAMES))
    u.inc.double_conv[1] = nn.BatchNorm2d(64)
    state = dict(u.state_dict(keep_vars=True))
    assert all(t is state[n] for t, n in zip(u._tensors(), _TENSOR_NAMES))
    conv = nn.SyncBatchNorm.convert_sync_batchnorm(u)
    state = dict(conv.state_dict(keep_vars=True))
    assert conv is u and isinstance(u.inc.double_conv[1], nn.SyncBatchNorm)
    assert all(t is state[n] for t, n in zip(u._tensors(), _TENSOR_NAMES))
    del u.outc.conv
    u.outc.conv = nn.Conv2d(64, 3, kernel_size=1)
    assert u._tensors()[-1] is u.outc.conv.bias
