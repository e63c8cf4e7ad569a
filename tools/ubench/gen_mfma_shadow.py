"""Generates tools/ubench/mfma_shadow.hip: what does VALU work cost when it is issued between the MFMAs of a saturated
matrix pipe?  One wave per SIMD, a fixed-register asm loop of 48 v_mfma_f32_16x16x4_f32 per iteration (three accumulator
chains, A operand shared by three MFMAs, as the renderer's k-quad) with a variant-specific group of extra instructions after
every MFMA.  Prints cycles per MFMA per variant (32.0 = the matrix pipe's own rate).
    python tools/ubench/gen_mfma_shadow.py && hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shadow.hip -o ubin/mfma_shadow"""
import os

VARIANTS = {
    "none": lambda i: [],
    "accread": lambda i: [f"v_accvgpr_read_b32 v{40 + i % 12}, a{16 + i % 12}"],
    "accread_max_dep": lambda i: [f"v_accvgpr_read_b32 v{40 + i % 12}, a{16 + i % 12}", f"v_max_f32 v{60 + i % 12}, 0, v{40 + i % 12}"],
    "accread_max_pipelined": lambda i: [f"v_max_f32 v{60 + i % 12}, 0, v{40 + (i + 11) % 12}", f"v_accvgpr_read_b32 v{40 + i % 12}, a{16 + i % 12}"],
    "two_max": lambda i: [f"v_max_f32 v{60 + i % 12}, 0, v{40 + i % 12}", f"v_max_f32 v{72 + i % 12}, 0, v{40 + (i + 5) % 12}"],
    "four_valu": lambda i: [f"v_accvgpr_read_b32 v{40 + i % 12}, a{16 + i % 12}", f"v_accvgpr_read_b32 v{52 + i % 6}, a{28 + i % 6}",
                            f"v_max_f32 v{60 + i % 12}, 0, v{40 + (i + 6) % 12}", f"v_max_f32 v{72 + i % 12}, 0, v{52 + (i + 3) % 6}"],
    "ds_read": lambda i: ([f"ds_read_b128 v[{84 + 4 * ((i // 12) % 3)}:{87 + 4 * ((i // 12) % 3)}], v96 offset:{1024 * ((i // 12) % 16)}"] if i % 12 == 0 else [])
                         + (["s_waitcnt lgkmcnt(2)"] if i % 12 == 0 else []),
    # ReLU through the LDS port: accumulator block -> LDS (from AGPRs), four float max atomics with 0, back into VGPRs
    "lds_relu": lambda i: {0: [f"ds_write_b128 v97, a[{16 + 4 * ((i // 4) % 3)}:{19 + 4 * ((i // 4) % 3)}]"],
                           1: ["ds_max_f32 v97, v98", "ds_max_f32 v97, v98 offset:4"],
                           2: ["ds_max_f32 v97, v98 offset:8", "ds_max_f32 v97, v98 offset:12"],
                           3: [f"ds_read_b128 v[{40 + 4 * ((i // 4) % 3)}:{43 + 4 * ((i // 4) % 3)}], v97"]}[i % 4],
    "lds_relu_half": lambda i: ({0: [f"ds_write_b128 v97, a[{16 + 4 * ((i // 8) % 3)}:{19 + 4 * ((i // 8) % 3)}]"],
                                 1: ["ds_max_f32 v97, v98", "ds_max_f32 v97, v98 offset:4"],
                                 2: ["ds_max_f32 v97, v98 offset:8", "ds_max_f32 v97, v98 offset:12"],
                                 3: [f"ds_read_b128 v[{40 + 4 * ((i // 8) % 3)}:{43 + 4 * ((i // 8) % 3)}], v97"]}[i % 4] if (i // 4) % 2 == 0 else []),
    "ds_read_used_as_A": lambda i: ([f"ds_read_b128 v[{84 + 4 * ((i // 12) % 3)}:{87 + 4 * ((i // 12) % 3)}], v96 offset:{1024 * ((i // 12) % 16)}",
                                     "s_waitcnt lgkmcnt(2)"] if i % 12 == 0 else []),
    # the convolution epilogue's stores (16 B per lane, 64-B segments at a 256-B pitch) behind MFMAs: one per 36 / 12 / 4 MFMAs
    "gstore_36": lambda i: [f"global_store_dwordx4 %5, v[40:43], %4 offset:{64 * ((i // 36) % 4)}"] if i % 36 == 0 else [],
    "gstore_12": lambda i: [f"global_store_dwordx4 %5, v[40:43], %4 offset:{64 * ((i // 12) % 4)}"] if i % 12 == 0 else [],
    "gstore_4": lambda i: [f"global_store_dwordx4 %5, v[40:43], %4 offset:{64 * ((i // 4) % 4)}"] if i % 4 == 0 else [],
    "gstore_burst16": lambda i: [f"global_store_dwordx4 %5, v[40:43], %4 offset:{64 * (k % 4)}" for k in range(16)] if i == 0 else [],
    # the same store rate with other address patterns: 16 B per lane at a 32-byte pitch (the bf16 images' two half-stores:
    # 32 half-filled 64-byte sectors per instruction) and at a 16-byte pitch (1 KiB contiguous: 16 full sectors)
    "gstore_4_pitch32": lambda i: [f"global_store_dwordx4 %6, v[40:43], %4 offset:{16 * ((i // 4) % 2)}"] if i % 4 == 0 else [],
    "gstore_4_pitch16": lambda i: [f"global_store_dwordx4 %7, v[40:43], %4 offset:{1024 * ((i // 4) % 2)}"] if i % 4 == 0 else [],
    "gstore_8_pitch32": lambda i: [f"global_store_dwordx4 %6, v[40:43], %4 offset:{16 * ((i // 8) % 2)}"] if i % 8 == 0 else [],
    "gstore_8_pitch16": lambda i: [f"global_store_dwordx4 %7, v[40:43], %4 offset:{1024 * ((i // 8) % 2)}"] if i % 8 == 0 else [],
    "accwrite": lambda i: [f"v_accvgpr_write_b32 a{16 + i % 12}, v{40 + i % 12}"],
}
NMF = 144
clob = [f"v{r}" for r in range(10, 100)] + [f"a{r}" for r in range(0, 40)] + ["s20", "scc"]


USE_LDS_A = {'ds_read_used_as_A'}


def body(fn, name=""):
    L = []
    for r in range(10, 100):
        L.append(f"v_mov_b32 v{r}, {1.0 if r % 2 else 0.5}")
    L.append("v_lshlrev_b32 v96, 4, %2")
    L.append("v_lshlrev_b32 v97, 4, %3")
    L.append("v_mov_b32 v98, 0")
    for r in range(40):
        L.append(f"v_accvgpr_write_b32 a{r}, 0")
    L += ["s_mov_b32 s20, %1", "s_nop 7", "1:"]
    for i in range(NMF):
        c = (i % 3) * 4
        a = 10 + (i // 3) % 4
        if name in USE_LDS_A:   # quad q = i // 12 uses the set loaded at quad q - 2
            a = 84 + 4 * ((i // 12 + 1) % 3) + (i // 3) % 4
        b = 20 + i % 16
        L.append(f"v_mfma_f32_16x16x4_f32 a[{c}:{c + 3}], v{a}, v{b}, a[{c}:{c + 3}]")
        L += fn(i)
    L += ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 7", "s_nop 7", "s_waitcnt vmcnt(0) lgkmcnt(0)",
          "v_accvgpr_read_b32 %0, a0"]
    return "\n".join(f'      "{x}\\n\\t"' for x in L)


src = ["// GENERATED by tools/ubench/gen_mfma_shadow.py -- do not edit.", "#include <hip/hip_runtime.h>", "#include <cstdio>", ""]
for n, (name, fn) in enumerate(VARIANTS.items()):
    src.append(f"""__global__ __launch_bounds__(256) void k_{name}(float* out, int iters, long long* clk, float* sink) {{
  __shared__ float lds[8192];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  float r;
  asm volatile(
{body(fn, name)}
      : "=v"(r) : "s"(iters), "v"(threadIdx.x & 63), "v"(threadIdx.x + 1024), "s"(sink),
        "v"((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096 + (threadIdx.x & 15) * 256 + ((threadIdx.x >> 4) & 3) * 16),
        "v"((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096 + (threadIdx.x & 63) * 32), "v"((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096 + (threadIdx.x & 63) * 16)
      : {", ".join('"%s"' % c for c in clob)});
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = r + lds[threadIdx.x ^ 1];
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[{n}] = t1 - t0;
}}
""")
src.append("int main() {\n  float* out; long long* clk; const int iters = 20000;\n  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 64 * 8); float* sink; hipMalloc(&sink, 8 << 20);\n  long long h[64];")
for n, name in enumerate(VARIANTS):
    src.append(f"  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_{name}, dim3(256), dim3(256), 0, 0, out, iters, clk, sink);")
src.append("  hipDeviceSynchronize(); hipMemcpy(h, clk, 64 * 8, hipMemcpyDeviceToHost);")
for n, name in enumerate(VARIANTS):
    src.append(f'  printf("%-24s %.2f cycles/MFMA\\n", "{name}", (double)h[{n}] / (iters * {NMF}.0));')
src.append("  return 0;\n}")
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_shadow.hip"), "w").write("\n".join(src) + "\n")
