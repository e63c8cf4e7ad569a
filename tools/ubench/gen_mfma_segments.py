"""Generates tools/ubench/mfma_segments.hip: what does an LDS-fed bf16 MFMA loop of the half-width convolution's shape sustain
when the two waves of a SIMD ALTERNATE roles (one in an MFMA-only segment on operands already in registers, the partner in
the LDS-read [+ LDS-DMA] segment, swapped at s_barrier: MI355X_MICROARCH.md "Two waves per SIMD") instead of both
interleaving their reads with their MFMAs (csrc/gen_convh8_body.py today)?  Same LDS traffic in every form: per wave and
"chunk" 72 v_mfma_f32_32x32x16_bf16 (9 taps x 2 k-steps x 2 x 2 blocks) behind 72 ds_read_b128 (one 1-KiB operand read per
MFMA), 8 waves per CU, random bf16 data (the clock is power-governed: zero data would flatter every form).  The `_dma` forms add
the convolution's 76 KiB of global_load_lds per chunk (ten 1-KiB pieces per wave, from an L2-resident source).
Prints, per form: cycles per chunk, MFMA-pipe occupancy (4608 = the pipe's minimum per chunk and SIMD), wall-clock PFLOP/s and
the implied shader clock.
    python tools/ubench/gen_mfma_segments.py && hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_segments.hip -o ubin/mfma_segments"""
import os

NB = 2
BUF = 76032


def acc(mb, nb):
    b = (mb * NB + nb) * 16
    return f"a[{b}:{b + 15}]"


def mfma(a, b, mb, nb):
    return f"v_mfma_f32_32x32x16_bf16 {acc(mb, nb)}, a[{a}:{a + 3}], a[{b}:{b + 3}], {acc(mb, nb)}"


def tap_regs(base):
    """operand registers of one tap in a 32-register set at `base`: A[ks][mb], B[ks][nb]"""
    return [[base + (ks * 2 + mb) * 4 for mb in range(2)] for ks in range(2)], [[base + 16 + (ks * 2 + nb) * 4 for nb in range(2)] for ks in range(2)]


def tap_reads(t, base, buf):
    """8 ds_read_b128 of tap t (4 A from the weight area, 4 B from the halo area) in the order the MFMAs use them"""
    A, B = tap_regs(base)
    out = []
    for ks in range(2):
        for mb in range(2):
            out.append(f"ds_read_b128 a[{A[ks][mb]}:{A[ks][mb] + 3}], v{2 + buf} offset:{((t * 2 + ks) * 2 + mb) * 1024}")
        for nb in range(2):
            out.append(f"ds_read_b128 a[{B[ks][nb]}:{B[ks][nb] + 3}], v{4 + buf} offset:{(t * 4 + ks * 2 + nb) * 1024}")
    return out


def tap_mfmas(base):
    A, B = tap_regs(base)
    return [mfma(A[ks][mb], B[ks][nb], mb, nb) for ks in range(2) for mb in range(2) for nb in range(2)]


def dma(i, buf):
    """piece i (0..9) of this wave's share of the next chunk's 76 KiB -> buffer buf (s24 = LDS base + wave * 9504 rounded, s[26:27] = source)"""
    return [f"s_add_u32 m0, s24, {buf * BUF + 1024 * i}", "s_nop 0", f"global_load_lds_dwordx4 v{6 + i // 4}, s[26:27] offset:{1024 * (i % 4)}"]


def interleaved(with_dma):
    """today's form: reads of tap t+1 (other operand set) behind the MFMAs of tap t, one per MFMA; both waves of a SIMD run it"""
    L = []
    for p in range(2):                      # two chunks: buffer 0, buffer 1
        L += tap_reads(0, 64, p)
        for t in range(9):
            base = 64 + 32 * (t & 1)
            nxt = tap_reads(t + 1, 64 + 32 * ((t + 1) & 1), p) if t < 8 else []
            L.append("s_waitcnt lgkmcnt(0)" if t == 0 else f"s_waitcnt lgkmcnt(0)")
            ms = tap_mfmas(base)
            for m in range(8):
                L.append(ms[m])
                if m < len(nxt):
                    L.append(nxt[m])
                if with_dma and t < 5 and m in (3, 6):
                    L += dma(2 * t + (m == 6), p ^ 1)
        L += ["s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)", "s_barrier"]
    return L


def alternating(with_dma, prio):
    """waves 0-3 (one per SIMD) and 4-7 in opposite roles: a load segment reads the 24 operand quads of three taps (96 registers),
    a compute segment issues their 24 MFMAs back to back; s_barrier between segments.  Returns (group-0 stream, group-1 stream)."""
    def load(seg, p, pieces):
        L = []
        n = 0
        for t in range(3 * seg, 3 * seg + 3):
            for r in tap_reads(t, 64 + 32 * (t % 3), p):
                L.append(r)
                n += 1
                if n > 15:
                    L.append("s_waitcnt lgkmcnt(14)")
        for i in pieces:
            L += dma(i, p ^ 1)
        L.append("s_waitcnt lgkmcnt(0)")
        return L

    def compute():
        L = []
        for k in range(3):
            L += tap_mfmas(64 + 32 * k)
        return L
    split = ([0, 1, 2, 3], [4, 5, 6], [7, 8, 9]) if with_dma else ([], [], [])
    g0, g1 = [], []
    if prio:
        g1.append("s_setprio 1")
    for p in range(2):
        for seg in range(3):
            last = seg == 2
            g0 += load(seg, p, split[seg]) + ["s_barrier"] + compute() + (["s_waitcnt vmcnt(0)"] if last else []) + ["s_barrier"]
            g1 += compute() + ["s_barrier"] + load(seg, p, split[seg]) + (["s_waitcnt vmcnt(0)"] if last else []) + ["s_barrier"]
    return g0, g1


def mfma_only():
    L = []
    for p in range(2):
        for t in range(9):
            L += tap_mfmas(64 + 32 * (t & 1))
        L.append("s_barrier")
    return L


def asm(lines):
    return "\n".join(f'      "{x}\\n\\t"' for x in lines)


def kernel(name, g0, g1=None):
    pro = ["s_mov_b32 s20, %[iters]", "s_mov_b32 s24, %[ldsw]", "s_mov_b64 s[26:27], %[src]", "v_mov_b32 v2, %[a0]", "v_mov_b32 v4, %[b0]",
           "v_mov_b32 v6, %[goff]", "v_add_u32 v7, 4096, v6", "v_add_u32 v8, 8192, v6", f"v_add_u32 v3, {BUF}, v2", f"v_add_u32 v5, {BUF}, v4"]
    pro += [f"v_accvgpr_write_b32 a{r}, 0" for r in range(64)]
    pro += [f"v_accvgpr_write_b32 a{r}, v6" for r in range(64, 160)]
    if g1 is None:
        body = pro + ["s_nop 7", "1:"] + g0 + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b"]
    else:
        body = pro + ["s_cmp_gt_u32 %[wave], 3", "s_cbranch_scc1 2f", "1:"] + g0 + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b",
                      "s_branch 3f", "2:"] + g1 + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 2b", "3:"]
    body += ["s_nop 7", "s_nop 7", "s_waitcnt vmcnt(0) lgkmcnt(0)", "v_accvgpr_read_b32 %[r], a0"]
    clob = [f"v{r}" for r in range(2, 10)] + [f"a{r}" for r in range(160)] + ["s20", "s24", "s26", "s27", "m0", "scc", "memory"]
    return f"""__global__ __launch_bounds__(512) void k_{name}(float* out, const char* src, int iters, long long* clk) {{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* l = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < {2 * BUF // 4}; i += 512) {{      // random bf16 pairs, +-[0.5, 2)
    unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    l[i] = 0x3f803f80u ^ (h & 0x80ff80ffu);
  }}
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)smem;
  const unsigned a0 = lds0 + 39168 + lane * 16, b0 = lds0 + lane * 16, ldsw = lds0 + wave * 9504 / 16 * 16;
  const unsigned goff = (wave * 10240 + lane * 16);
  const char* mine = src + (size_t)blockIdx.x * 81920;
  long long t0 = __builtin_readcyclecounter();
  float r;
  asm volatile(
{asm(body)}
      : [r] "=v"(r)
      : [iters] "s"(iters), [ldsw] "s"(ldsw), [src] "s"(mine), [a0] "v"(a0), [b0] "v"(b0), [goff] "v"(goff), [wave] "s"(wave)
      : {", ".join('"%s"' % c for c in clob)});
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}}
"""


forms = [("mfma_only", mfma_only(), None),
         ("interleaved", interleaved(False), None),
         ("alternating", *alternating(False, False)),
         ("alternating_prio", *alternating(False, True)),
         ("interleaved_dma", interleaved(True), None),
         ("alternating_dma", *alternating(True, False)),
         ("alternating_dma_prio", *alternating(True, True))]
src = ["// GENERATED by tools/ubench/gen_mfma_segments.py -- do not edit.", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdlib>", ""]
for name, g0, g1 in forms:
    src.append(kernel(name, g0, g1))
src.append("""#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <typename K> void run(const char* name, K k, float* out, const char* src, long long* clk) {
  const int lds = 2 * 76032 + 1024, iters = 3000;      // iters x 2 chunks
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, out, src, 200, clk);
  CK(hipDeviceSynchronize());
  float best = 1e30f; long long cyc = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, out, src, iters, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) { best = ms; CK(hipMemcpy(&cyc, clk, 8, hipMemcpyDeviceToHost)); }
  }
  const double chunks = 2.0 * iters, flop = 256.0 * 8 * chunks * 72 * 32768.0;
  printf("%-24s %8.0f cycles/chunk  pipe %5.3f  %7.3f ms  %6.3f PFLOP/s  clock %.2f GHz\\n", name, cyc / chunks, 4608.0 / (cyc / chunks), best,
         flop / (best * 1e-3) / 1e15, cyc / (best * 1e-3) / 1e9);
}
int main() {
  float* out; char* src; long long* clk;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&src, 256 * 81920)); CK(hipMalloc(&clk, 64));
  CK(hipMemset(src, 0x3c, 256 * 81920));""")
for name, g0, g1 in forms:
    src.append(f'  run("{name}", k_{name}, out, src, clk);')
src.append("  return 0;\n}")
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_segments.hip"), "w").write("\n".join(src) + "\n")
print("wrote mfma_segments.hip")
