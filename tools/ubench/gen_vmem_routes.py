"""Generates tools/ubench/vmem_routes.hip: the two un-priced routes on the vector-memory side of the half-width convolution
(csrc/gen_convh8_body.py), measured next to tools/ubench/gen_mfma_segments.py with MFMAs co-running so that the clock is the loaded one.

PART A -- what does ONE CU sustain on a weight stream, by route?  Four waves (one per SIMD) spin v_mfma_f32_32x32x16_bf16 on registers;
k = 1 / 2 / 4 / 8 further waves stream 1-KiB pieces into the LDS buffers as fast as they can, 8 requests in flight each:
    dma    global_load_lds_dwordx4                (the kernel's route: L2 -> texture path -> LDS, no registers)
    vgpr   global_load_dwordx4 -> VGPR -> ds_write_b128
from  l2   a 590-KB table every workgroup walks (the 256 -> 128 layer's weights: L2-resident, each byte once per CU and pass), or
      hbm  a private 16-MiB region per workgroup (4 GiB in all: streams from HBM like the halo planes).
Reported: bytes per clock per CU, aggregate TB/s, the clock.

PART B -- who should wait on vmcnt?  The convolution's chunk loop at its real LDS traffic (72 MFMAs behind 72 ds_read_b128 per wave and
chunk; 36 KiB of weights from the shared table + 40 KiB of halo from the private HBM region per chunk by LDS-DMA into the other buffer; every
T chunks a tile's 64 KiB leave as 1-KiB global stores):
    own      today: every MFMA wave issues its share of the requests and its stores, waits vmcnt(0) before the chunk's barrier
    loader   eight MFMA waves that touch only LDS (lgkmcnt) and issue the tile's stores WITHOUT ever waiting on them; a NINTH wave owns every
             DMA request and is the only one that waits on vmcnt (9 waves = 3 on one SIMD: <= 168 registers per wave)
    loader_vgpr  the ninth wave brings the weights by the VGPR route (AGPR destinations) and the halo by LDS-DMA
T = 2 without the weight requests (the 64 -> 64 layers: weights resident in the two buffers) and T = 8 / 16 with them.
Reported: cycles per chunk (4608 = the matrix pipe's minimum), PFLOP/s, clock.
    python tools/ubench/gen_vmem_routes.py && hipcc --offload-arch=gfx950 -O3 tools/ubench/vmem_routes.hip -o ubin/vmem_routes"""
import os

NB = 2
BUF = 76032
HALO, WBYTES = 39168, 36864
TABLE = 16 * WBYTES                     # 589 824 B: 256 -> 128 channels x 9 taps x bf16
PRIV = 16 << 20                         # private region per workgroup


def acc(mb, nb):
    b = (mb * NB + nb) * 16
    return f"a[{b}:{b + 15}]"


def mfma(a, b, mb, nb):
    return f"v_mfma_f32_32x32x16_bf16 {acc(mb, nb)}, a[{a}:{a + 3}], a[{b}:{b + 3}], {acc(mb, nb)}"


def tap_regs(base):
    return [[base + (ks * 2 + mb) * 4 for mb in range(2)] for ks in range(2)], [[base + 16 + (ks * 2 + nb) * 4 for nb in range(2)] for ks in range(2)]


def tap_reads(t, base, buf):
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


def asm(lines):
    return "\n".join(f'      "{x}\\n\\t"' for x in lines)


LDS_INIT = f"""  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* l = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < {2 * BUF // 4}; i += blockDim.x) {{      // random bf16 pairs, +-[0.5, 2)
    unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    l[i] = 0x3f803f80u ^ (h & 0x80ff80ffu);
  }}
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)smem;
"""


# ======================================================================================================================= PART A
def part_a_kernel(route, k):
    """waves 0..3 (one per SIMD): a fixed, longer loop of MFMAs on registers; waves 4..4+k-1 stream 8-KiB groups of eight 1-KiB pieces
    (group g of wave sw at byte (g * k + sw) * 8192 of the pass, wrapping at the end of the region), 8 requests in flight behind 8 landing;
    their cycle counts go to clk[block][wave].  Register budget: k = 8 means 12 waves = 3 per SIMD = 168 registers per wave."""
    name = f"a_{route}_{k}"
    nthreads = 64 * (4 + k)
    mf = ["s_mov_b32 s20, %[miters]"] + [f"v_accvgpr_write_b32 a{r}, 0" for r in range(64)] + [f"v_accvgpr_write_b32 a{r}, %[seed]" for r in range(64, 80)]
    mf += ["1:"]
    for t in range(8):      # 32 MFMAs per iteration on four accumulators, two A and two B operands (80 AGPRs: the 12-wave forms have 168 registers per wave)
        mf += [mfma(64 + 4 * mb, 72 + 4 * nb, mb, nb) for mb in range(2) for nb in range(2)]
    mf += ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 7", "s_nop 7", "v_accvgpr_read_b32 %[r], a0"]
    mf_clob = [f"a{r}" for r in range(80)] + ["s20", "scc"]
    G = 8
    st = ["s_mov_b32 s21, %[groups]", "s_mov_b64 s[26:27], %[src]", "s_mov_b32 s24, %[ldsw]", "s_mov_b32 s25, 0", "v_mov_b32 v6, %[goff]",
          "v_mov_b32 v7, %[ldsv]"]
    wrap = ["s_add_u32 s25, s25, 1", "s_cmp_eq_u32 s25, %[gpp]", "s_cselect_b32 s23, %[passbytes], 0", "s_cselect_b32 s25, 0, s25",
            "s_sub_u32 s26, s26, s23", "s_subb_u32 s27, s27, 0"]

    def bump(i):
        if i == 3:
            return ["s_add_u32 s26, s26, 4096", "s_addc_u32 s27, s27, 0"]
        if i == 7:
            return ["s_add_u32 s26, s26, %[stride]", "s_addc_u32 s27, s27, 0"] + wrap
        return []
    if route == "dma":
        def group():
            L = []
            for i in range(G):
                L += [f"s_add_u32 m0, s24, {1024 * i}", "s_nop 0", f"global_load_lds_dwordx4 v6, s[26:27] offset:{1024 * (i % 4)}"] + bump(i)
            return L
        st += group() + ["2:"] + group() + [f"s_waitcnt vmcnt({G})"]
        st += ["s_sub_u32 s21, s21, 1", "s_cmp_gt_i32 s21, 1", "s_cbranch_scc1 2b", "s_waitcnt vmcnt(0)"]
        st_clob = ["v6", "v7", "s21", "s23", "s24", "s25", "s26", "s27", "m0", "scc", "memory"]
    else:
        def loads(base):
            L = []
            for i in range(G):
                L += [f"global_load_dwordx4 v[{base + 4 * i}:{base + 4 * i + 3}], v6, s[26:27] offset:{1024 * (i % 4)}"] + bump(i)
            return L

        def writes(base):
            return [f"ds_write_b128 v7, v[{base + 4 * i}:{base + 4 * i + 3}] offset:{1024 * i}" for i in range(G)]
        st += loads(8) + ["2:"] + loads(40) + [f"s_waitcnt vmcnt({G})"] + writes(8) + loads(8) + [f"s_waitcnt vmcnt({G})"] + writes(40)
        st += ["s_sub_u32 s21, s21, 2", "s_cmp_gt_i32 s21, 2", "s_cbranch_scc1 2b", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
        st_clob = [f"v{r}" for r in range(6, 72)] + ["s21", "s23", "s24", "s25", "s26", "s27", "scc", "memory"]
    return name, nthreads, f"""__global__ __launch_bounds__({nthreads}) void k_{name}(float* out, const char* table, const char* priv, int miters, int groups, int from_hbm,
                                                       long long* clk) {{
{LDS_INIT}
  float r = 0.f;
  if (wave < 4) {{
    const unsigned seed = 0x3f803f80u ^ ((lane * 0x01010101u) & 0x007f007fu);
    asm volatile(
{asm(mf)}
        : [r] "=v"(r) : [miters] "s"(miters), [seed] "v"(seed) : {", ".join('"%s"' % c for c in mf_clob)});
  }} else {{
    const int sw = wave - 4;                                   // streaming wave 0..k-1
    const unsigned passbytes = from_hbm ? {PRIV}u : {TABLE}u, gpp = passbytes / 8192 / {k};      // groups of this wave per pass over the region
    const char* mine = (from_hbm ? priv + (size_t)blockIdx.x * {PRIV} : table) + (size_t)sw * 8192;
    const unsigned ldsw = lds0 + sw * 8192, ldsv = ldsw + lane * 16, goff = lane * 16;
    const unsigned stride = {k} * 8192 - 4096;
    long long t0 = __builtin_readcyclecounter();
    asm volatile(
{asm(st)}
        : : [groups] "s"(groups), [src] "s"(mine), [ldsw] "s"(ldsw), [goff] "v"(goff), [ldsv] "v"(ldsv), [stride] "s"(stride), [gpp] "s"(gpp),
            [passbytes] "s"(passbytes)
        : {", ".join('"%s"' % c for c in st_clob)});
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) clk[blockIdx.x * 16 + sw] = t1 - t0;
  }}
  out[blockIdx.x * {nthreads} + threadIdx.x] = r;
}}
"""


# ======================================================================================================================= PART B
def dma_req(i, buf, vreg, sbase, lds_s):
    return [f"s_add_u32 m0, {lds_s}, {buf * BUF + 1024 * i}", "s_nop 0", f"global_load_lds_dwordx4 v{vreg}, {sbase} offset:{1024 * (i % 4)}"]


def stores(vdata, n=8):
    """n 1-KiB stores of this wave's share of a tile's output (source registers: whatever they hold), to s[28:29] + v9"""
    L = []
    for i in range(n):
        L += [f"global_store_dwordx4 v9, v[{vdata}:{vdata + 3}], s[28:29] offset:{1024 * (i % 4)}"]
        if i % 4 == 3:
            L += ["s_add_u32 s28, s28, 4096", "s_addc_u32 s29, s29, 0"]
    return L


def part_b_own(T, with_w):
    """today's form.  Per wave and chunk: weights 4.5 pieces (4 + one more for waves 0..3), halo 5 pieces; here 5 + 5 for waves 0..3 and 4 + 5
    for the others = 76 pieces = 76 KiB per chunk (the halo region is 40 KiB in this model)."""
    L = []
    for p in range(2):
        L += tap_reads(0, 64, p)
        for t in range(9):
            base = 64 + 32 * (t & 1)
            nxt = tap_reads(t + 1, 64 + 32 * ((t + 1) & 1), p) if t < 8 else []
            L.append("s_waitcnt lgkmcnt(0)")
            ms = tap_mfmas(base)
            for m in range(8):
                L.append(ms[m])
                if m < len(nxt):
                    L.append(nxt[m])
                if with_w and t == 0 and 3 <= m <= 6:                      # weight pieces behind tap 0 (s[26:27] = this chunk's weights + wave * 4 KiB)
                    L += dma_req(m - 3, p ^ 1, 6, "s[26:27]", "s24")
                if t in (1, 2) and m in (2, 4, 6) and not (t == 2 and m == 6):   # five halo pieces behind taps 1..2 (s[30:31] = halo stream)
                    i = (t - 1) * 3 + (m - 2) // 2
                    L += dma_req(i, p ^ 1, 7, "s[30:31]", "s25")
                    if i == 3:
                        L += ["s_add_u32 s30, s30, 4096", "s_addc_u32 s31, s31, 0"]
            if t == 2:
                L += ["s_add_u32 s30, s30, %[hstep]", "s_addc_u32 s31, s31, 0"]   # next chunk's halo: 40 KiB further (minus the 4 KiB bumped above)
        if with_w:
            L += ["s_add_u32 s26, s26, %[wstep]", "s_addc_u32 s27, s27, 0", "s_add_u32 s22, s22, 1", "s_cmp_eq_u32 s22, 16",
                  "s_cselect_b32 s23, %[wback], 0", "s_cselect_b32 s22, 0, s22", "s_sub_u32 s26, s26, s23", "s_subb_u32 s27, s27, 0"]
        L += ["s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)", "s_barrier"]
    return L


def part_b_mfma_only_lds(T):
    """the eight MFMA waves of the loader forms: LDS reads + MFMAs + the chunk's barrier, nothing else"""
    L = []
    for p in range(2):
        L += tap_reads(0, 64, p)
        for t in range(9):
            base = 64 + 32 * (t & 1)
            nxt = tap_reads(t + 1, 64 + 32 * ((t + 1) & 1), p) if t < 8 else []
            L.append("s_waitcnt lgkmcnt(0)")
            ms = tap_mfmas(base)
            for m in range(8):
                L.append(ms[m])
                if m < len(nxt):
                    L.append(nxt[m])
        L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    return L


def part_b_loader(with_w, vgpr_w):
    """the ninth wave, per pair of chunks: every request of the NEXT chunk into the other buffer, vmcnt(0), barrier.  32 weight pieces (as
    the `own` model) + 40 halo pieces per chunk."""
    L = []
    for p in range(2):
        if with_w:
            if vgpr_w:                                              # the 32 weight pieces in ONE round through a[0:127] -> 32 ds_write_b128
                for j in range(32):
                    L += [f"global_load_dwordx4 a[{4 * j}:{4 * j + 3}], v6, s[26:27] offset:{1024 * (j % 4)}"]
                    if j % 4 == 3:
                        L += ["s_add_u32 s26, s26, 4096", "s_addc_u32 s27, s27, 0"]
                for i in range(40):                                 # the halo requests ride behind them (vmcnt is in order: 40 newer requests)
                    L += dma_req(i, p ^ 1, 7, "s[30:31]", "s25")
                    if i % 4 == 3:
                        L += ["s_add_u32 s30, s30, 4096", "s_addc_u32 s31, s31, 0"]
                L += ["s_waitcnt vmcnt(40)"]
                for j in range(32):
                    L += [f"ds_write_b128 v{10 + (p ^ 1)}, a[{4 * j}:{4 * j + 3}] offset:{1024 * j}"]
                L += ["s_waitcnt lgkmcnt(0)"]
            else:
                for j in range(32):
                    L += dma_req(j, p ^ 1, 6, "s[26:27]", "s24")
                    if j % 4 == 3:
                        L += ["s_add_u32 s26, s26, 4096", "s_addc_u32 s27, s27, 0"]
            L += ["s_add_u32 s26, s26, 4096", "s_addc_u32 s27, s27, 0",      # (32 of the chunk's 36 KiB: skip the rest)
                  "s_add_u32 s22, s22, 1", "s_cmp_eq_u32 s22, 16", "s_cselect_b32 s23, %[wback], 0", "s_cselect_b32 s22, 0, s22",
                  "s_sub_u32 s26, s26, s23", "s_subb_u32 s27, s27, 0"]
        if not (with_w and vgpr_w):
            for i in range(40):
                L += dma_req(i, p ^ 1, 7, "s[30:31]", "s25")
                if i % 4 == 3:
                    L += ["s_add_u32 s30, s30, 4096", "s_addc_u32 s31, s31, 0"]
        L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    return L


def part_b_kernel(form, T, with_w):
    name = f"b_{form}_T{T}" + ("" if with_w else "_nw")
    nine = form != "own"
    nthreads = 576 if nine else 512
    pro = ["s_mov_b32 s20, %[iters]", "s_mov_b32 s21, 0", "s_mov_b32 s22, 0", "s_mov_b32 s24, %[ldsw]", "s_mov_b32 s25, %[ldsh]",
           "s_mov_b64 s[26:27], %[wsrc]", "s_mov_b64 s[30:31], %[hsrc]", "s_mov_b64 s[28:29], %[osrc]",
           "v_mov_b32 v2, %[a0]", "v_mov_b32 v4, %[b0]", "v_mov_b32 v6, %[goff]", "v_mov_b32 v7, %[goff]", "v_mov_b32 v9, %[goff]",
           f"v_add_u32 v3, {BUF}, v2", f"v_add_u32 v5, {BUF}, v4", "v_mov_b32 v10, %[wldsv]", f"v_add_u32 v11, {BUF}, v10"]
    pro += [f"v_accvgpr_write_b32 a{r}, 0" for r in range(64)] + [f"v_accvgpr_write_b32 a{r}, v6" for r in range(64, 128)]
    # a tile ends every T chunks = T/2 loop iterations
    tile_end = ["s_add_u32 s21, s21, 2", f"s_cmp_lt_u32 s21, {T}", "s_cbranch_scc1 8f", "s_mov_b32 s21, 0"] + stores(12) + ["8:"]
    loop_tail = ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0"]
    if not nine:
        body = pro + ["s_nop 7", "1:"] + part_b_own(T, with_w) + tile_end + loop_tail + ["s_cbranch_scc1 1b"]
    else:
        body = pro + ["s_cmp_eq_u32 %[wave], 8", "s_cbranch_scc1 2f", "1:"] + part_b_mfma_only_lds(T) + tile_end + loop_tail + ["s_cbranch_scc1 1b", "s_branch 3f",
                      "2:"] + part_b_loader(with_w, form == "loader_vgpr") + loop_tail + ["s_cbranch_scc1 2b", "3:"]
    body += ["s_nop 7", "s_nop 7", "s_waitcnt vmcnt(0) lgkmcnt(0)", "v_accvgpr_read_b32 %[r], a0"]
    clob = [f"v{r}" for r in range(2, 16)] + [f"a{r}" for r in range(128)] + [f"s{r}" for r in range(20, 32)] + ["m0", "scc", "memory"]
    return name, nthreads, f"""__global__ __launch_bounds__({nthreads}) void k_{name}(float* out, const char* table, const char* priv, char* sink, int iters, long long* clk) {{
{LDS_INIT}
  // own form: wave w's weight pieces are 4 KiB apart (4 per wave = 32 KiB; the model drops the kernel's last 4 KiB), its five halo pieces
  // 5 KiB apart; loader forms: the ninth wave walks everything
  const int w8 = wave & 7;
  const unsigned a0 = lds0 + {HALO} + lane * 16, b0 = lds0 + lane * 16;
  const unsigned ldsw = lds0 + {HALO} + ({'0' if nine else 'w8 * 4096'}), ldsh = lds0 + ({'0' if nine else 'w8 * 5120'});
  const unsigned wldsv = lds0 + {HALO} + lane * 16;
  const unsigned goff = lane * 16;
  const char* wsrc = table + ({'0' if nine else '(size_t)w8 * 4096'});
  const char* hsrc = priv + (size_t)blockIdx.x * {PRIV} + ({'0' if nine else '(size_t)w8 * 5120'});
  char* osrc = sink + (size_t)blockIdx.x * {PRIV} + (size_t)w8 * {PRIV // 8};
  const unsigned wstep = {WBYTES}, wback = {TABLE}, hstep = {40960 - 4096};
  long long t0 = __builtin_readcyclecounter();
  float r;
  asm volatile(
{asm(body)}
      : [r] "=v"(r)
      : [iters] "s"(iters), [ldsw] "s"(ldsw), [ldsh] "s"(ldsh), [wsrc] "s"(wsrc), [hsrc] "s"(hsrc), [osrc] "s"(osrc), [a0] "v"(a0), [b0] "v"(b0),
        [goff] "v"(goff), [wave] "s"(wave), [wldsv] "v"(wldsv), [wstep] "s"(wstep), [wback] "s"(wback), [hstep] "s"(hstep)
      : {", ".join('"%s"' % c for c in clob)});
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * {nthreads} + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}}
"""


def main():
    src = ["// GENERATED by tools/ubench/gen_vmem_routes.py -- do not edit.", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdlib>",
           "#include <cstring>", ""]
    a_forms, b_forms = [], []
    for route in ("dma", "vgpr"):
        for k in (1, 2, 4, 8):
            name, nt, text = part_a_kernel(route, k)
            src.append(text)
            a_forms.append((name, nt, route, k))
    for form in ("own", "loader", "loader_vgpr"):
        for T, with_w in ((2, False), (8, True), (16, True)):
            if form == "loader_vgpr" and not with_w:
                continue
            name, nt, text = part_b_kernel(form, T, with_w)
            src.append(text)
            b_forms.append((name, nt, form, T, with_w))
    src.append(f"""#define CK(x) do {{ hipError_t e_ = (x); if (e_ != hipSuccess) {{ printf("HIP error %s line %d\\n", hipGetErrorString(e_), __LINE__); exit(1); }} }} while (0)
static const int LDS = 2 * {BUF} + 1024;
template <typename K> void run_a(const char* name, K k, int nthreads, int kw, int from_hbm, float* out, const char* table, const char* priv, long long* clk) {{
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  // per streaming wave: `groups` groups of 8 KiB; a pass over the table is {TABLE} bytes = 72 groups in all
  const int groups = from_hbm ? ({PRIV} / 8192 / kw) & ~1 : (72 * 24 / kw) & ~1;      // hbm: the whole private region once; l2: 24 passes over the table
  const int miters = from_hbm ? 9000 : 9000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(256), dim3(nthreads), LDS, 0, out, table, priv, 200, 16, from_hbm, clk);
  CK(hipDeviceSynchronize());
  double best = 1e30; float ms_best = 0;
  for (int rep = 0; rep < 3; ++rep) {{
    CK(hipMemset(clk, 0, 256 * 16 * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(nthreads), LDS, 0, out, table, priv, miters, groups, from_hbm, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    static long long h[256 * 16];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double sum = 0;                                    // per CU: the slowest of its streaming waves
    for (int b = 0; b < 256; ++b) {{ long long m = 0; for (int w = 0; w < kw; ++w) if (h[b * 16 + w] > m) m = h[b * 16 + w]; sum += (double)m; }}
    if (sum / 256 < best) {{ best = sum / 256; ms_best = ms; }}
  }}
  const double bytes = (double)groups * 8192.0 * kw;   // per CU
  // the clock: the MFMA waves' 9000 x 32 MFMAs x 8 passes x 4 cycles = 9.2 M cycles take ms_best when they outlast the stream (they do)
  const double clk_ghz = 9000.0 * 32 * 32 / (ms_best * 1e-3) / 1e9;
  printf("A %-6s %-4s waves %d  %7.2f B/clk/CU  %6.2f TB/s aggregate at %.2f GHz  (stream %.0f kcycles of %.0f; %.1f KiB per CU)\\n", strstr(name, "vgpr") ? "vgpr" : "dma",
         from_hbm ? "hbm" : "l2", kw, bytes / best, bytes / best * 256 * clk_ghz / 1e3, clk_ghz, best / 1e3, 9000.0 * 32 * 32 / 1e3, bytes / 1024);
}}
template <typename K> void run_b(const char* name, K k, int nthreads, float* out, const char* table, const char* priv, char* sink, long long* clk) {{
  const int iters = 200;                                 // x 2 chunks x 40 KiB of halo = 16.0 MB: one pass over the workgroup's private 16-MiB region (4 GiB in all, from HBM)
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(256), dim3(nthreads), LDS, 0, out, table, priv, sink, 20, clk);
  CK(hipDeviceSynchronize());
  float best = 1e30f; long long cyc = 0;
  for (int rep = 0; rep < 3; ++rep) {{
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(nthreads), LDS, 0, out, table, priv, sink, iters, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) {{ best = ms; CK(hipMemcpy(&cyc, clk, 8, hipMemcpyDeviceToHost)); }}
  }}
  const double chunks = 2.0 * iters, flop = 256.0 * 8 * chunks * 72 * 32768.0;
  printf("B %-22s %8.0f cycles/chunk  pipe %5.3f  %7.3f ms  %6.3f PFLOP/s  clock %.2f GHz\\n", name, cyc / chunks, 4608.0 / (cyc / chunks), best,
         flop / (best * 1e-3) / 1e15, cyc / (best * 1e-3) / 1e9);
}}
int main(int argc, char** argv) {{
  const char* only = argc > 1 ? argv[1] : "";
  float* out; char *table, *priv, *sink; long long* clk;
  const size_t region = 256ull * {PRIV} + (1 << 20);
  CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&table, {TABLE} + 65536)); CK(hipMalloc(&priv, region)); CK(hipMalloc(&sink, region)); CK(hipMalloc(&clk, 256 * 16 * 8));
  {{   // random bf16 pairs everywhere the DMA reads
    unsigned* h = (unsigned*)malloc(region);
    unsigned x = 12345u;
    for (size_t i = 0; i < region / 4; ++i) {{ x = x * 1664525u + 1013904223u; h[i] = 0x3f803f80u ^ (x & 0x80ff80ffu); }}
    CK(hipMemcpy(priv, h, region, hipMemcpyHostToDevice));
    CK(hipMemcpy(table, h, {TABLE} + 65536, hipMemcpyHostToDevice));
    free(h);
  }}""")
    for name, nt, route, k in a_forms:
        for hbm in (0, 1):
            src.append(f'  if (!*only || *only == \'a\') run_a("{name}", k_{name}, {nt}, {k}, {hbm}, out, table, priv, clk);')
    for name, nt, form, T, with_w in b_forms:
        src.append(f'  if (!*only || *only == \'b\') run_b("{name}", k_{name}, {nt}, out, table, priv, sink, clk);')
    src.append("  return 0;\n}")
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vmem_routes.hip"), "w").write("\n".join(src) + "\n")
    print("wrote vmem_routes.hip")


if __name__ == "__main__":
    main()
