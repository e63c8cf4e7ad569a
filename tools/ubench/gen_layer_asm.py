"""Generates tools/ubench/layer_asm.hip: the renderer's layer loop written as ONE fixed-register asm body (1 wave per SIMD,
G = 3 sample groups per wave, 192 B-operand VGPRs, 192 accumulator AGPRs, A quads from the LDS ring two quads ahead, bias
through an AGPR srcC, LDS-DMA ring of 9 x 16 KiB refilled after one s_barrier per slab, ReLU as accvgpr_read + v_max per
layer) -- to price what a hand-scheduled body costs per layer against the compiled kernel (106.9 k ticks; MFMAs alone 98.3 k).
    python tools/ubench/gen_layer_asm.py && hipcc --offload-arch=gfx950 -O3 tools/ubench/layer_asm.hip -o ubin/layer_asm"""
import os
import sys

G = 3
V_IN, V_W, V_LANE16, V_RING, V_DMAOFF, V_BIAS, V_T = 0, 192, 208, 209, 210, 211, 212
A_ACC, A_BIAS = 0, 192
S_CUR, S_T, S_LDSDST, S_WPTR, S_WBASE, S_IS, S_LAYERS, S_T2 = 20, 21, 22, 24, 26, 28, 29, 30   # S_WPTR, S_WBASE are pairs
KRING, SLAB_BYTES, NSLABS = 9, 16384, 113


class Body:
    def __init__(self, feat):
        self.L, self.lds, self.feat = [], [], feat   # self.lds: tags of LDS ops in flight, oldest first

    def e(self, s):
        self.L.append(s)

    def lds_op(self, text, tag):
        self.e(text)
        self.lds.append(tag)

    def wait_lds(self, tag):
        """wait until the LDS op `tag` (and everything older) has returned"""
        if tag not in self.lds:
            return
        newer = len(self.lds) - 1 - self.lds.index(tag)
        self.e(f"s_waitcnt lgkmcnt({newer})")
        self.lds = self.lds[len(self.lds) - newer:] if newer else []

    def acc(self, g, mb):
        b = A_ACC + (g * 16 + mb) * 4
        return f"a[{b}:{b + 3}]"

    def slab(self, mb, bias_imm):
        f = self.feat
        for j in range(16):
            pending = []
            if j == 14:
                self.wait_lds(("A", mb, 15))
                if "dma" in f:
                    self.e("s_waitcnt vmcnt(28)")
                if "barrier" in f:
                    self.e("s_barrier")
                if "dma" in f:
                    self.e(f"s_lshl_b32 s{S_T}, s{S_CUR}, 14")
                    self.e(f"s_add_u32 s{S_T}, s{S_T}, s{S_LDSDST}")
                    self.e(f"s_mov_b32 m0, s{S_T}")
                    self.e("s_nop 0")
                    if "spread" in f:      # one DMA instruction behind every third MFMA of this quad
                        pending = [f"global_load_lds_dwordx4 v{V_DMAOFF}, s[{S_T2 + 6}:{S_T2 + 7}] offset:{1024 * k}" for k in range(4)]
                        self.e(f"s_mov_b64 s[{S_T2 + 6}:{S_T2 + 7}], s[{S_WPTR}:{S_WPTR + 1}]")
                    else:
                        for k in range(4):
                            self.e(f"global_load_lds_dwordx4 v{V_DMAOFF}, s[{S_WPTR}:{S_WPTR + 1}] offset:{1024 * k}")
                    self.e(f"s_add_u32 s{S_WPTR}, s{S_WPTR}, {SLAB_BYTES}")
                    self.e(f"s_addc_u32 s{S_WPTR + 1}, s{S_WPTR + 1}, 0")
                    self.e(f"s_add_u32 s{S_IS}, s{S_IS}, 1")
                    self.e(f"s_cmp_eq_u32 s{S_IS}, {NSLABS}")
                    self.e(f"s_cselect_b64 s[{S_WPTR}:{S_WPTR + 1}], s[{S_WBASE}:{S_WBASE + 1}], s[{S_WPTR}:{S_WPTR + 1}]")
                    self.e(f"s_cselect_b32 s{S_IS}, 0, s{S_IS}")
                if "ring" in f:
                    self.e(f"s_add_u32 s{S_CUR}, s{S_CUR}, 1")
                    self.e(f"s_cmp_eq_u32 s{S_CUR}, {KRING}")
                    self.e(f"s_cselect_b32 s{S_CUR}, 0, s{S_CUR}")
                    self.e(f"s_lshl_b32 s{S_T2}, s{S_CUR}, 14")
                    self.e(f"v_add_u32 v{V_RING}, s{S_T2}, v{V_LANE16}")
            # A quad two ahead: quads 16, 17 are quads 0, 1 of the next slab (address already moved on at j == 14)
            tq = j + 2
            wset = V_W + 4 * (tq % 4)
            self.lds_op(f"ds_read_b128 v[{wset}:{wset + 3}], v{V_RING} offset:{1024 * (tq % 16)}", ("A", mb + tq // 16, tq % 16))
            if j == 8 and "bias" in f:    # next slab's bias (srcC of its first MFMAs), long after this slab's first MFMAs read it
                self.lds_op(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BIAS} offset:{bias_imm}", ("B", mb + 1))
            self.wait_lds(("A", mb, j))
            if j == 0 and "bias" in f:
                self.wait_lds(("B", mb))
            w = V_W + 4 * (j % 4)
            for jj in range(4):
                for g in range(G):
                    c = f"a[{A_BIAS}:{A_BIAS + 3}]" if (j == 0 and jj == 0 and "bias" in f) else self.acc(g, mb)
                    self.e(f"v_mfma_f32_16x16x4_f32 {self.acc(g, mb)}, v{w + jj}, v{V_IN + g * 64 + j * 4 + jj}, {c}")
                    if pending and (jj * G + g) % 3 == 0:
                        self.e(pending.pop(0))

    def relu(self):
        for mb in range(16):
            for g in range(G):
                b = A_ACC + (g * 16 + mb) * 4
                for r in range(4):
                    self.e(f"v_accvgpr_read_b32 v{V_IN + g * 64 + mb * 4 + r}, a{b + r}")
                for r in range(4):
                    v = V_IN + g * 64 + mb * 4 + r
                    self.e(f"v_max_f32 v{v}, 0, v{v}")

    def layer(self):
        # tags are relative to the slab index within this layer; reads of slab 16 quads 0/1 are the next layer's first quads
        for mb in range(16):
            self.slab(mb, (mb + 1) % 16 * 64)
        if "relu" in self.feat:
            self.e("s_nop 7")
            self.relu()


def make_body(feat):
    b = Body(feat)
    e = b.e
    # ---- set-up
    for i in range(192):
        e(f"v_mul_f32 v{V_IN + i}, {0.01 * (1 + i % 7):.2f}, %[seed]")
    for i in range(208):
        e(f"v_accvgpr_write_b32 a{i}, 0")
    e(f"v_mov_b32 v{V_LANE16}, %[lane16]")
    e(f"v_mov_b32 v{V_RING}, %[lane16]")
    e(f"v_mov_b32 v{V_DMAOFF}, %[dmaoff]")
    e(f"v_mov_b32 v{V_BIAS}, %[biasaddr]")
    e(f"s_mov_b32 s{S_LAYERS}, %[layers]")
    e(f"s_mov_b32 s{S_LDSDST}, %[ldsdst]")
    e(f"s_mov_b64 s[{S_WBASE}:{S_WBASE + 1}], %[wsrc]")
    e(f"s_mov_b64 s[{S_WPTR}:{S_WPTR + 1}], %[wsrc]")
    e(f"s_mov_b32 s{S_IS}, 0")
    e(f"s_mov_b32 s{S_CUR}, 0")
    if "dma" in feat:   # prime: steps 0..8 into buffers 0..8 (the consumer starts on buffer 0)
        for s in range(KRING):
            e(f"s_mov_b32 m0, s{S_LDSDST}" if s == 0 else f"s_add_u32 m0, m0, {SLAB_BYTES}")
            e("s_nop 0")
            for k in range(4):
                e(f"global_load_lds_dwordx4 v{V_DMAOFF}, s[{S_WPTR}:{S_WPTR + 1}] offset:{1024 * k}")
            e(f"s_add_u32 s{S_WPTR}, s{S_WPTR}, {SLAB_BYTES}")
            e(f"s_addc_u32 s{S_WPTR + 1}, s{S_WPTR + 1}, 0")
            e(f"s_add_u32 s{S_IS}, s{S_IS}, 1")
        e("s_waitcnt vmcnt(32)")
        e("s_barrier")
    # the first bias, complete before the loop (as it is at every later layer start), then the first two A quads
    if "bias" in feat:
        e(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BIAS}")
        e("s_waitcnt lgkmcnt(0)")
    b.lds_op(f"ds_read_b128 v[{V_W}:{V_W + 3}], v{V_RING}", ("A", 0, 0))
    b.lds_op(f"ds_read_b128 v[{V_W + 4}:{V_W + 7}], v{V_RING} offset:1024", ("A", 0, 1))
    e("s_memtime s[32:33]")
    e("1:")
    pre = list(b.lds)
    b.layer()
    # loop-carried LDS ops must be the same set as at loop entry (quads 0,1 + bias of the next layer's first slab)
    carried = [(t[0], t[1] - 16) + t[2:] for t in b.lds]
    assert carried == pre, (carried, pre)
    e(f"s_sub_u32 s{S_LAYERS}, s{S_LAYERS}, 1")
    e(f"s_cmp_lg_u32 s{S_LAYERS}, 0")
    e("s_cbranch_scc1 1b")
    e("s_memtime s[34:35]")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    e("s_sub_u32 %[tlo], s34, s32")
    e("s_subb_u32 %[thi], s35, s33")
    e("v_accvgpr_read_b32 %[res], a5")
    e(f"v_add_f32 %[res], %[res], v{V_IN + 17}")
    return b.L


VARIANTS = {
    "mfma+A": {"ring0"},
    "+bias": {"bias"},
    "+relu": {"bias", "relu"},
    "+barrier": {"bias", "relu", "barrier"},
    "+ring index": {"bias", "relu", "barrier", "ring"},
    "+LDS-DMA ring (all)": {"bias", "relu", "barrier", "ring", "dma"},
    "all but relu": {"bias", "barrier", "ring", "dma"},
    "all, DMA spread": {"bias", "relu", "barrier", "ring", "dma", "spread"},
}
clob = [f"v{r}" for r in range(0, 216)] + [f"a{r}" for r in range(0, 208)] + [f"s{r}" for r in range(20, 40)] + ["scc", "m0", "memory"]

src = ["// GENERATED by tools/ubench/gen_layer_asm.py -- do not edit.", "#include <hip/hip_runtime.h>", "#include <cstdio>",
       "typedef float f4 __attribute__((ext_vector_type(4)));",
       "__device__ inline float rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return (float)(int)x * (1.0f / 2147483648.0f); }", ""]
for n, (name, feat) in enumerate(VARIANTS.items()):
    text = "\n".join(f'      "{x}\\n\\t"' for x in make_body(feat))
    src.append(f"""__global__ __launch_bounds__(256) void k{n}(float* out, int layers, long long* clk, const float* wsrc) {{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* ring = reinterpret_cast<f4*>(smem);
  for (int i = threadIdx.x; i < 9 * 1024 + 520; i += 256)
    ring[i] = (f4){{rnd(i * 4 + 1) / 16, rnd(i * 4 + 2) / 16, rnd(i * 4 + 3) / 16, rnd(i * 4 + 4) / 16}};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float res; unsigned tlo, thi;
  asm volatile(
{text}
      : [res] "=v"(res), [tlo] "=s"(tlo), [thi] "=s"(thi)
      : [seed] "v"(rnd(threadIdx.x * 1000 + blockIdx.x * 7919)), [lane16] "v"(lane * 16), [dmaoff] "v"(wave * 4096 + lane * 16),
        [biasaddr] "v"(9 * 16384 + 16 * (lane >> 4)), [layers] "s"(layers), [ldsdst] "s"((unsigned)(unsigned long long)smem + wave * 4096),
        [wsrc] "s"(wsrc)
      : {", ".join('"%s"' % c for c in clob)});
  out[blockIdx.x * 256 + threadIdx.x] = res;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = ((long long)thi << 32) | tlo;
}}
""")
src.append("""template <typename K> void run(K kern, const char* name) {
  float* out; long long* clk; float* wsrc;
  hipMalloc(&clk, 16); hipMalloc(&out, sizeof(float) * 256 * 256); hipMalloc(&wsrc, 122 * 16384); hipMemset(wsrc, 0, 122 * 16384);
  const int lds = (9 * 1024 + 520) * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int layers = 70;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, 7, clk, wsrc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, layers, clk, wsrc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  double nm = layers * 3072.0;
  printf("%-28s %8.3f ms %7.2f TFLOP/s  %.2f ticks/MFMA  (%.0f ticks/layer)\\n", name, ms, 1024.0 * nm * 2048 / (ms * 1e-3) / 1e12, h / nm, (double)h / layers);
}
int main() {""")
for n, name in enumerate(VARIANTS):
    src.append(f'  run(k{n}, "{name}");')
src.append("  return 0;\n}")
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "layer_asm.hip"), "w").write("\n".join(src) + "\n")
