"""Generates render16_body*.inc: the body of s2l::render16_tiles_kernel (csrc/render16.hip) -- the OPT-IN split-half speed mode of
the clip renderer (`s2l_render_lip_split`, TalkingFace.render_clip(precision="split")).  The exact fp32 kernel
(gen_render_body.py) stays the default and the headline.

Arithmetic.  Every fp32 operand x of the seven 256x256 layers and of the output layer is carried as two IEEE half numbers,
hi = f16(x) and lo = f16(x - hi) (x - hi is exact in fp32), and a product is evaluated as  W_lo a_hi + W_hi a_lo + W_hi a_hi  on
v_mfma_f32_16x16x32_f16 with fp32 accumulation, smallest terms first; the dropped lo x lo term is below 2^-20 of the product.
Activations are split with round-toward-zero (v_cvt_pkrtz_f16_f32: |x - hi - lo| < 2^-20 |x|, and a value beyond the half range
saturates instead of becoming inf), weights at pack time with round-to-nearest (2^-22).  Half subnormals are honoured by the
MFMA (measured: with them flushed the W_lo terms of every weight below 0.12 would vanish and the error would be ~1e-4; it is
1.4e-6), so the only range condition is |pre-activation| < 65504 -- the exact kernel has none.
Why half and not bf16: the same MFMA rate, and 11 + 11 instead of 8 + 8 significant bits -- measured RMSE against the CPU oracle
1.4e-6 (117 dB; the exact kernel: 6.5e-7) instead of 1.0e-5 (100 dB) with bf16 parts (`S2L_RENDER16_HALF=bf16`, an A/B build).
A 16x16x32 half MFMA does 8 x the work of a 16x16x4 fp32 MFMA in half its cycles, so three of them per product need 3/16 of
the fp32 kernel's matrix-pipe time.  The first layer and the skip half of pts_linears[5] stay exact fp32 sums of table rows
(h0 = relu(p0 + q0), h5 = relu(W5b h4 + p5 + q5)), as in the fp32 kernel; biases enter as srcC in fp32.

What is shared with the fp32 body (this module subclasses its `Body`): the tile shapes, the 117-step LDS-DMA ring of 16-KiB steps
and its protocol (one barrier per step, requests 8 steps ahead across tile boundaries), the q / p table steps, the output store.
A weight slab is again 16 KiB: 16 output features x K = 256 as [k-step 8][hi | lo][lane 64][8 halves] -- `s2l_pack_render16`
derives it from the fp32 slab ([2 s + e / 4][lane][e % 4] -> [s][part][lane][e]), so the A operand of k-step s pairs with the B
operand built from accumulator blocks 2 s and 2 s + 1 of the previous layer: activations still never leave the register file.

What differs: a slab is 8 k-steps x (2 A reads into AGPRs + 9 MFMAs: 3 products x 3 sample groups); between two layers the
wave converts its 192 accumulators: per pair of values  2 v_accvgpr_read, 2 v_max (ReLU), v_cvt_pkrtz_f16_f32 (hi),
2 v_cvt_f32_f16 (hi back to fp32), 2 v_sub (exact), v_cvt_pkrtz_f16_f32 (lo) -- 960 VALU instructions per layer and wave next
to 1152 MFMAs of 16 cycles.  Measured (MI355X, 96x96 x 1000 frames): 16.1 ms = 62 k frames/s, 3.5 x the exact kernel.

Can the conversions hide behind the MFMAs?  Measured with `S2L_RENDER16_FILLER=n` (n dummy VALU instructions behind every MFMA;
16.2 ms without): n = 1: 17.8 ms, 2: 18.9, 3: 20.5 -- a filler costs ~2.2 cycles there against ~4.5 in the block between two
layers, i.e. hiding halves its price, it does not remove it; with the B registers live until a layer's last slab the only
hideable form (pack the accumulators in place in AGPRs behind the following slabs, move 192 registers at the layer's end) adds
384 moves per layer and comes out ~15 % of the conversion time ahead: not built.  PMC (profiles/r04a): matrix pipe busy 0.70,
2.16 GHz under this load; the MFMAs issue back to back at 16 cycles inside the slabs.

Register map (per wave):  v0-191 B operands: group g, k-step s: hi v[64 g + 8 s .. + 3], lo v[64 g + 8 s + 4 .. + 7]
                          (the same 192 registers hold fp32 values, one feature each, while a table step is being added)
                          v192-207 temporaries | v208-223 addresses | v224-240 ring addresses of a layer
                          a0-191 accumulators | a192-195 bias / srcC | a196-207 output accumulators | a208-239 four A sets (hi, lo)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_render_body as R
from gen_render_body import (A_ACC, A_BIAS, A_RGB, KRING, NLAYERS, NSLABS, SLAB, V_BIAS, V_BIAS0, V_BOUT, V_DMAOFF, V_IN, V_LANE16, V_PIX,
                             V_PIXOFF, V_QADDR, V_RA, V_SCR, V_TMP, V_W, s, s2, S)

A_W = 208                 # four rotating A-operand sets of 8 AGPRs: [hi 4 | lo 4]
A_LAST = A_W + 31
V_LAST = R.V_LAST
KSTEPS = 8                # k-steps of 32 per slab (K = 256)
HALF = os.environ.get("S2L_RENDER16_HALF", "f16")       # "f16" (default) or "bf16" (A/B builds: compile render16.hip with -DS2L_RENDER16_BF16)
MFMA = f"v_mfma_f32_16x16x32_{HALF}"
FILLER = int(os.environ.get("S2L_RENDER16_FILLER", "0"))
FILLER_KIND = os.environ.get("S2L_RENDER16_FILLER_KIND", "max")


class Body16(R.Body):
    # ---- registers
    @staticmethod
    def bhi(g, ks):
        return V_IN + g * 64 + ks * 8

    @staticmethod
    def blo(g, ks):
        return V_IN + g * 64 + ks * 8 + 4

    @staticmethod
    def aset(ks):
        return A_W + 8 * (ks % 4)

    # ---- one slab: 8 k-steps of (A_hi, A_lo) x 3 products x G groups
    def kstep_mfmas(self, ks, dst, first_c=None, sprinkle=None):
        """3 G MFMAs of k-step ks, smallest terms first: W_lo a_hi, W_hi a_lo, W_hi a_hi.  sprinkle[n]: instructions tucked
        behind MFMA n (positions are written for G = 3: nine slots)."""
        G = R.G
        if G != 3 and sprinkle:
            packed = {}
            for pos in sorted(sprinkle):
                packed.setdefault(pos * G // 3, []).extend(sprinkle[pos])
            sprinkle = packed
        sprinkle = dict(sprinkle or {})
        a = self.aset(ks)
        whi, wlo = f"a[{a}:{a + 3}]", f"a[{a + 4}:{a + 7}]"
        n = 0
        for part, (w, breg) in enumerate(((wlo, self.bhi), (whi, self.blo), (whi, self.bhi))):
            for g in range(G):
                b = breg(g, ks)
                c = first_c if (first_c and ks == 0 and part == 0) else dst(g)
                self.e(f"{MFMA} {dst(g)}, {w}, v[{b}:{b + 3}], {c}")
                for _ in range(FILLER):      # experiment: dummy VALU instructions behind every MFMA (what does a filler cost?)
                    self.e(f"v_cvt_pkrtz_f16_f32 v{V_W + 15}, v{V_W + 14}, v{V_W + 13}" if FILLER_KIND == "cvt" else
                           f"v_accvgpr_read_b32 v{V_W + 15}, a{A_BIAS}" if FILLER_KIND == "acc" else f"v_max_f32 v{V_W + 15}, 0, v{V_W + 14}")
                for t in sprinkle.pop(n, []):
                    if isinstance(t, tuple):
                        self.lds_op(t[1], t[2])
                    else:
                        self.e(t)
                n += 1
        assert not sprinkle, sprinkle

    def a_read(self, slab_tag, tks, areg):
        """A operands of k-step tks (counted from `slab_tag`'s first: tks >= 8 is the next slab) into set tks % 4"""
        a = self.aset(tks)
        off = 2048 * (tks % KSTEPS)
        tag = slab_tag + tks // KSTEPS
        self.lds_op(f"ds_read_b128 a[{a}:{a + 3}], v{areg} offset:{off}", ("A", tag, tks % KSTEPS, 0))
        self.lds_op(f"ds_read_b128 a[{a + 4}:{a + 7}], v{areg} offset:{off + 1024}", ("A", tag, tks % KSTEPS, 1))

    def slab(self, mb, special=None):
        e = self.e
        dst = lambda g: self.acc(g, mb)
        for ks in range(KSTEPS):
            sprinkle = {}
            if ks == 5:     # the scalar side of `advance`, ahead of the barrier and behind MFMAs
                pre = [f"s_add_u32 {s('T')}, {s('CURB')}, {s('LDSBASE')}", f"s_add_u32 {s('CUR')}, {s('CUR')}, 1",
                       f"s_cmp_eq_u32 {s('CUR')}, {KRING}", f"s_cselect_b32 {s('CUR')}, 0, {s('CUR')}",
                       f"s_lshl_b32 {s('CURB')}, {s('CUR')}, 14", f"s_mov_b32 m0, {s('T')}", f"s_mov_b64 {s2('T4')}, {s2('WPTR')}"]
                sprinkle = {1 + n: [t] for n, t in enumerate(pre)}
            if ks == 6:
                self.wait_lds(("A", mb, 7, 1))      # both remaining k-steps of this slab are in registers
                e("s_waitcnt vmcnt(28)")
                e("s_barrier")
            self.a_read(mb, ks + 2, V_RA + mb + (ks + 2) // KSTEPS)
            if ks == 4:     # next slab's bias (srcC of its first MFMAs)
                self.lds_op(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BIAS} offset:{(mb + 1) * 64}", ("B", mb + 1))
            self.wait_lds(("A", mb, ks, 1))
            if ks == 0:
                self.wait_lds(("B", mb))
            if ks == 6:
                join = self.label("join")
                for layer, emit in (special or []):
                    ol = self.label("table")
                    e(f"s_cmp_eq_u32 {s('LAYER')}, {layer}")
                    e(f"s_cbranch_scc1 {ol}")
                    self.outofline.append((ol, join, emit, mb, ks))
                dma = self.dma4()
                ptr = [f"s_add_u32 {s('WPTR')}, {s('WPTR')}, {SLAB}", f"s_addc_u32 {s('WPTR1')}, {s('WPTR1')}, 0",
                       f"s_add_u32 {s('IS')}, {s('IS')}, 1", f"s_cmp_eq_u32 {s('IS')}, {NSLABS}",
                       f"s_cselect_b64 {s2('WPTR')}, {s2('WBASE')}, {s2('WPTR')}", f"s_cselect_b32 {s('IS')}, 0, {s('IS')}"]
                sprinkle = {0: [dma[0]], 2: [dma[1]], 4: [dma[2]], 6: [dma[3], *ptr]}
            self.kstep_mfmas(ks, dst, first_c=f"a[{A_BIAS}:{A_BIAS + 3}]", sprinkle=sprinkle)
            if ks == 6:
                e(f"{join}:")

    # ---- between two layers
    def after_mfma_nops(self):
        self.e("s_nop 7")
        self.e("s_nop 3")

    def convert_pair(self, x0, x1, hi, lo, t):
        """(x0, x1) fp32 VGPRs (clobbered) -> packed bf16 pairs hi, lo; t: two temporaries"""
        e = self.e
        if HALF == "f16":
            e(f"v_cvt_pkrtz_f16_f32 v{hi}, v{x0}, v{x1}")
            e(f"v_cvt_f32_f16 v{t}, v{hi}")
            e(f"v_cvt_f32_f16_sdwa v{t + 1}, v{hi} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
            e(f"v_sub_f32 v{x0}, v{x0}, v{t}")
            e(f"v_sub_f32 v{x1}, v{x1}, v{t + 1}")
            e(f"v_cvt_pkrtz_f16_f32 v{lo}, v{x0}, v{x1}")
            return
        e(f"v_cvt_pk_bf16_f32 v{hi}, v{x0}, v{x1}")
        e(f"v_lshlrev_b32 v{t}, 16, v{hi}")
        e(f"v_and_b32 v{t + 1}, 0xffff0000, v{hi}")
        e(f"v_sub_f32 v{x0}, v{x0}, v{t}")
        e(f"v_sub_f32 v{x1}, v{x1}, v{t + 1}")
        e(f"v_cvt_pk_bf16_f32 v{lo}, v{x0}, v{x1}")

    def relu_all(self):
        """accumulators -> ReLU -> (hi, lo) B operands of the next layer.  Pair p of k-step ks: block 2 ks + p / 2, registers
        2 (p % 2), + 1 -> dword p of the k-step's hi and lo operands."""
        e = self.e
        self.wait_all_lds()
        self.after_mfma_nops()
        n = 0
        for g in range(R.G):
            for ks in range(KSTEPS):
                for p in range(4):
                    a0 = A_ACC + (g * 16 + 2 * ks + p // 2) * 4 + 2 * (p % 2)
                    t = V_W + 4 * (n % 4)
                    n += 1
                    e(f"v_accvgpr_read_b32 v{t}, a{a0}")
                    e(f"v_accvgpr_read_b32 v{t + 1}, a{a0 + 1}")
                    e(f"v_max_f32 v{t}, 0, v{t}")
                    e(f"v_max_f32 v{t + 1}, 0, v{t + 1}")
                    self.convert_pair(t, t + 1, self.bhi(g, ks) + p, self.blo(g, ks) + p, t + 2)

    def last_block_to_b_registers(self):
        """the skip layer's table steps work on fp32 values, one feature per B register: in[g][4 mb + r] = acc[g][mb][r]"""
        e = self.e
        self.wait_all_lds()
        self.after_mfma_nops()
        for g in range(R.G):
            for k in range(64):
                e(f"v_accvgpr_read_b32 v{self.inreg(g, k)}, a{A_ACC + g * 64 + k}")

    def convert_in_place(self):
        """the 192 B registers hold fp32 activations (already ReLU'd by the p step), register 64 g + 4 mb + r = feature
        16 mb + 4 q + r: turn each run of 8 (blocks 2 ks, 2 ks + 1) into the k-step's hi (first 4) and lo (last 4) operands"""
        e = self.e
        for g in range(R.G):
            for ks in range(KSTEPS):
                base = V_IN + g * 64 + ks * 8
                for p in range(4):
                    # hi -> v[V_W + p], lo -> v[V_W + 4 + p], work registers v[V_W + 8 + 2 (p % 2)], + 1
                    self.convert_pair(base + 2 * p, base + 2 * p + 1, V_W + p, V_W + 4 + p, V_W + 8 + 2 * (p % 2))
                for i in range(8):
                    e(f"v_mov_b32 v{base + i}, v{V_W + i}")

    def prefetch_first_quads(self, with_bias):
        self.ring_table()
        if with_bias:
            self.e(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BIAS}")
            self.wait_all_lds()
        self.a_read(0, 0, V_RA)
        self.a_read(0, 1, V_RA)


def generate():
    G, PGT, FT = R.G, R.PGT, R.FT
    b = Body16()
    b.outofline = []
    e = b.e
    # ================= prologue: operands -> owned registers (as the fp32 body)
    for dst, src in (("LDSBASE", "ldsbase"), ("NFG", "nfg"), ("TILE", "tile0"), ("TEND", "tile_end"), ("NFRAMES", "nframes"), ("HW", "hw"),
                     ("FG", "fg0"), ("PG", "pg0"), ("WAVE", "wave")) + ((("NPGM1", "npgm1"),) if PGT > 1 else ()):
        e(f"s_mov_b32 {s(dst)}, %[{src}]")
    for dst, src in (("WBASE", "wsrc"), ("Q0", "q0"), ("Q5", "q5"), ("P0", "p0"), ("P5", "p5"), ("OUT", "out")):
        e(f"s_mov_b64 {s2(dst)}, %[{src}]")
    e(f"s_mov_b64 {s2('WPTR')}, {s2('WBASE')}")
    e(f"s_sub_u32 {s('NFM1')}, {s('NFRAMES')}, 1")
    e(f"s_mov_b32 {s('IS')}, 0")
    e(f"s_mov_b32 {s('CUR')}, 0")
    e(f"s_mov_b32 {s('CURB')}, 0")
    for dst, src in ((V_LANE16, "lane16"), (V_DMAOFF, "dmaoff"), (V_BIAS0, "biasaddr"), (V_QADDR, "qaddr"), (V_BOUT, "boutaddr"), (V_PIX, "px")):
        e(f"v_mov_b32 v{dst}, %[{src}]")

    def prime(st):
        if st == 0:
            e(f"s_mov_b32 {s('T')}, {s('LDSBASE')}")
        else:
            e(f"s_add_u32 {s('T')}, {s('LDSBASE')}, {SLAB * st}")
        emit = b.refill_at(st - KRING)
        if emit is not None:
            emit()
        else:
            b.weight_refill_setup()
            e("s_nop 0")
            for t in b.dma4():
                e(t)
    for st in range(KRING - 1):
        prime(st)
    e("s_waitcnt lgkmcnt(0)")
    e("s_waitcnt vmcnt(28)")
    e("s_barrier")
    prime(KRING - 1)

    # ================= tile loop
    e("S2L_TILE:")
    e(f"s_add_u32 {s('FGN')}, {s('FG')}, 1")
    e(f"s_cmp_eq_u32 {s('FGN')}, {s('NFG')}")
    e(f"s_cselect_b32 {s('FGN')}, 0, {s('FGN')}")
    e(f"s_cselect_b32 {s('T6')}, 1, 0")
    e(f"s_add_u32 {s('PGN')}, {s('PG')}, {s('T6')}")
    e(f"s_add_u32 {s('T7')}, {s('TILE')}, 1")
    e(f"s_cmp_lt_u32 {s('T7')}, {s('TEND')}")
    e(f"s_cselect_b32 {s('FGN')}, {s('FGN')}, {s('FG')}")
    e(f"s_cselect_b32 {s('PGN')}, {s('PGN')}, {s('PG')}")
    # ---- h0 = relu(p0[pixel] + q0[frame]) in fp32, then split
    b.q_step(False)
    b.advance_and_refill(0)
    for k in range(PGT):
        b.p_step(k)
        b.advance_and_refill(1 + k)
    b.convert_in_place()
    e(f"v_mov_b32 v{V_BIAS}, v{V_BIAS0}")
    e(f"s_mov_b32 {s('LAYER')}, 0")
    b.prefetch_first_quads(True)
    loop_state = list(b.lds)

    # ---- the layer loop
    e("S2L_LAYER:")
    base1 = 1 + PGT
    base2 = base1 + 81 + PGT
    special = {}
    for L in range(NLAYERS):
        for mb in range(16):
            emit = b.refill_at((base1 + 16 * L if L < 5 else base2 + 16 * (L - 5)) + mb)
            if emit is not None:
                special.setdefault(mb, []).append((L, emit))
    for mb in range(16):
        b.slab(mb, special.get(mb))
    slab_end = list(b.lds)
    assert [t[:3] for t in slab_end if t[0] == "A"] == [("A", 16, 0), ("A", 16, 0), ("A", 16, 1), ("A", 16, 1)], slab_end
    e(f"s_cmp_eq_u32 {s('LAYER')}, 4")
    e("s_cbranch_scc1 S2L_SKIP")
    b.relu_all()
    b.ring_table()
    b.lds = list(loop_state)       # (complete, which is stronger than "in flight")
    e(f"s_add_u32 {s('LAYER')}, {s('LAYER')}, 1")
    e(f"v_add_u32 v{V_BIAS}, 1024, v{V_BIAS}")
    e(f"s_cmp_lt_u32 {s('LAYER')}, {NLAYERS}")
    e("s_cbranch_scc1 S2L_LAYER")

    # ---- output layer (3 rows padded to one M-block), no activation; k-steps 0, 1 of its slab are in flight
    b.lds = list(loop_state)
    b.lds_op(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BOUT}", ("B", 0))
    rgb = lambda g: f"a[{A_RGB + 4 * g}:{A_RGB + 4 * g + 3}]"
    for ks in range(KSTEPS):
        if ks + 2 < KSTEPS:
            b.a_read(0, ks + 2, V_RA)
        b.wait_lds(("A", 0, ks, 1))
        if ks == 0:
            b.wait_lds(("B", 0))
        b.kstep_mfmas(ks, rgb, first_c=f"a[{A_BIAS}:{A_BIAS + 3}]")
    b.wait_all_lds()
    b.advance_and_refill(len(b.tile_steps()) - 1)
    b.after_mfma_nops()            # (the stores read the output accumulators)
    if PGT == 1:
        e(f"s_lshl_b32 {s('T6')}, {s('PG')}, 4")
        e(f"v_add_u32 v{V_TMP}, {s('T6')}, v{V_PIX}")
        e(f"v_cmp_gt_u32 vcc, {s('HW')}, v{V_TMP}")
        e(f"v_mul_u32_u24 v{V_PIXOFF}, 12, v{V_TMP}")
        e(f"s_mov_b64 {s2('EX')}, exec")
        e("s_nop 3")
        e("s_and_b64 exec, vcc, 0xffff")
        e(f"s_mul_i32 {s('T6')}, {s('FG')}, 12")
        e(f"s_mul_i32 {s('T7')}, {s('WAVE')}, {G}")
        e(f"s_add_u32 {s('T6')}, {s('T6')}, {s('T7')}")      # frame0
        e("s_nop 7")
        e("s_nop 7")
    else:
        e(f"s_mov_b64 {s2('EX')}, exec")
        e(f"s_mov_b32 {s('T6')}, {s('FG')}")
    for g in range(G):
        skip = b.label("nostore")
        if PGT > 1:
            e(f"s_mul_i32 {s('T8')}, {s('PG')}, {PGT}")
            e(f"s_mul_i32 {s('T9')}, {s('WAVE')}, {G}")
            e(f"s_add_u32 {s('T8')}, {s('T8')}, {s('T9')}")
            e(f"s_add_u32 {s('T8')}, {s('T8')}, {g}")
            e(f"s_lshl_b32 {s('T8')}, {s('T8')}, 4")
            e(f"v_add_u32 v{V_TMP}, {s('T8')}, v{V_PIX}")
            e(f"v_cmp_gt_u32 vcc, {s('HW')}, v{V_TMP}")
            e(f"v_mul_u32_u24 v{V_PIXOFF}, 12, v{V_TMP}")
            e("s_nop 3")
            e("s_and_b64 exec, vcc, 0xffff")
            e("s_nop 7")
            e("s_nop 7")
        e(f"s_add_u32 {s('T7')}, {s('T6')}, {g if PGT == 1 else 0}")
        e(f"s_cmp_ge_u32 {s('T7')}, {s('NFRAMES')}")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_mul_hi_u32 {s('T9')}, {s('T7')}, {s('HW')}")
        e(f"s_mul_i32 {s('T8')}, {s('T7')}, {s('HW')}")
        e(f"s_mul_i32 {s('T9')}, {s('T9')}, 12")
        e(f"s_mul_hi_u32 {s('T7')}, {s('T8')}, 12")
        e(f"s_mul_i32 {s('T8')}, {s('T8')}, 12")
        e(f"s_add_u32 {s('T9')}, {s('T9')}, {s('T7')}")
        e(f"s_add_u32 {s('T4')}, {s('OUT')}, {s('T8')}")
        e(f"s_addc_u32 {s('T5')}, {s('OUT1')}, {s('T9')}")
        e(f"global_store_dwordx3 v{V_PIXOFF}, a[{A_RGB + 4 * g}:{A_RGB + 4 * g + 2}], {s2('T4')}")
        e(f"{skip}:")
        if PGT > 1:
            e(f"s_mov_b64 exec, {s2('EX')}")
    e(f"s_mov_b64 exec, {s2('EX')}")
    # ---- next tile of this workgroup
    e(f"s_add_u32 {s('TILE')}, {s('TILE')}, 1")
    e(f"s_mov_b32 {s('FG')}, {s('FGN')}")
    e(f"s_mov_b32 {s('PG')}, {s('PGN')}")
    e(f"s_cmp_lt_u32 {s('TILE')}, {s('TEND')}")
    e("s_cbranch_scc1 S2L_TILE")
    e("s_waitcnt vmcnt(0)")
    e("s_branch S2L_END")

    # ================= out of line: pts_linears[5] on cat([skip, h4]): + q5[frame] + p5[pixel], ReLU, split
    e("S2L_SKIP:")
    b.lds = list(slab_end)     # (the k-steps read from the q5 step are discarded)
    b.q_step(True)
    b.advance_and_refill(base1 + 80)
    for k in range(PGT):
        b.p_step(k)
        b.advance_and_refill(base1 + 81 + k)
    b.convert_in_place()
    e(f"s_mov_b32 {s('LAYER')}, 5")
    e(f"v_add_u32 v{V_BIAS}, 1024, v{V_BIAS}")
    b.prefetch_first_quads(False)
    assert b.lds == loop_state
    e("s_branch S2L_LAYER")
    # ================= out of line: the table refills of a tile (the slab's k-step 6 is repeated here without DMA tucks)
    for ol, join, emit, mb, ks in b.outofline:
        e(f"{ol}:")
        emit()
        b.kstep_mfmas(ks, lambda g: b.acc(g, mb))
        e(f"s_branch {join}")
    e("S2L_END:")
    return [x for x in b.L if x is not None]


OPERANDS = """      :
      : [ldsbase] "s"(ldsbase), [nfg] "s"(nfg), [tile0] "s"(tile0), [tile_end] "s"(tile_end), [nframes] "s"(a.nframes), [hw] "s"(a.hw),
        [fg0] "s"(fg0), [pg0] "s"(pg0),NPGM1_OPERAND
        [wave] "s"(wave), [wsrc] "s"(wsrc), [q0] "s"(a.q0), [q5] "s"(a.q5), [p0] "s"(a.p0t), [p5] "s"(a.p5t), [out] "s"(a.out),
        [lane16] "v"(lane16), [dmaoff] "v"(dmaoff), [biasaddr] "v"(biasaddr), [qaddr] "v"(qaddr), [boutaddr] "v"(boutaddr), [px] "v"(px)
"""


def main(path, variant="long"):
    R.set_variant(variant)
    lines = generate()
    tag = "S2LH" + variant[0].upper() + "_"
    lines = [x.replace("S2L_", tag) for x in lines]
    s_last = R.S_LAST - (1 if R.PGT == 1 else 0)
    clob = [f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)] + [f"s{r}" for r in range(36, s_last + 1)]
    clob += ["vcc", "scc", "memory"]
    out = ["// GENERATED by csrc/gen_render16_body.py -- do not edit; the generator is the source.", "asm volatile("]
    out += [f'    "{x}\\n\\t"' for x in lines]
    out.append(OPERANDS.rstrip("\n").replace("NPGM1_OPERAND", ' [npgm1] "s"(npgm1),' if R.PGT > 1 else ""))
    out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return len(lines)


def main_all(objdir):
    n = {}
    for variant in R.VARIANTS:
        n[variant] = main(os.path.join(objdir, f"render16_body_{variant}.inc"), variant)
    R.set_variant("long")
    return n


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    for k, v in main_all(d).items():
        print(f"render16 body, {k}: {v} instructions")
