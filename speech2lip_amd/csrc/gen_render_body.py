"""Generates render_body.inc: the body of s2l::render_tiles_kernel (csrc/render.hip) as ONE fixed-register gfx950 assembly
text (build.py runs this before hipcc; the kernel `#include`s the result inside an `asm volatile`).

Why assembly: the renderer runs one wave per SIMD (512 registers), and on CDNA4 a wave's VALU instructions do NOT overlap its
own MFMAs -- every VALU instruction between two MFMAs costs its 4 cycles plus a ~9-cycle pipe switch
(tools/ubench/gen_mfma_shadow.py), while LDS reads, LDS-DMA issues and SALU are free.  So the layer body must contain nothing
but MFMAs, ds_reads and scalar code, with the few unavoidable VALU instructions (the v_max of the ReLU, the 17 ring addresses of a
layer) bunched between two layers; even the accumulator -> B-register move of the ReLU goes through LDS (ds_write from AGPRs,
ds_read into VGPRs) because that is free and v_accvgpr_read is not.  hipcc's version of the same loop spent 106.9 k cycles per layer; this
schedule spends 99.95 k (tools/trace_tiles.py; tools/ubench/gen_layer_asm.py is the same loop as a stand-alone benchmark), the
MFMAs alone 98.3 k.  The arithmetic -- operands, accumulation order, roundings -- is that of the C++ kernel it replaced: frames
are bit-identical (tools/ab_render.py).

Structure (see render.hip for the tile / ring / table description): 117 ring steps per tile; a layer = 16 slabs of
16 k-quads x 4 k-steps x 3 sample groups = 3072 MFMAs; A quads come from the ring two quads ahead into four rotating
register sets; the slab's bias is read straight into AGPRs and enters as srcC of the first three MFMAs; the scalar half of
the ring protocol runs behind the MFMAs of k-quad 13, then lgkmcnt / vmcnt / s_barrier retire the slab's buffer and four
LDS-DMA loads refill it from behind the MFMAs of k-quad 14.  The four table refills per tile and the q5/p5 step of
pts_linears[5] are out of line, behind scalar branches on the layer counter.  The generator tracks every LDS operation in
flight and derives each s_waitcnt lgkmcnt(n) from that list; it asserts that all paths into the layer loop carry the same list.

Register map (per wave):  v0-191 B operands in[g][k] | v192-207 four A-quad sets | v208.. addresses and temporaries
                          a0-191 accumulators acc[g][mb][r] | a192-195 bias / srcC | a196-207 output-layer accumulators
                          s36.. scalar state (ring position, weight-slab pointer, tile coordinates, table pointers).

Tile shapes (VARIANTS below).  A tile is 4 waves x G groups of 16 samples, arranged as PGT pixel groups x FT frames:
  long   G = 3, 1 x 12:   wave w, group g = frame 3 w + g of ONE pixel group: one p0 / p5 ring step per tile.  The shape for clips;
                          its generated text is pinned (frames of a clip of F = 12 k frames are rendered by exactly this code).
  wide   G = 3, 12 x 1:   group = pixel group 3 w + g of ONE frame: no frame slot is ever wasted, at the price of 12 p0 + 12 p5 ring
                          steps per tile (+19 % ring traffic, +3 % cycles).  Short clips whose length is not a multiple of 12.
  single G = 1, 4 x 1:    64 samples per tile: a 96x96 frame is 144 tiles instead of 48, i.e. the chip's 256 CUs can all work on
                          ONE frame -- the reference's own operating mode (one frame per call, inference.py:129,140).
Every sample column sees the same MFMA sequence in every shape, so frames are bit-identical whichever shape rendered them.
A ring step j+9 is requested by the `advance` that retires step j; `Body.refill_at` maps a position in the tile's step sequence
(q0, PGT x p0, 80 slabs, q5, PGT x p5, 33 slabs) to what that request must be."""
import os
import sys

G, PGT, FT = 3, 1, 12          # set_variant() below
KRING, SLAB, NSLABS, NLAYERS = 9, 16384, 113, 7
VARIANTS = {"long": (3, 1, 12), "wide": (3, 12, 1), "single": (1, 4, 1)}


def set_variant(name):
    global G, PGT, FT
    G, PGT, FT = VARIANTS[name]
    assert 4 * G == PGT * FT and (FT == 1 or PGT == 1)

# ---- vector registers
V_IN, V_W = 0, 192
V_LANE16, V_RING, V_DMAOFF, V_BIAS, V_QADDR, V_TMP, V_BOUT, V_PIXOFF = 208, 209, 210, 211, 212, 213, 214, 215
V_T0, V_BIAS0, V_PIX, V_SCR = 216, 220, 221, 222   # V_T0..V_T0+3: one f4 of temporaries; V_SCR: this wave's 1 KiB LDS scratch
V_RA = 224                                      # 17 ring addresses of a layer: slabs 0..15 and the slab after
V_LAST = 240
A_ACC, A_BIAS, A_RGB = 0, 192, 196
A_LAST = 207
# ---- scalar registers (all owned by the body; the compiler's operands live elsewhere)
def _scalar_map(first, singles, pairs):
    """names -> SGPR numbers from `first` on; a pair NAME / NAME1 starts on an even register"""
    m, r = {}, first
    for n in pairs:
        r += r & 1
        m[n], m[n + "1"] = r, r + 1
        r += 2
    for n in singles:
        m[n] = r
        r += 1
    return m


S = _scalar_map(36,
                singles="CUR T LDSBASE CURB IS LAYER FG PG NFG TILE TEND NFM1 HW FGN PGN WAVE NFRAMES NPGM1".split(),
                pairs=("WPTR", "WBASE", "Q0", "Q5", "P0", "P5", "OUT", "T4", "T6", "T8", "EX", "TRACE"))
S.update(T5=S["T41"], T7=S["T61"], T9=S["T81"])          # the halves of the temporary pairs by their own names
TRACE = os.environ.get("S2L_RENDER_TRACE") == "1"      # experiment builds only (tools/trace_tiles.py): per-tile phase timestamps
S_LAST = max(S.values())


def s(n):
    return f"s{S[n]}"


def s2(n):
    return f"s[{S[n]}:{S[n] + 1}]"


class Body:
    def __init__(self):
        self.L, self.lds, self.nlabel = [], [], 0

    def e(self, t):
        self.L.append(t)

    def label(self, stem):
        self.nlabel += 1
        return f"S2L_{stem}_{self.nlabel}"

    # ---- LDS bookkeeping: ops return in order, lgkmcnt(n) = "at most n still out"
    def lds_op(self, text, tag):
        self.e(text)
        self.lds.append(tag)

    def wait_lds(self, tag):
        if tag not in self.lds:
            return
        newer = len(self.lds) - 1 - self.lds.index(tag)
        assert newer <= 15, "lgkmcnt is a 4-bit counter"
        self.e(f"s_waitcnt lgkmcnt({newer})")
        self.lds = self.lds[len(self.lds) - newer:] if newer else []

    def wait_all_lds(self):
        self.e("s_waitcnt lgkmcnt(0)")
        self.lds = []

    def trace(self, slot):
        """experiment builds: wave 0 lane 0 stores s_memtime to trace[tile*16 + slot] (slot: int, or 'layer' = 2 + LAYER)"""
        if not TRACE:
            return
        e = self.e
        skip = self.label("notrace")
        e(f"s_memtime {s2('T8')}")
        self.wait_all_lds()
        e(f"s_cmp_eq_u64 {s2('TRACE')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_cmp_eq_u32 {s('WAVE')}, 0")
        e(f"s_cbranch_scc0 {skip}")
        e(f"s_lshl_b32 {s('T6')}, {s('TILE')}, 4")
        e(f"s_add_u32 {s('T6')}, {s('T6')}, {slot}" if isinstance(slot, int) else f"s_add_u32 {s('T6')}, {s('T6')}, {s('LAYER')}")
        if not isinstance(slot, int):
            e(f"s_add_u32 {s('T6')}, {s('T6')}, 2")
        e(f"s_mov_b32 {s('T7')}, 0")
        e(f"s_lshl_b64 {s2('T6')}, {s2('T6')}, 3")
        e(f"s_add_u32 {s('T4')}, {s('TRACE')}, {s('T6')}")
        e(f"s_addc_u32 {s('T5')}, {s('TRACE1')}, {s('T7')}")
        e(f"v_mov_b32 v{V_T0}, {s('T8')}")
        e(f"v_mov_b32 v{V_T0 + 1}, {s('T9')}")
        e(f"v_mov_b32 v{V_T0 + 2}, 0")
        e(f"s_mov_b64 {s2('EX')}, exec")
        e("s_mov_b64 exec, 1")
        e(f"global_store_dwordx2 v{V_T0 + 2}, v[{V_T0}:{V_T0 + 1}], {s2('T4')}")
        e(f"s_mov_b64 exec, {s2('EX')}")
        e(f"{skip}:")

    # ---- registers
    @staticmethod
    def acc(g, mb):
        b = A_ACC + (g * 16 + mb) * 4
        return f"a[{b}:{b + 3}]"

    @staticmethod
    def inreg(g, k):
        return V_IN + g * 64 + k

    # ---- ring protocol
    def advance(self):
        """Publish the next step, retire the current one: s[T] = LDS address of the retired buffer's quarter owned by this
        wave, CUR / CURB move on (callers derive the vector addresses they need).  Precondition: this wave's LDS reads of the current step have returned."""
        e = self.e
        e("s_waitcnt vmcnt(28)")
        e("s_barrier")
        e(f"s_add_u32 {s('T')}, {s('CURB')}, {s('LDSBASE')}")
        e(f"s_add_u32 {s('CUR')}, {s('CUR')}, 1")
        e(f"s_cmp_eq_u32 {s('CUR')}, {KRING}")
        e(f"s_cselect_b32 {s('CUR')}, 0, {s('CUR')}")
        e(f"s_lshl_b32 {s('CURB')}, {s('CUR')}, 14")

    def weight_refill_setup(self):
        """m0 / s[T4:T5] for the four loads of the next weight slab into the retired buffer; moves the slab pointer on."""
        e = self.e
        e(f"s_mov_b32 m0, {s('T')}")
        e(f"s_mov_b64 {s2('T4')}, {s2('WPTR')}")
        e(f"s_add_u32 {s('WPTR')}, {s('WPTR')}, {SLAB}")
        e(f"s_addc_u32 {s('WPTR1')}, {s('WPTR1')}, 0")
        e(f"s_add_u32 {s('IS')}, {s('IS')}, 1")
        e(f"s_cmp_eq_u32 {s('IS')}, {NSLABS}")
        e(f"s_cselect_b64 {s2('WPTR')}, {s2('WBASE')}, {s2('WPTR')}")
        e(f"s_cselect_b32 {s('IS')}, 0, {s('IS')}")

    def dma4(self):
        return [f"global_load_lds_dwordx4 v{V_DMAOFF}, {s2('T4')} offset:{1024 * k}" for k in range(4)]

    def weight_refill(self):
        self.weight_refill_setup()
        for t in self.dma4():
            self.e(t)

    def p_refill(self, ptab, pg, k=0):
        """issue_p: 16 KiB of the pixel table of pixel group `pg` * PGT + k (clamped to the last group: a tile block may reach past
        the image; those samples are never stored) -- same shape as a weight slab."""
        e = self.e
        e(f"s_mov_b32 m0, {s('T')}")
        if PGT == 1:
            e(f"s_mov_b32 {s('T8')}, {s(pg)}")
        else:
            e(f"s_mul_i32 {s('T8')}, {s(pg)}, {PGT}")
            e(f"s_add_u32 {s('T8')}, {s('T8')}, {k}")
            e(f"s_min_u32 {s('T8')}, {s('T8')}, {s('NPGM1')}")
        e(f"s_mov_b32 {s('T9')}, 0")
        e(f"s_lshl_b64 {s2('T8')}, {s2('T8')}, 14")
        e(f"s_add_u32 {s('T4')}, {s(ptab)}, {s('T8')}")
        e(f"s_addc_u32 {s('T5')}, {s(ptab + '1')}, {s('T9')}")
        for t in self.dma4():
            e(t)

    def q_refill(self, qtab, fg):
        """issue_q: rows of the frame table for frames fg*FT + wave*4 + i, clamped to the last frame; 1 KiB each (row r of the
        step = frame fg*FT + r; rows past the tile's FT frames are loaded all the same: every step is 16 DMA instructions)."""
        e = self.e
        e(f"s_mul_i32 {s('T6')}, {s(fg)}, {FT}")
        e(f"s_lshl_b32 {s('T7')}, {s('WAVE')}, 2")
        e(f"s_add_u32 {s('T6')}, {s('T6')}, {s('T7')}")
        for i in range(4):
            e(f"s_add_u32 {s('T8')}, {s('T6')}, {i}")
            e(f"s_min_u32 {s('T8')}, {s('T8')}, {s('NFM1')}")
            e(f"s_mov_b32 {s('T9')}, 0")
            e(f"s_lshl_b64 {s2('T8')}, {s2('T8')}, 10")
            e(f"s_add_u32 {s('T4')}, {s(qtab)}, {s('T8')}")
            e(f"s_addc_u32 {s('T5')}, {s(qtab + '1')}, {s('T9')}")
            e(f"s_add_u32 m0, {s('T')}, {1024 * i}")
            e("s_nop 0")
            e(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('T4')}")

    # ---- one slab
    def quad_mfmas(self, mb, j, dst, first_c=None, sprinkle=None):
        """12 MFMAs of k-quad j; sprinkle[n]: instructions tucked behind MFMA n (scalar, LDS and DMA issue is free there)."""
        if G != 3 and sprinkle:      # positions are written for the 12 MFMAs of a G = 3 quad: compress them onto 4 G slots
            packed = {}
            for pos in sorted(sprinkle):
                packed.setdefault(pos * G // 3, []).extend(sprinkle[pos])
            sprinkle = packed
        sprinkle = dict(sprinkle or {})
        w = V_W + 4 * (j % 4)
        for jj in range(4):
            for g in range(G):
                c = first_c if (first_c and j == 0 and jj == 0) else dst(g)
                self.e(f"v_mfma_f32_16x16x4_f32 {dst(g)}, v{w + jj}, v{self.inreg(g, j * 4 + jj)}, {c}")
                for t in sprinkle.pop(jj * G + g, []):
                    if isinstance(t, tuple):
                        self.lds_op(t[1], t[2])
                    else:
                        self.e(t)
        assert not sprinkle

    def a_read(self, slab_tag, tq, areg):
        wset = V_W + 4 * (tq % 4)
        self.lds_op(f"ds_read_b128 v[{wset}:{wset + 3}], v{areg} offset:{1024 * (tq % 16)}", ("A", slab_tag + tq // 16, tq % 16))

    def slab(self, mb, special=None):
        """special: list of (layer, out-of-line refill emitter) for the table steps issued from this slab position."""
        e = self.e
        dst = lambda g: self.acc(g, mb)
        for j in range(16):
            sprinkle = {}
            moves = {}
            if mb == 15 and j >= 1:
                # The layer's last slab: once k-quad j-1 has been issued, block j-1 of the B registers is dead and block j-1
                # of the accumulators has been final since slab j-1 -- move it across through this wave's LDS scratch
                # (ds_write from AGPRs, ds_read into VGPRs: free next to MFMAs; LDS executes a wave's operations in order),
                # so that only v_max is left for the VALU block between the layers.  (`moves` also goes to the out-of-line
                # copies of k-quad 14 that the table requests of the one-frame shapes put on this slab.)
                for g in range(G):
                    b0, a0 = self.inreg(g, (j - 1) * 4), A_ACC + (g * 16 + j - 1) * 4
                    moves.setdefault(1 + 4 * g, []).append(("lds", f"ds_write_b128 v{V_SCR}, a[{a0}:{a0 + 3}]", ("W", g, j - 1)))
                    moves.setdefault(2 + 4 * g, []).append(("lds", f"ds_read_b128 v[{b0}:{b0 + 3}], v{V_SCR}", ("R", g, j - 1)))
            if j == 13:     # the scalar side of `advance`, ahead of the barrier and behind MFMAs
                pre = [f"s_add_u32 {s('T')}, {s('CURB')}, {s('LDSBASE')}", f"s_add_u32 {s('CUR')}, {s('CUR')}, 1",
                       f"s_cmp_eq_u32 {s('CUR')}, {KRING}", f"s_cselect_b32 {s('CUR')}, 0, {s('CUR')}",
                       f"s_lshl_b32 {s('CURB')}, {s('CUR')}, 14", f"s_mov_b32 m0, {s('T')}", f"s_mov_b64 {s2('T4')}, {s2('WPTR')}"]
                sprinkle = {2 + n: [t] for n, t in enumerate(pre)}
            if j == 14:
                self.wait_lds(("A", mb, 15))      # both remaining quads of this slab are in registers
                e("s_waitcnt vmcnt(28)")
                e("s_barrier")
            self.a_read(mb, j + 2, V_RA + mb + (j + 2) // 16)
            if j == 8:      # next slab's bias (srcC of its first MFMAs); row offset of the next layer for the last slab
                self.lds_op(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BIAS} offset:{(mb + 1) * 64}", ("B", mb + 1))
            self.wait_lds(("A", mb, j))
            if j == 0:
                self.wait_lds(("B", mb))
            if j == 14:
                join = self.label("join")
                for layer, emit in (special or []):
                    ol = self.label("table")
                    e(f"s_cmp_eq_u32 {s('LAYER')}, {layer}")
                    e(f"s_cbranch_scc1 {ol}")
                    self.outofline.append((ol, join, emit, mb, j, {k: list(v) for k, v in moves.items()}))
                dma = self.dma4()
                ptr = [f"s_add_u32 {s('WPTR')}, {s('WPTR')}, {SLAB}", f"s_addc_u32 {s('WPTR1')}, {s('WPTR1')}, 0",
                       f"s_add_u32 {s('IS')}, {s('IS')}, 1", f"s_cmp_eq_u32 {s('IS')}, {NSLABS}",
                       f"s_cselect_b64 {s2('WPTR')}, {s2('WBASE')}, {s2('WPTR')}", f"s_cselect_b32 {s('IS')}, 0, {s('IS')}"]
                sprinkle = {0: [dma[0]], 3: [dma[1]], 6: [dma[2]], 9: [dma[3], *ptr]}
            for pos, items in moves.items():
                sprinkle.setdefault(pos, []).extend(items)
            self.quad_mfmas(mb, j, dst, first_c=f"a[{A_BIAS}:{A_BIAS + 3}]", sprinkle=sprinkle)
            if j == 14:
                e(f"{join}:")

    def ring_table(self):
        """v[RA + i] = LDS address of this lane's A quads in the buffer i steps after the current one, i = 0..16"""
        e = self.e
        e(f"s_mov_b32 {s('T6')}, {s('CURB')}")
        for i in range(17):
            if i:
                e(f"s_add_u32 {s('T6')}, {s('T6')}, {SLAB}")
                e(f"s_cmp_ge_u32 {s('T6')}, {KRING * SLAB}")
                e(f"s_cselect_b32 {s('T7')}, {KRING * SLAB}, 0")
                e(f"s_sub_u32 {s('T6')}, {s('T6')}, {s('T7')}")
            e(f"v_add_u32 v{V_RA + i}, {s('T6')}, v{V_LANE16}")

    def last_block_to_b_registers(self):
        """after the layer's last slab: blocks 0..14 of the accumulators already sit in the B registers (slab 15 moved them
        through LDS); block 15, which that slab produced, follows by v_accvgpr_read"""
        e = self.e
        self.wait_all_lds()
        e("s_nop 7")
        e("s_nop 3")
        for g in range(G):
            b = A_ACC + (g * 16 + 15) * 4
            for r in range(4):
                e(f"v_accvgpr_read_b32 v{self.inreg(g, 60 + r)}, a{b + r}")

    def relu_all(self):
        self.last_block_to_b_registers()
        for mb in range(16):
            for g in range(G):
                for r in range(4):
                    v = self.inreg(g, mb * 4 + r)
                    self.e(f"v_max_f32 v{v}, 0, v{v}")

    # ---- table steps
    def q_step(self, with_acc):
        """in[g][k] = (acc[g][k] +) q[frame of g][k].  q step: 16 rows of 1 KiB; row of (wave, g): 3 wave + g in the long shape
        (the wave part sits in V_QADDR), row 0 -- the tile's one frame -- in the others."""
        e = self.e
        QG = 1024 if FT > 1 else 0
        e(f"v_add_u32 v{V_TMP}, {s('CURB')}, v{V_QADDR}")
        if not with_acc:
            for g in range(G):
                for mb in range(16):
                    b = self.inreg(g, mb * 4)
                    self.lds_op(f"ds_read_b128 v[{b}:{b + 3}], v{V_TMP} offset:{g * QG + mb * 64}", ("Q", g, mb))
            self.wait_all_lds()
            return
        self.last_block_to_b_registers()      # in = acc
        items = [(g, mb) for g in range(G) for mb in range(16)]
        for n in range(3):
            g, mb = items[n]
            t = V_W + 4 * (n % 4)
            self.lds_op(f"ds_read_b128 v[{t}:{t + 3}], v{V_TMP} offset:{g * QG + mb * 64}", ("Q", g, mb))
        for n, (g, mb) in enumerate(items):
            if n + 3 < len(items):
                g2, mb2 = items[n + 3]
                t2 = V_W + 4 * ((n + 3) % 4)
                self.lds_op(f"ds_read_b128 v[{t2}:{t2 + 3}], v{V_TMP} offset:{g2 * QG + mb2 * 64}", ("Q", g2, mb2))
            b, t = self.inreg(g, mb * 4), V_W + 4 * (n % 4)
            self.wait_lds(("Q", g, mb))
            e(f"v_pk_add_f32 v[{b}:{b + 1}], v[{b}:{b + 1}], v[{t}:{t + 1}]")
            e(f"v_pk_add_f32 v[{b + 2}:{b + 3}], v[{b + 2}:{b + 3}], v[{t + 2}:{t + 3}]")
        self.wait_all_lds()

    def p_step(self, k=0):
        """in[g][k] = relu(in[g][k] + p[pixel][k]); p step: [mb][lane] f4, i.e. lane-linear like an A quad.  Long shape: every
        group of every wave has the tile's one pixel group.  Other shapes: step k holds pixel group k of the tile's block, which is
        group k % G of wave k // G -- that wave adds it, the others only pass the step."""
        e = self.e
        groups = range(G)
        skip = None
        if PGT > 1:
            groups = [k % G]
            skip = self.label("pskip")
            e(f"s_cmp_eq_u32 {s('WAVE')}, {k // G}")
            e(f"s_cbranch_scc0 {skip}")
        e(f"v_add_u32 v{V_RING}, {s('CURB')}, v{V_LANE16}")
        for mb in range(3):
            t = V_W + 4 * (mb % 4)
            self.lds_op(f"ds_read_b128 v[{t}:{t + 3}], v{V_RING} offset:{mb * 1024}", ("P", mb))
        for mb in range(16):
            if mb + 3 < 16:
                t2 = V_W + 4 * ((mb + 3) % 4)
                self.lds_op(f"ds_read_b128 v[{t2}:{t2 + 3}], v{V_RING} offset:{(mb + 3) * 1024}", ("P", mb + 3))
            self.wait_lds(("P", mb))
            t = V_W + 4 * (mb % 4)
            for g in groups:
                b = self.inreg(g, mb * 4)
                e(f"v_pk_add_f32 v[{b}:{b + 1}], v[{b}:{b + 1}], v[{t}:{t + 1}]")
                e(f"v_pk_add_f32 v[{b + 2}:{b + 3}], v[{b + 2}:{b + 3}], v[{t + 2}:{t + 3}]")
                for r in range(4):
                    e(f"v_max_f32 v{b + r}, 0, v{b + r}")
        self.wait_all_lds()
        if skip:
            e(f"{skip}:")

    # ---- what the advance that retires position `pos` of the tile's step sequence has to request: step pos + KRING
    @staticmethod
    def tile_steps():
        return ["Q0"] + [("P0", k) for k in range(PGT)] + ["W"] * 80 + ["Q5"] + [("P5", k) for k in range(PGT)] + ["W"] * 33

    def refill_at(self, pos):
        """emitter (or None = the next weight slab) for the step KRING after position pos; steps past the tile's end are the NEXT
        tile's (FGN, PGN)"""
        steps = self.tile_steps()
        tgt, nxt = pos + KRING, False
        if tgt >= len(steps):
            tgt, nxt = tgt - len(steps), True
        kind = steps[tgt]
        fg, pg = ("FGN", "PGN") if nxt else ("FG", "PG")
        if kind == "W":
            return None
        if kind == "Q0":
            return lambda: self.q_refill("Q0", fg)
        if kind == "Q5":
            return lambda: self.q_refill("Q5", fg)
        tab, k = kind
        return lambda: self.p_refill(tab, pg, k)

    def advance_and_refill(self, pos):
        self.advance()
        emit = self.refill_at(pos)
        if emit is None:
            self.weight_refill()
        else:
            emit()

    def prefetch_first_quads(self, with_bias):
        self.ring_table()
        if with_bias:
            self.e(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BIAS}")
            self.wait_all_lds()
        self.a_read(0, 0, V_RA)
        self.a_read(0, 1, V_RA)


def generate():
    b = Body()
    b.outofline = []
    e = b.e
    # ================= prologue: operands -> owned registers
    for dst, src in (("LDSBASE", "ldsbase"), ("NFG", "nfg"), ("TILE", "tile0"), ("TEND", "tile_end"), ("NFRAMES", "nframes"), ("HW", "hw"),
                     ("FG", "fg0"), ("PG", "pg0"), ("WAVE", "wave")) + ((("NPGM1", "npgm1"),) if PGT > 1 else ()):
        e(f"s_mov_b32 {s(dst)}, %[{src}]")
    for dst, src in (("WBASE", "wsrc"), ("Q0", "q0"), ("Q5", "q5"), ("P0", "p0"), ("P5", "p5"), ("OUT", "out")) + ((("TRACE", "trace"),) if TRACE else ()):
        e(f"s_mov_b64 {s2(dst)}, %[{src}]")
    e(f"s_mov_b64 {s2('WPTR')}, {s2('WBASE')}")
    e(f"s_sub_u32 {s('NFM1')}, {s('NFRAMES')}, 1")
    e(f"s_mov_b32 {s('IS')}, 0")
    e(f"s_mov_b32 {s('CUR')}, 0")
    e(f"s_mov_b32 {s('CURB')}, 0")
    for dst, src in ((V_LANE16, "lane16"), (V_DMAOFF, "dmaoff"), (V_BIAS0, "biasaddr"), (V_QADDR, "qaddr"), (V_BOUT, "boutaddr"), (V_PIX, "px"), (V_SCR, "scraddr")):
        e(f"v_mov_b32 v{dst}, %[{src}]")
    # prime the ring with the first tile's steps 0..7 (buffers 0..7): q0, its p0 steps, weight slabs
    def prime(st):
        if st == 0:
            e(f"s_mov_b32 {s('T')}, {s('LDSBASE')}")
        else:
            e(f"s_add_u32 {s('T')}, {s('LDSBASE')}, {SLAB * st}")
        emit = b.refill_at(st - KRING)       # (the request for step st)
        if emit is not None:
            emit()
        else:
            b.weight_refill_setup()
            e("s_nop 0")
            for t in b.dma4():
                e(t)
    for st in range(KRING - 1):
        prime(st)
    # step 0 landed and published (with the bias block the C++ prologue wrote); top the ring up
    e("s_waitcnt lgkmcnt(0)")
    e("s_waitcnt vmcnt(28)")
    e("s_barrier")
    prime(KRING - 1)

    # ================= tile loop
    e("S2L_TILE:")
    # A workgroup owns a contiguous range of tiles in pixel-group-major order (tile t = pixel group t / nfg, frame group
    # t % nfg): for ~84 tiles in a row the p0 / p5 rows it streams are the ones it streamed for the previous tile, i.e. L2
    # hits instead of HBM reads.  (FGN, PGN): the tile whose tables are prefetched during layer 6 -- the next one, or, on
    # the range's last tile, the same one again (valid addresses, unused data).
    e(f"s_add_u32 {s('FGN')}, {s('FG')}, 1")
    e(f"s_cmp_eq_u32 {s('FGN')}, {s('NFG')}")
    e(f"s_cselect_b32 {s('FGN')}, 0, {s('FGN')}")
    e(f"s_cselect_b32 {s('T6')}, 1, 0")
    e(f"s_add_u32 {s('PGN')}, {s('PG')}, {s('T6')}")
    e(f"s_add_u32 {s('T7')}, {s('TILE')}, 1")
    e(f"s_cmp_lt_u32 {s('T7')}, {s('TEND')}")
    e(f"s_cselect_b32 {s('FGN')}, {s('FGN')}, {s('FG')}")
    e(f"s_cselect_b32 {s('PGN')}, {s('PGN')}, {s('PG')}")
    # ---- h0 = relu(p0[pixel] + q0[frame])
    b.trace(0)
    b.q_step(False)
    b.advance_and_refill(0)
    for k in range(PGT):
        b.p_step(k)
        b.advance_and_refill(1 + k)
    e(f"v_mov_b32 v{V_BIAS}, v{V_BIAS0}")
    e(f"s_mov_b32 {s('LAYER')}, 0")
    b.trace(1)
    b.prefetch_first_quads(True)
    loop_state = list(b.lds)

    # ---- the layer loop
    e("S2L_LAYER:")
    # the table requests issued from inside the layer loop, keyed by (slab, runtime layer): slab mb of layer L is position
    # base1 + 16 L + mb of the tile for L < 5 and base2 + 16 (L - 5) + mb behind the q5 / p5 steps
    base1 = 1 + PGT
    base2 = base1 + 81 + PGT
    special = {}
    for L in range(NLAYERS):
        for mb in range(16):
            emit = b.refill_at((base1 + 16 * L if L < 5 else base2 + 16 * (L - 5)) + mb)
            if emit is not None:
                special.setdefault(mb, []).append((L, emit))
    if (G, PGT, FT) == (3, 1, 12):
        assert {mb: [L for L, _ in v] for mb, v in special.items()} == {7: [4], 8: [4, 6], 9: [6]}
    for mb in range(16):
        b.slab(mb, special.get(mb))
    slab_end = list(b.lds)         # the first quads of the next slab + the tail of the accumulator moves: both continuations
    assert [t for t in slab_end if t[0] == "A"] == [("A", 16, 0), ("A", 16, 1)]      # below wait for all of it first
    e(f"s_cmp_eq_u32 {s('LAYER')}, 4")
    e("s_cbranch_scc1 S2L_SKIP")
    b.relu_all()
    b.ring_table()
    b.lds = list(loop_state)       # (complete, which is stronger than "in flight")
    if TRACE:
        b.trace("layer")
        e("s_nop 0")      # (the first quads are waited for as "in flight": already complete, which is stronger)
        b.lds = list(loop_state)
    e(f"s_add_u32 {s('LAYER')}, {s('LAYER')}, 1")
    e(f"v_add_u32 v{V_BIAS}, 1024, v{V_BIAS}")
    e(f"s_cmp_lt_u32 {s('LAYER')}, {NLAYERS}")
    e("s_cbranch_scc1 S2L_LAYER")

    # ---- output layer (3 rows padded to one M-block), no activation; quads 0, 1 of its slab are in flight
    b.lds = list(loop_state)
    b.lds_op(f"ds_read_b128 a[{A_BIAS}:{A_BIAS + 3}], v{V_BOUT}", ("B", 0))
    rgb = lambda g: f"a[{A_RGB + 4 * g}:{A_RGB + 4 * g + 3}]"
    for j in range(16):
        if j + 2 < 16:
            b.a_read(0, j + 2, V_RA)
        b.wait_lds(("A", 0, j))
        if j == 0:
            b.wait_lds(("B", 0))
        b.quad_mfmas(0, j, rgb, first_c=f"a[{A_BIAS}:{A_BIAS + 3}]")
    b.wait_all_lds()
    b.advance_and_refill(len(b.tile_steps()) - 1)      # (long shape: the next weight slab; publishes the next tile's q0 step)
    if PGT == 1:
        # ---- store: lanes 0..15 (k-subgroup 0) hold rgb of pixel pg*16 + lane for the wave's three frames
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
        e(f"s_mov_b32 {s('T6')}, {s('FG')}")                  # the tile's one frame (always < NFRAMES)
    for g in range(G):
        skip = b.label("nostore")
        if PGT > 1:
            # group g of this wave = pixel group PG * PGT + WAVE * G + g: lanes 0..15 hold rgb of its pixels
            e(f"s_mul_i32 {s('T8')}, {s('PG')}, {PGT}")
            e(f"s_mul_i32 {s('T9')}, {s('WAVE')}, {G}")
            e(f"s_add_u32 {s('T8')}, {s('T8')}, {s('T9')}")
            e(f"s_add_u32 {s('T8')}, {s('T8')}, {g}")
            e(f"s_lshl_b32 {s('T8')}, {s('T8')}, 4")
            e(f"v_add_u32 v{V_TMP}, {s('T8')}, v{V_PIX}")
            e(f"v_cmp_gt_u32 vcc, {s('HW')}, v{V_TMP}")
            e(f"v_mul_u32_u24 v{V_PIXOFF}, 12, v{V_TMP}")
            e("s_nop 3")
            e(f"s_and_b64 exec, vcc, 0xffff")
            e("s_nop 7")
            e("s_nop 7")
        e(f"s_add_u32 {s('T7')}, {s('T6')}, {g if PGT == 1 else 0}")
        e(f"s_cmp_ge_u32 {s('T7')}, {s('NFRAMES')}")
        e(f"s_cbranch_scc1 {skip}")
        # byte offset of the frame = frame * hw * 12 (64-bit)
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
    b.trace(9)
    # ---- next tile of this workgroup
    e(f"s_add_u32 {s('TILE')}, {s('TILE')}, 1")
    e(f"s_mov_b32 {s('FG')}, {s('FGN')}")
    e(f"s_mov_b32 {s('PG')}, {s('PGN')}")
    e(f"s_cmp_lt_u32 {s('TILE')}, {s('TEND')}")
    e("s_cbranch_scc1 S2L_TILE")
    e("s_waitcnt vmcnt(0)")     # run-ahead DMAs must land before the workgroup's LDS is released
    e("s_branch S2L_END")

    # ================= out of line: pts_linears[5] on cat([skip, h4]): + q5[frame] + p5[pixel], ReLU
    e("S2L_SKIP:")
    b.lds = list(slab_end)     # (the two quads read from the q5 step are discarded)
    b.q_step(True)
    b.advance_and_refill(base1 + 80)
    for k in range(PGT):
        b.p_step(k)
        b.advance_and_refill(base1 + 81 + k)
    b.trace("layer")
    e(f"s_mov_b32 {s('LAYER')}, 5")
    e(f"v_add_u32 v{V_BIAS}, 1024, v{V_BIAS}")
    b.prefetch_first_quads(False)
    assert b.lds == loop_state
    e("s_branch S2L_LAYER")
    # ================= out of line: the four table refills per tile (the slab's k-quad 14 is repeated here without DMA tucks)
    for ol, join, emit, mb, j, moves in b.outofline:
        e(f"{ol}:")
        emit()
        b.quad_mfmas(mb, j, lambda g: b.acc(g, mb), sprinkle=moves)       # (same LDS operations, in the same order, as the in-line copy)
        e(f"s_branch {join}")
    e("S2L_END:")
    return [x for x in b.L if x is not None]


OPERANDS = """      :
      : [ldsbase] "s"(ldsbase), [nfg] "s"(nfg), [tile0] "s"(tile0), [tile_end] "s"(tile_end), [nframes] "s"(a.nframes), [hw] "s"(a.hw),
        [fg0] "s"(fg0), [pg0] "s"(pg0),NPGM1_OPERAND
        [wave] "s"(wave), [wsrc] "s"(wsrc), [q0] "s"(a.q0), [q5] "s"(a.q5), [p0] "s"(a.p0t), [p5] "s"(a.p5t), [out] "s"(a.out),
        [lane16] "v"(lane16), [dmaoff] "v"(dmaoff), [biasaddr] "v"(biasaddr), [qaddr] "v"(qaddr), [boutaddr] "v"(boutaddr), [px] "v"(px), [scraddr] "v"(scraddr)TRACE_OPERAND
"""


def main(path, variant="long"):
    set_variant(variant)
    lines = generate()
    if variant != "long":      # the variants share a translation unit: their labels must differ (the long shape keeps its text)
        lines = [x.replace("S2L_", f"S2L{variant[0].upper()}_") for x in lines]
    assert S["NPGM1"] == S_LAST       # the long shape does not use it: its text (clobber list included) stays what it was
    s_last = S_LAST - (1 if PGT == 1 else 0)
    clob = [f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)] + [f"s{r}" for r in range(36, s_last + 1)]
    clob += ["vcc", "scc", "memory"]   # (m0 and exec: nothing follows the body; exec is restored)
    out = ["// GENERATED by csrc/gen_render_body.py -- do not edit; the generator is the source.", "asm volatile("]
    out += [f'    "{x}\\n\\t"' for x in lines]
    out.append(OPERANDS.rstrip("\n").replace("TRACE_OPERAND", ', [trace] "s"(g_trace)' if TRACE else "")
               .replace("NPGM1_OPERAND", ' [npgm1] "s"(npgm1),' if PGT > 1 else ""))
    out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return len(lines)


def main_all(objdir):
    """render_body.inc (the long shape, its text pinned) + render_body_wide.inc + render_body_single.inc"""
    n = {}
    for variant in VARIANTS:
        n[variant] = main(os.path.join(objdir, "render_body.inc" if variant == "long" else f"render_body_{variant}.inc"), variant)
    return n


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    for k, v in main_all(d).items():
        print(f"render body, {k}: {v} instructions")
