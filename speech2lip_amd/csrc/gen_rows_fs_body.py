"""Generates rows_fs_body.inc: the FEATURE-SPLIT tile of the general-row MLP (s2l::rows_fs_kernel, csrc/rows.hip) -- TalkingFace.rgb_forward on a
few thousand arbitrary rows, which is what the reference's per-frame driver calls once per frame (inference.py:152-159: 4 096 rows at 64 x 64,
9 216 at 96 x 96; tf_nerf.py:225-285).

rows_fwd_kernel gives a wave 16 (or 32) rows and all 256 features of every layer: 2 x 128 + 7 x 1024 + 64 v_mfma_f32_16x16x4_f32 in one
dependent stream = 100 us and more, on 64 of 256 CUs at 4 096 rows.  Here a tile is 16 rows and its four waves split the features as in
gen_render_fs_body.py (read that file first: ring, exchange blocks, A-quad sets, in-flight lists and derived waits are the same): wave w
owns M-blocks 4 w .. 4 w + 3 of every layer.  What differs from the renderer's tile: no pixel / frame tables -- layer 0 and the skip half of
pts_linears[5] are MFMA layers of their own on the row's embedded input x[128] (the folded matrices G0 / G5 of s2l_layout.h, K = 128: eight A quads
per M-block against 32 B registers in the order kin(j, q) = 32 q + j of rows_fwd_kernel::load_x), and the skip layer continues the accumulators
G5 x + c5 left, exactly the order of rows_fwd_kernel -- so every output is the same chain of MFMAs on the same operands: THE SAME BITS as the
column form (tests/test_gpu_parity.py holds the two against each other).

A tile's weight stream, per wave, in consumption order (4-KiB pieces = 4 A quads = 16 k-steps):
    0 .. 7      G0: the wave's four M-blocks x 2 pieces             (OFF_WG0 + 32 KiB x wave)
    8 .. 71     pts_linears 1 .. 4: four M-blocks x 4 pieces each   (the wave's 64 contiguous KiB per layer, as the renderer)
    72 .. 79    G5
    80 .. 127   pts_linears[5][:, 256:], 6, 7
    128 .. 131  the output layer's slab
    132 .. 135  nothing (a tile is a whole number of ring revolutions and of A-set rotations)
x of the NEXT tile is requested into the same 32 registers right behind the G5 pass, the last reader of this tile's.

Register map (per wave):  v0-63 B operands in[j] | v64-79 acc[mbl][r] | v80-111 x[32] | v112.. addresses
                          a0-111 bias[L][mbl][r] | a112-115 output bias | a116-119 rgb | a120-135 c0 (G0's bias) | a136-151 c5 | a152.. A-quad sets."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_render_fs_body as F      # noqa: E402  (the in-flight lists and the waits derived from them)

NLAYERS, RING, PIECE = 7, 8, 4096
NPOS, NREAL = 136, 132
LOOK = 6
NSETS = LOOK + 2
assert (NPOS - NREAL) * 4 % NSETS == 0 and NPOS * 4 % NSETS == 0 and NPOS % RING == 0 and (NPOS - NREAL) == RING // 2
LDS_RING_PER_WAVE = RING * PIECE
LDS_X = 4 * LDS_RING_PER_WAVE
LDS_BYTES = LDS_X + 2 * 16384                     # 160 KiB, as the renderer's feature-split tile

V_IN, V_ACC, V_X = 0, 64, 80
V_LANE16, V_OWNQ, V_RINGA, V_XW, V_XR, V_PIX, V_Q128, V_XOFF, V_ROWOFF, V_TMP, V_ZERO = range(112, 123)
V_LAST = 122
A_BIAS, A_BOUT, A_RGB, A_C0, A_C5, A_W = 0, 112, 116, 120, 136, 152
A_LAST = A_W + 4 * NSETS - 1

S = F._scalar_map(36, singles="TILE TEND TOTAL WAVE LDSW T6 T7".split(),
                  pairs=("WB", "WOUT", "WG0", "WG5", "X", "OUT", "BIAS", "T4", "T8", "EX"))
S.update(T5=S["T41"], T9=S["T81"])
S_LAST = max(S.values())


def s(n):
    return f"s{S[n]}"


def s2(n):
    return f"s[{S[n]}:{S[n] + 1}]"


class Body(F.Body):
    def label(self, stem):
        self.nlabel += 1
        return f"S2LR_{stem}_{self.nlabel}"

    @staticmethod
    def pos_source(p):
        """(scalar base pair, byte offset) of stream position p (0 .. NREAL-1)"""
        if p < 8:
            return "WG0", p * PIECE
        if p < 72:
            return "WB", ((p - 8) // 16) * 262144 + ((p - 8) % 16) * PIECE
        if p < 80:
            return "WG5", (p - 72) * PIECE
        if p < 128:
            return "WB", (4 + (p - 80) // 16) * 262144 + ((p - 80) % 16) * PIECE
        return "WOUT", (p - 128) * PIECE

    def dma(self, p):
        """the four 1-KiB requests of stream position p (mod NPOS: the next tile's) into ring slot p % RING of this wave"""
        tag, p = p, p % NPOS
        if p >= NREAL:
            return
        e = self.e
        base, off = self.pos_source(p)
        e(f"s_add_u32 m0, {s('LDSW')}, {(p % RING) * PIECE}")
        if off:
            e(f"s_add_u32 {s('T4')}, {s(base)}, {off}")
            e(f"s_addc_u32 {s('T5')}, {s(base + '1')}, 0")
        else:
            e(f"s_mov_b64 {s2('T4')}, {s2(base)}")
            e("s_nop 0")
        for k in range(4):
            self.vm_op(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('T4')} offset:{1024 * k}", ("D", tag))

    def a_read(self, p, quad):
        g = p * 4 + quad
        reg = A_W + 4 * (g % NSETS)
        if quad == 0:
            self.wait_vm(("D", p))                      # the piece has landed
        self.lds_op(f"ds_read_b128 a[{reg}:{reg + 3}], v{V_RINGA} offset:{(p % RING) * PIECE + quad * 1024}", ("A", g))

    def block(self, p0, nq, bbase, dst, first_c, tucks=None):
        """nq A quads from stream position p0 on (8: a folded K = 128 M-block, 16: a slab) against B registers bbase .. bbase + 4 nq - 1 into the
        accumulator quad dst; first_c: srcC of the first MFMA (an AGPR bias is copied into a VGPR accumulator first, see gen_render_fs_body.py;
        first_c == dst: the chain continues).  Precondition: the first LOOK A quads from position p0 on are in flight."""
        e = self.e
        if dst.startswith("v[") and first_c.startswith("a["):
            d0, c0 = int(dst[2:].split(":")[0]), int(first_c[2:].split(":")[0])
            for r in range(4):
                e(f"v_accvgpr_read_b32 v{d0 + r}, a{c0 + r}")
            e("s_nop 1")                                # (VALU write -> MFMA srcC)
            first_c = dst
        for q in range(nq):
            p, quad = p0 + q // 4, q % 4
            g = p * 4 + quad
            nxt = self.following(p0, q + LOOK)
            if nxt is not None:
                self.a_read(*nxt)
            self.wait_lds(("A", g))
            if quad == 3:
                self.dma(p + RING)                      # position p's last quad is in registers: its slot takes position p + 8
            reg = A_W + 4 * (g % NSETS)
            for jj in range(4):
                j = q * 4 + jj
                c = first_c if j == 0 else dst
                e(f"v_mfma_f32_16x16x4_f32 {dst}, a{reg + jj}, v{bbase + j}, {c}")
            for t in (tucks or {}).get(q, []):
                t()

    def following(self, p0, q):
        p = p0 + q // 4
        if p >= NREAL:
            p = (p - NREAL) % NPOS                      # positions 132 .. 135 do not exist: the next tile's 0 ..
            return (p + NPOS, q % 4)
        return (p, q % 4)


def acc(mbl):
    return f"v[{V_ACC + 4 * mbl}:{V_ACC + 4 * mbl + 3}]"


def generate():
    b = Body()
    e = b.e
    # ================= prologue
    for dst, src in (("TILE", "tile0"), ("TEND", "tile_end"), ("TOTAL", "total"), ("WAVE", "wave"), ("LDSW", "ldsw")):
        e(f"s_mov_b32 {s(dst)}, %[{src}]")
    for dst, src in (("WB", "wb"), ("WOUT", "wout"), ("WG0", "wg0"), ("WG5", "wg5"), ("X", "x"), ("OUT", "out"), ("BIAS", "bias")):
        e(f"s_mov_b64 {s2(dst)}, %[{src}]")
    e(f"v_mov_b32 v{V_TMP}, %[lane]")
    e(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_TMP}")                         # lane * 16
    e(f"v_and_b32 v{V_PIX}, 15, v{V_TMP}")                               # row of the lane inside the tile
    e(f"v_lshrrev_b32 v{V_Q128}, 4, v{V_TMP}")                           # q
    e(f"v_lshlrev_b32 v{V_OWNQ}, 4, v{V_Q128}")                          # q * 16: bias rows hold features 16 mb + 4 q .. + 3
    e(f"s_lshl_b32 {s('T6')}, {s('WAVE')}, 8")
    e(f"v_add_u32 v{V_OWNQ}, {s('T6')}, v{V_OWNQ}")                      # ... of the wave's own M-blocks
    e(f"v_lshlrev_b32 v{V_Q128}, 7, v{V_Q128}")                          # q * 128 bytes: x[32 q ..] of a row (rows_fwd_kernel::load_x)
    e(f"v_add_u32 v{V_RINGA}, {s('LDSW')}, v{V_LANE16}")                 # A quads are lane-linear in a piece
    e(f"s_add_u32 {s('T7')}, %[lds0], {LDS_X}")
    e(f"v_add_u32 v{V_XR}, {s('T7')}, v{V_LANE16}")                      # the exchange blocks: [M-block][lane][4]
    e(f"s_lshl_b32 {s('T6')}, {s('WAVE')}, 12")
    e(f"v_add_u32 v{V_XW}, {s('T6')}, v{V_XR}")                          # ... this wave's four M-blocks
    e(f"v_mov_b32 v{V_ZERO}, 0")
    # the wave's biases: bias[L][16 (4 w + mbl) + 4 q + r]; then the output layer's four, then c0 / c5 (OFF_BOUT, OFF_BG0, OFF_BG5 follow OFF_BIAS)
    for L in range(NLAYERS):
        if L:
            e(f"s_add_u32 {s('BIAS')}, {s('BIAS')}, 1024")
            e(f"s_addc_u32 {s('BIAS1')}, {s('BIAS1')}, 0")
        for mbl in range(4):
            r = A_BIAS + (4 * L + mbl) * 4
            b.vm_op(f"global_load_dwordx4 a[{r}:{r + 3}], v{V_OWNQ}, {s2('BIAS')} offset:{mbl * 64}", ("B", L, mbl))
    e(f"s_add_u32 {s('BIAS')}, {s('BIAS')}, 1024")
    e(f"s_addc_u32 {s('BIAS1')}, {s('BIAS1')}, 0")
    b.vm_op(f"global_load_dwordx4 a[{A_BOUT}:{A_BOUT + 3}], v{V_ZERO}, {s2('BIAS')}", ("B", 7, 0))
    for k, a0 in enumerate((A_C0, A_C5)):
        e(f"s_add_u32 {s('BIAS')}, {s('BIAS')}, {16 if k == 0 else 1024}")
        e(f"s_addc_u32 {s('BIAS1')}, {s('BIAS1')}, 0")
        for mbl in range(4):
            b.vm_op(f"global_load_dwordx4 a[{a0 + 4 * mbl}:{a0 + 4 * mbl + 3}], v{V_OWNQ}, {s2('BIAS')} offset:{mbl * 64}", ("B", 8 + k, mbl))

    def load_x(tile):
        """x[32 q .. 32 q + 31] of row min(16 tile + lane's row, total - 1) -> v[V_X ..] (rows_fwd_kernel::load_x: the same operands in the same registers)"""
        e(f"s_lshl_b32 {s('T6')}, {s(tile)}, 4")
        e(f"s_sub_u32 {s('T7')}, {s('TOTAL')}, 1")
        e(f"v_add_u32 v{V_TMP}, {s('T6')}, v{V_PIX}")
        e(f"v_min_u32 v{V_TMP}, {s('T7')}, v{V_TMP}")
        e(f"v_lshlrev_b32 v{V_XOFF}, 9, v{V_TMP}")                        # 128 floats per row (the launcher keeps total * 512 < 2^31)
        e(f"v_add_u32 v{V_XOFF}, v{V_XOFF}, v{V_Q128}")
        for i in range(8):
            b.vm_op(f"global_load_dwordx4 v[{V_X + 4 * i}:{V_X + 4 * i + 3}], v{V_XOFF}, {s2('X')} offset:{16 * i}", ("X", i))

    load_x("TILE")
    for p in range(RING):
        b.dma(p)
    b.wait_vm(("X", 7))
    b.vm = [t for t in b.vm if t[0] == "D"]
    assert len(b.vm) == 4 * RING
    for q in range(LOOK):
        b.a_read(q // 4, q % 4)
    e("S2LR_BODY:")
    vm_body, lds_body = list(b.vm), list(b.lds)
    b.wait_vm(("X", 7))                        # (a no-op in the model: the tile's x was waited for at the previous tile's end / in the prologue)

    def relu_write(par, mbl):
        """ReLU of M-block mbl and its 1 KiB of exchange block `par` ([M-block][lane][4])"""
        def run():
            a0 = V_ACC + 4 * mbl
            for r in range(4):
                e(f"v_max_f32 v{a0 + r}, 0, v{a0 + r}")
            b.lds_op(f"ds_write_b128 v{V_XW}, v[{a0}:{a0 + 3}] offset:{par * 16384 + mbl * 1024}", ("X", par, mbl))
        return run

    def exchange(par):
        """the layer's last M-block, the barrier, and all 256 activations back as B operands"""
        e("s_nop 7")
        e("s_nop 7")
        relu_write(par, 3)()
        b.wait_all_lds()
        e("s_barrier")
        for mb in range(16):
            b.lds_op(f"ds_read_b128 v[{V_IN + 4 * mb}:{V_IN + 4 * mb + 3}], v{V_XR} offset:{par * 16384 + mb * 1024}", ("H", mb))
            if mb == 13:                       # (lgkmcnt counts to 15)
                b.wait_lds(("H", 0))
        b.wait_all_lds()
        e("s_nop 1")

    par = 0
    # ---- h0 = relu(G0 x + c0): positions 0 .. 7
    for mbl in range(4):
        tucks = {1: [relu_write(par, mbl - 1)]} if mbl else {}
        b.block(2 * mbl, 8, V_X, acc(mbl), f"a[{A_C0 + 4 * mbl}:{A_C0 + 4 * mbl + 3}]", tucks)
    exchange(par)
    par ^= 1
    for L in range(NLAYERS):
        if L == 4:
            # pts_linears[5] on cat([skip, h4]): G5 x + c5 first (positions 72 .. 79), then the slabs of W5[:, 256:] continue the accumulators
            for mbl in range(4):
                b.block(72 + 2 * mbl, 8, V_X, acc(mbl), f"a[{A_C5 + 4 * mbl}:{A_C5 + 4 * mbl + 3}]")
            # the next tile's x into the registers the G5 pass has just read (its last MFMAs are issued: B operands are read in the first passes)
            e(f"s_add_u32 {s('T8')}, {s('TILE')}, 1")
            e(f"s_cmp_lt_u32 {s('T8')}, {s('TEND')}")
            e(f"s_cselect_b32 {s('T8')}, {s('T8')}, {s('TILE')}")     # (the range's last tile: its own rows again -- valid addresses, unused data)
            e("s_nop 7")
            e("s_nop 7")
            load_x("T8")
        for mbl in range(4):
            p0 = (8 if L < 4 else 80 - 64) + L * 16 + mbl * 4
            bias = A_BIAS + (4 * L + mbl) * 4
            tucks = {1: [relu_write(par, mbl - 1)]} if mbl else {}
            b.block(p0, 16, V_IN, acc(mbl), acc(mbl) if L == 4 else f"a[{bias}:{bias + 3}]", tucks)
        exchange(par)
        par ^= 1
    # ---- output layer (3 rows padded to one M-block), every wave (only wave 0 stores): positions 128 .. 131
    b.block(128, 16, V_IN, f"a[{A_RGB}:{A_RGB + 3}]", f"a[{A_BOUT}:{A_BOUT + 3}]")
    for p in range(NPOS + 4, NPOS + RING):     # positions 132 .. 135 hold nothing: the next tile's 4 .. 7 take their slots now
        b.dma(p)
    # ---- store: wave 0, lanes 0 .. 15 hold rgb of row 16 tile + lane
    skip = b.label("nostore")
    e(f"s_cmp_eq_u32 {s('WAVE')}, 0")
    e(f"s_cbranch_scc0 {skip}")
    e(f"s_lshl_b32 {s('T6')}, {s('TILE')}, 4")
    e(f"v_add_u32 v{V_TMP}, {s('T6')}, v{V_PIX}")
    e(f"v_cmp_gt_u32 vcc, {s('TOTAL')}, v{V_TMP}")
    e(f"v_mul_u32_u24 v{V_ROWOFF}, 12, v{V_PIX}")
    e(f"s_mov_b64 {s2('EX')}, exec")
    e("s_nop 3")
    e("s_and_b64 exec, vcc, 0xffff")
    e(f"s_mul_hi_u32 {s('T9')}, {s('T6')}, 12")             # byte offset of the tile's first row = 16 tile * 12 (64-bit)
    e(f"s_mul_i32 {s('T8')}, {s('T6')}, 12")
    e(f"s_add_u32 {s('T4')}, {s('OUT')}, {s('T8')}")
    e(f"s_addc_u32 {s('T5')}, {s('OUT1')}, {s('T9')}")
    e("s_nop 7")
    e("s_nop 7")
    e(f"global_store_dwordx3 v{V_ROWOFF}, a[{A_RGB}:{A_RGB + 2}], {s2('T4')}")      # (never counted: a wait computed without it is stricter for wave 0)
    e(f"s_mov_b64 exec, {s2('EX')}")
    e(f"{skip}:")
    # ---- next tile
    b.wait_vm(("X", 7))
    e(f"s_add_u32 {s('TILE')}, {s('TILE')}, 1")
    e(f"s_cmp_lt_u32 {s('TILE')}, {s('TEND')}")
    e("s_cbranch_scc0 S2LR_END")
    b.vm = [(t[0], t[1] - NPOS) if t[0] == "D" else t for t in b.vm]
    b.lds = [(t[0], t[1] - NPOS * 4) for t in b.lds]
    assert b.vm == vm_body, (b.vm[:6], vm_body[:6], len(b.vm), len(vm_body))
    assert b.lds == lds_body, (b.lds, lds_body)
    e("s_branch S2LR_BODY")
    e("S2LR_END:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")      # run-ahead requests must land before the workgroup's LDS is released
    return b


OPERANDS = """      :
      : [tile0] "s"(tile0), [tile_end] "s"(tile_end), [total] "s"(total), [wave] "s"(wave), [ldsw] "s"(ldsw),
        [wb] "s"(wb), [wout] "s"(wout), [wg0] "s"(wg0), [wg5] "s"(wg5), [x] "s"(a.x), [out] "s"(a.out), [bias] "s"(biasp),
        [lds0] "s"(lds0), [lane] "v"(lane)
"""


def main(path):
    b = generate()
    lines = b.L
    clob = [f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)] + [f"s{r}" for r in range(36, S_LAST + 1)]
    clob += ["vcc", "scc", "memory"]
    out = ["// GENERATED by csrc/gen_rows_fs_body.py -- do not edit; the generator is the source.", "asm volatile("]
    out += [f'    "{x}\\n\\t"' for x in lines]
    out.append(OPERANDS.rstrip("\n"))
    out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return len(lines)


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"rows body, feature-split: {main(os.path.join(d, 'rows_fs_body.inc'))} instructions")
