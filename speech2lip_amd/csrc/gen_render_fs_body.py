"""Generates render_fs_body.inc: the FEATURE-SPLIT tile of the lip renderer (s2l::render_fs_kernel, csrc/render.hip) -- the shape for ONE or a
few frames per call, the reference's own operating mode (inference.py:129, 140-159).

Why another shape: in the other three (gen_render_body.py) a wave carries its 16-sample column through all 256 features of every layer:
7 x 1024 v_mfma_f32_16x16x4_f32 x 32 cycles = 97 us, however idle the rest of the chip is (a 64 x 64 frame is 64 tiles of the `single` shape on
256 CUs).  Here a tile is 16 samples (one pixel group of one frame) and its four waves SPLIT THE FEATURES: wave w owns M-blocks 4 w .. 4 w + 3
(features 64 w .. 64 w + 63) of every layer -- 256 MFMAs per layer and wave -- and the layer's activations are exchanged through LDS.

The bits are those of the other shapes: every output is the same chain -- bias as srcC, then k-steps j = 0 .. 63 in order on the same A quads
(the slab's packing, s2l_layout.h) against the same B values -- because the exchange block is laid out [M-block][lane][4], i.e. wave w's
accumulator registers of M-block mb ARE what every wave reads back as B registers 4 mb .. 4 mb + 3 (kfeat order); ReLU (v_max_f32 0, x), the
layer-0 sum q0 + p0 and the skip step (acc + q5) + p5 are the same instructions on the same operands.

Memory: a wave streams ITS OWN slabs (nobody else reads them): a private ring of eight 4-KiB pieces (a quarter slab = 16 k-steps) filled by
global_load_lds_dwordx4, eight pieces ahead, across tile boundaries -- no barrier for weights, only counted vmcnt.  The tile's stream is static:
positions 0 .. 111 = layer L = p // 16, piece p % 16 of the wave's 64 contiguous KiB of that layer, 112 .. 115 = the output layer's slab, 116 ..
119 = nothing (so that a tile is a whole number of ring revolutions); the request for position p + 8 leaves when the last A quad of position p
is in registers.  Tables: p0 / q0 (the tile's pixel group and frame) come by ordinary loads straight into registers, one tile ahead; of p5 / q5
a wave needs only its own 64 features.  Biases: the wave's 7 x 4 x 4 values live in AGPRs for the whole launch.  LDS: 4 x 32 KiB of rings + two
16-KiB exchange blocks (alternating by layer: one barrier per layer) = 160 KiB.

The generator keeps a list of LDS operations AND a list of vector-memory operations in flight; every s_waitcnt is derived from them (loads return
in order: "at most n outstanding" retires everything older than the n newest).  The rgb store, which only wave 0 issues, is never counted: a wait
computed without it is stricter for wave 0 and exact for the others.

Register map (per wave):  v0-63 B operands in[j] | v64-79 acc[mbl][r] | v80-143 p0 of the NEXT tile | v144-207 q0 of the next tile | v208-223 p5,
                          v224-239 q5 (own features) | v240.. addresses
                          a0-111 bias[L][mbl][r] | a112-115 output bias | a116-119 rgb accumulator | a120.. NSETS A-quad sets (a120-151 at look-ahead 6)."""
import os
import sys

NLAYERS, RING, PIECE, NPOS, NREAL = 7, 8, 4096, 120, 116
LOOK = int(os.environ.get("S2L_FS_LOOK", "6"))     # A quads requested ahead of the one the MFMAs are on (register sets: LOOK + 2)
NSETS = LOOK + 2
assert (NPOS - NREAL) * 4 % NSETS == 0 and NPOS * 4 % NSETS == 0, "the A sets rotate with the quad index: the skipped positions and a tile must be whole rotations"
EXP = int(os.environ.get("S2L_FS_EXP", "0"))      # pricing builds (results wrong): 1 no exchange (writes, barrier, reads), 2 nobody waits for a piece, 4 no MFMAs
TRACE = os.environ.get("S2L_FS_TRACE") == "1"     # experiment build (tools/dev/trace_render_fs.py; render.hip with -DS2L_EXP_TRACE): wave 0 stamps s_memtime per phase
LDS_RING_PER_WAVE = RING * PIECE                  # 32 KiB
LDS_X = 4 * LDS_RING_PER_WAVE                     # two exchange blocks of 16 KiB behind the four rings
LDS_BYTES = LDS_X + 2 * 16384                     # 163 840 = 160 KiB

V_IN, V_ACC, V_P0, V_Q0, V_P5, V_Q5 = 0, 64, 80, 144, 208, 224
V_LANE16, V_QOFF, V_OWNQ, V_OWNP, V_RINGA, V_XW, V_XR, V_PIX, V_PIXOFF, V_TMP, V_ZERO = 240, 241, 242, 243, 244, 245, 246, 247, 248, 249, 250
V_LAST = 250
A_BIAS, A_BOUT, A_RGB, A_W = 0, 112, 116, 120
A_LAST = A_W + 4 * NSETS - 1


def _scalar_map(first, singles, pairs):
    m, r = {}, first
    for n in pairs:
        r += r & 1
        m[n], m[n + "1"] = r, r + 1
        r += 2
    for n in singles:
        m[n] = r
        r += 1
    return m


S = _scalar_map(36, singles="TILE TEND NFG FG PG FGN PGN HW WAVE LDSW T6 T7".split(),
                pairs=("WB", "WOUT", "P0", "P5", "Q0", "Q5", "OUT", "BIAS", "T4", "T8", "EX"))
S.update(T5=S["T41"], T9=S["T81"])
S_LAST = max(S.values())
if TRACE:
    S.update(TRACE=S_LAST + 1 + (S_LAST + 1) % 2)
    S.update(TRACE1=S["TRACE"] + 1, TS=S["TRACE"] + 2, TS1=S["TRACE"] + 3)
    S_LAST = S["TS1"]


def s(n):
    return f"s{S[n]}"


def s2(n):
    return f"s[{S[n]}:{S[n] + 1}]"


class Body:
    def __init__(self):
        self.L, self.lds, self.vm, self.nlabel = [], [], [], 0

    def e(self, t):
        self.L.append(t)

    def label(self, stem):
        self.nlabel += 1
        return f"S2LF_{stem}_{self.nlabel}"

    # ---- LDS operations in flight (lgkmcnt: 4 bits)
    def lds_op(self, text, tag):
        self.e(text)
        self.lds.append(tag)

    def wait_lds(self, tag):
        if tag not in self.lds:
            return
        newer = len(self.lds) - 1 - self.lds.index(tag)
        assert newer <= 15, newer
        self.e(f"s_waitcnt lgkmcnt({newer})")
        self.lds = self.lds[len(self.lds) - newer:] if newer else []

    def wait_all_lds(self):
        if self.lds:
            self.e("s_waitcnt lgkmcnt(0)")
        self.lds = []

    # ---- vector-memory operations in flight (vmcnt: 6 bits; a stricter count is always safe)
    def vm_op(self, text, tag):
        self.e(text)
        self.vm.append(tag)

    def wait_vm(self, tag):
        if tag not in self.vm:
            return
        last = len(self.vm) - 1 - self.vm[::-1].index(tag)      # the NEWEST operation that carries the tag
        newer = len(self.vm) - 1 - last
        if not (EXP & 2 and tag[0] == "D"):
            self.e(f"s_waitcnt vmcnt({min(newer, 63)})")
        self.vm = self.vm[len(self.vm) - newer:] if newer else []

    def wait_all_vm(self):
        self.e("s_waitcnt vmcnt(0)")
        self.vm = []

    def stamp(self, slot, real=False):
        """experiment builds: wave 0's lane 0 stores s_memtime (real: s_memrealtime, the constant 100 MHz clock) to trace[tile * 32 + slot].  Everything in
        flight through LDS is waited for first (scalar loads return out of order with it): the phases are slightly serialised by the measurement."""
        if not TRACE:
            return
        e = self.e
        skip = self.label("notrace")
        self.wait_all_lds()
        e(f"{'s_memrealtime' if real else 's_memtime'} {s2('TS')}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_cmp_eq_u64 {s2('TRACE')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_cmp_eq_u32 {s('WAVE')}, 0")
        e(f"s_cbranch_scc0 {skip}")
        e(f"s_lshl_b32 {s('T6')}, {s('TILE')}, 5")
        e(f"s_add_u32 {s('T6')}, {s('T6')}, {slot}")
        e(f"s_mov_b32 {s('T7')}, 0")
        e(f"s_lshl_b64 {s2('T6')}, {s2('T6')}, 3")
        e(f"s_add_u32 {s('T6')}, {s('TRACE')}, {s('T6')}")
        e(f"s_addc_u32 {s('T7')}, {s('TRACE1')}, {s('T7')}")
        e(f"v_mov_b32 v{V_PIXOFF}, {s('TS')}")
        e(f"v_mov_b32 v{V_TMP}, {s('TS1')}")
        e(f"s_mov_b64 {s2('EX')}, exec")
        e("s_mov_b64 exec, 1")
        e(f"global_store_dwordx2 v{V_ZERO}, v[{V_PIXOFF}:{V_TMP}], {s2('T6')}")      # (never counted, like the rgb store)
        e(f"s_mov_b64 exec, {s2('EX')}")
        e(f"{skip}:")

    # ---- the weight stream
    @staticmethod
    def pos_source(p):
        """(scalar base pair, byte offset) of stream position p (0 .. NREAL-1)"""
        if p < 112:
            return "WB", (p // 16) * 262144 + (p % 16) * PIECE       # layer stride: 16 slabs of 16 KiB
        return "WOUT", (p - 112) * PIECE

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
        """A quad `quad` (0..3) of stream position p -> register set (global quad index) % 4"""
        g = p * 4 + quad
        reg = A_W + 4 * (g % NSETS)
        if quad == 0:
            self.wait_vm(("D", p))                      # the piece has landed
        self.lds_op(f"ds_read_b128 a[{reg}:{reg + 3}], v{V_RINGA} offset:{(p % RING) * PIECE + quad * 1024}", ("A", g))

    # ---- one M-block (or the output layer): 16 quads = stream positions p0 .. p0 + 3
    def block(self, p0, dst, first_c, tucks=None):
        """(LOOK quads ahead: with two, an MFMA chain stood still whenever a piece's 4 KiB landing in LDS delayed its next A read.)
        dst: 'v[..]' / 'a[..]' accumulator quad; first_c: srcC of the first MFMA (an MFMA's vdst and srcC are both VGPRs or both AGPRs: a
        VGPR accumulator takes its bias by four v_accvgpr_read first -- the same value enters the chain); tucks: {quad index: [emitters]} run
        behind that quad's MFMAs (the previous M-block's ReLU + exchange write).  Precondition: the first LOOK A quads from position p0 on are in flight."""
        e = self.e
        if dst.startswith("v[") and first_c.startswith("a["):
            d0, c0 = int(dst[2:].split(":")[0]), int(first_c[2:].split(":")[0])
            for r in range(4):
                e(f"v_accvgpr_read_b32 v{d0 + r}, a{c0 + r}")
            e("s_nop 1")                                # (VALU write -> MFMA srcC)
            first_c = dst
        for q in range(16):
            p, quad = p0 + q // 4, q % 4
            g = p * 4 + quad
            # two quads ahead (into the next block / layer / tile as it comes: the caller says what follows)
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
                if not EXP & 4:
                    e(f"v_mfma_f32_16x16x4_f32 {dst}, a{reg + jj}, v{V_IN + j}, {c}")
            for t in (tucks or {}).get(q, []):
                t()

    def following(self, p0, q):
        """(position, quad) of quad q counted from position p0's first, following the stream into the next tile"""
        p = p0 + q // 4
        if p >= NREAL:
            p = (p - NREAL) % NPOS                      # positions 116 .. 119 do not exist: the next tile's 0 ..
            return (p + NPOS, q % 4)                    # (tagged as the NEXT tile's: + NPOS)
        return (p, q % 4)


def acc(mbl):
    return f"v[{V_ACC + 4 * mbl}:{V_ACC + 4 * mbl + 3}]"


def generate():
    b = Body()
    e = b.e
    # ================= prologue
    for dst, src in (("TILE", "tile0"), ("TEND", "tile_end"), ("NFG", "nfg"), ("FG", "fg0"), ("PG", "pg0"), ("HW", "hw"), ("WAVE", "wave"),
                     ("LDSW", "ldsw")):
        e(f"s_mov_b32 {s(dst)}, %[{src}]")
    for dst, src in (("WB", "wb"), ("WOUT", "wout"), ("P0", "p0"), ("P5", "p5"), ("Q0", "q0"), ("Q5", "q5"), ("OUT", "out"), ("BIAS", "bias")) + ((("TRACE", "trace"),) if TRACE else ()):
        e(f"s_mov_b64 {s2(dst)}, %[{src}]")
    # per-lane addresses from the lane number (one vector operand: the body owns v0 .. v250)
    e(f"v_mov_b32 v{V_TMP}, %[lane]")
    e(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_TMP}")                         # lane * 16: this lane's 16 bytes of a 1-KiB row (pieces, p rows)
    e(f"v_and_b32 v{V_PIX}, 15, v{V_TMP}")                               # sample of the lane
    e(f"v_lshrrev_b32 v{V_QOFF}, 4, v{V_TMP}")
    e(f"v_lshlrev_b32 v{V_QOFF}, 4, v{V_QOFF}")                          # q * 16: q rows hold features 16 mb + 4 q .. + 3
    e(f"s_lshl_b32 {s('T6')}, {s('WAVE')}, 8")
    e(f"v_add_u32 v{V_OWNQ}, {s('T6')}, v{V_QOFF}")                      # ... of the wave's own M-blocks (bias, q5 rows)
    e(f"s_lshl_b32 {s('T6')}, {s('WAVE')}, 12")
    e(f"v_add_u32 v{V_OWNP}, {s('T6')}, v{V_LANE16}")                    # p5 rows of the wave's own M-blocks
    e(f"v_add_u32 v{V_RINGA}, {s('LDSW')}, v{V_LANE16}")                 # A quads are lane-linear in a piece
    e(f"s_add_u32 {s('T7')}, %[lds0], {LDS_X}")
    e(f"v_add_u32 v{V_XR}, {s('T7')}, v{V_LANE16}")                      # the exchange blocks: [M-block][lane][4]
    e(f"v_add_u32 v{V_XW}, {s('T6')}, v{V_XR}")                          # ... this wave's four M-blocks
    e(f"v_mov_b32 v{V_ZERO}, 0")
    # the wave's biases: bias[L][16 (4 w + mbl) + 4 q + r] -> a[A_BIAS + (4 L + mbl) 4 + r]; the output layer's four
    for L in range(NLAYERS):
        if L:
            e(f"s_add_u32 {s('BIAS')}, {s('BIAS')}, 1024")
            e(f"s_addc_u32 {s('BIAS1')}, {s('BIAS1')}, 0")
        for mbl in range(4):
            r = A_BIAS + (4 * L + mbl) * 4
            b.vm_op(f"global_load_dwordx4 a[{r}:{r + 3}], v{V_OWNQ}, {s2('BIAS')} offset:{mbl * 64}", ("B", L, mbl))
    e(f"s_add_u32 {s('BIAS')}, {s('BIAS')}, 1024")      # OFF_BOUT follows the seven bias rows
    e(f"s_addc_u32 {s('BIAS1')}, {s('BIAS1')}, 0")
    b.vm_op(f"global_load_dwordx4 a[{A_BOUT}:{A_BOUT + 3}], v{V_ZERO}, {s2('BIAS')}", ("B", 7, 0))

    def table_rows(fg, pg):
        """T4 = &p0[pg], T8 = &q0[fg] (16 KiB / 1 KiB rows)"""
        e(f"s_mov_b32 {s('T6')}, {s(pg)}")
        e(f"s_mov_b32 {s('T7')}, 0")
        e(f"s_lshl_b64 {s2('T6')}, {s2('T6')}, 14")
        e(f"s_add_u32 {s('T4')}, {s('P0')}, {s('T6')}")
        e(f"s_addc_u32 {s('T5')}, {s('P01')}, {s('T7')}")
        e(f"s_mov_b32 {s('T6')}, {s(fg)}")
        e(f"s_mov_b32 {s('T7')}, 0")
        e(f"s_lshl_b64 {s2('T6')}, {s2('T6')}, 10")
        e(f"s_add_u32 {s('T8')}, {s('Q0')}, {s('T6')}")
        e(f"s_addc_u32 {s('T9')}, {s('Q01')}, {s('T7')}")

    def load_p0q0(fg, pg):
        table_rows(fg, pg)
        for mb in range(16):
            if mb and mb % 4 == 0:
                e(f"s_add_u32 {s('T4')}, {s('T4')}, 4096")
                e(f"s_addc_u32 {s('T5')}, {s('T5')}, 0")
            b.vm_op(f"global_load_dwordx4 v[{V_P0 + 4 * mb}:{V_P0 + 4 * mb + 3}], v{V_LANE16}, {s2('T4')} offset:{(mb % 4) * 1024}", ("P0", mb))
        for mb in range(16):
            b.vm_op(f"global_load_dwordx4 v[{V_Q0 + 4 * mb}:{V_Q0 + 4 * mb + 3}], v{V_QOFF}, {s2('T8')} offset:{mb * 64}", ("Q0", mb))

    def load_p5q5(fg, pg):
        e(f"s_mov_b32 {s('T6')}, {s(pg)}")
        e(f"s_mov_b32 {s('T7')}, 0")
        e(f"s_lshl_b64 {s2('T6')}, {s2('T6')}, 14")
        e(f"s_add_u32 {s('T4')}, {s('P5')}, {s('T6')}")
        e(f"s_addc_u32 {s('T5')}, {s('P51')}, {s('T7')}")
        e(f"s_mov_b32 {s('T6')}, {s(fg)}")
        e(f"s_mov_b32 {s('T7')}, 0")
        e(f"s_lshl_b64 {s2('T6')}, {s2('T6')}, 10")
        e(f"s_add_u32 {s('T8')}, {s('Q5')}, {s('T6')}")
        e(f"s_addc_u32 {s('T9')}, {s('Q51')}, {s('T7')}")
        for mbl in range(4):
            b.vm_op(f"global_load_dwordx4 v[{V_P5 + 4 * mbl}:{V_P5 + 4 * mbl + 3}], v{V_OWNP}, {s2('T4')} offset:{mbl * 1024}", ("P5", mbl))
        for mbl in range(4):
            b.vm_op(f"global_load_dwordx4 v[{V_Q5 + 4 * mbl}:{V_Q5 + 4 * mbl + 3}], v{V_OWNQ}, {s2('T8')} offset:{mbl * 64}", ("Q5", mbl))

    load_p0q0("FG", "PG")
    load_p5q5("FG", "PG")
    for p in range(RING):
        b.dma(p)
    # loop-entry state: only the eight pieces may still be on their way
    b.wait_vm(("Q5", 3))
    b.vm = [t for t in b.vm if t[0] == "D"]
    assert len(b.vm) == 4 * RING
    vm_entry = list(b.vm)

    def head():
        """the tile (FG, PG): the next one's coordinates, h0 = relu(q0[frame] + p0[pixel]) for all 256 features in every wave (the tables are
        this tile's: requested a tile ago), and the request for the next tile's tables"""
        e(f"s_add_u32 {s('FGN')}, {s('FG')}, 1")
        e(f"s_cmp_eq_u32 {s('FGN')}, {s('NFG')}")
        e(f"s_cselect_b32 {s('FGN')}, 0, {s('FGN')}")
        e(f"s_cselect_b32 {s('T6')}, 1, 0")
        e(f"s_add_u32 {s('PGN')}, {s('PG')}, {s('T6')}")
        e(f"s_add_u32 {s('T7')}, {s('TILE')}, 1")
        e(f"s_cmp_lt_u32 {s('T7')}, {s('TEND')}")
        e(f"s_cselect_b32 {s('FGN')}, {s('FGN')}, {s('FG')}")      # (the range's last tile: its own rows again -- valid addresses, unused data)
        e(f"s_cselect_b32 {s('PGN')}, {s('PGN')}, {s('PG')}")
        e("s_nop 7")                                               # (the previous tile's last MFMAs have read in[])
        for j in range(0, 64, 2):
            e(f"v_pk_add_f32 v[{V_IN + j}:{V_IN + j + 1}], v[{V_Q0 + j}:{V_Q0 + j + 1}], v[{V_P0 + j}:{V_P0 + j + 1}]")
        for j in range(64):
            e(f"v_max_f32 v{V_IN + j}, 0, v{V_IN + j}")
        load_p0q0("FGN", "PGN")

    head()
    for q in range(LOOK):
        b.a_read(q // 4, q % 4)
    e("S2LF_BODY:")
    vm_body, lds_body = list(b.vm), list(b.lds)
    b.stamp(18, real=True)
    b.stamp(0)

    def relu_write(L, mbl, skip):
        """ReLU of M-block mbl of layer L and its 1 KiB of the exchange block L & 1 ([M-block][lane][4]); the skip step first"""
        def run():
            a0 = V_ACC + 4 * mbl
            if skip:      # pts_linears[5]: (acc + q5[frame]) + p5[pixel], then the ReLU (gen_render_body.py q_step / p_step)
                b.wait_vm(("Q5", 3))
                for h in (0, 2):
                    e(f"v_pk_add_f32 v[{a0 + h}:{a0 + h + 1}], v[{a0 + h}:{a0 + h + 1}], v[{V_Q5 + 4 * mbl + h}:{V_Q5 + 4 * mbl + h + 1}]")
                for h in (0, 2):
                    e(f"v_pk_add_f32 v[{a0 + h}:{a0 + h + 1}], v[{a0 + h}:{a0 + h + 1}], v[{V_P5 + 4 * mbl + h}:{V_P5 + 4 * mbl + h + 1}]")
            for r in range(4):
                e(f"v_max_f32 v{a0 + r}, 0, v{a0 + r}")
            if not EXP & 1:
                b.lds_op(f"ds_write_b128 v{V_XW}, v[{a0}:{a0 + 3}] offset:{(L & 1) * 16384 + mbl * 1024}", ("X", L, mbl))
        return run

    for L in range(NLAYERS):
        for mbl in range(4):
            p0 = L * 16 + mbl * 4
            bias = A_BIAS + (4 * L + mbl) * 4
            tucks = {}
            if mbl:      # the previous M-block's results are final once this block's second quad has been issued (> 128 MFMA cycles later)
                tucks[1] = [relu_write(L, mbl - 1, L == 4)]
            b.block(p0, acc(mbl), f"a[{bias}:{bias + 3}]", tucks)
        # the layer's last M-block: nothing to hide behind
        e("s_nop 7")
        e("s_nop 7")
        b.stamp(1 + 2 * L)
        relu_write(L, 3, L == 4)()
        if L == 4:
            load_p5q5("FGN", "PGN")            # the next tile's skip rows
        b.wait_all_lds()                       # (this wave's four writes have landed; the next block's first two A quads too)
        if not EXP & 1:
            e("s_barrier")
        for mb in range(16 if not EXP & 1 else 0):
            b.lds_op(f"ds_read_b128 v[{V_IN + 4 * mb}:{V_IN + 4 * mb + 3}], v{V_XR} offset:{(L & 1) * 16384 + mb * 1024}", ("H", mb))
            if mb == 13:                       # (lgkmcnt counts to 15)
                b.wait_lds(("H", 0))
        b.wait_all_lds()
        e("s_nop 1")
        b.stamp(2 + 2 * L)
    # ---- output layer (3 rows padded to one M-block), every wave (only wave 0 stores): positions 112 .. 115
    b.block(112, f"a[{A_RGB}:{A_RGB + 3}]", f"a[{A_BOUT}:{A_BOUT + 3}]")
    for p in range(NPOS + 4, NPOS + RING):     # positions 116 .. 119 hold nothing: the next tile's 4 .. 7 take their slots now
        b.dma(p)
    b.stamp(15)
    # ---- store: wave 0, lanes 0 .. 15 hold rgb of pixel 16 pg + lane of frame fg
    skip = b.label("nostore")
    e(f"s_cmp_eq_u32 {s('WAVE')}, 0")
    e(f"s_cbranch_scc0 {skip}")
    e(f"s_lshl_b32 {s('T6')}, {s('PG')}, 4")
    e(f"v_add_u32 v{V_TMP}, {s('T6')}, v{V_PIX}")
    e(f"v_cmp_gt_u32 vcc, {s('HW')}, v{V_TMP}")
    e(f"v_mul_u32_u24 v{V_PIXOFF}, 12, v{V_TMP}")
    e(f"s_mov_b64 {s2('EX')}, exec")
    e("s_nop 3")
    e("s_and_b64 exec, vcc, 0xffff")
    e(f"s_mul_hi_u32 {s('T9')}, {s('FG')}, {s('HW')}")     # byte offset of the frame = frame * hw * 12 (64-bit)
    e(f"s_mul_i32 {s('T8')}, {s('FG')}, {s('HW')}")
    e(f"s_mul_i32 {s('T9')}, {s('T9')}, 12")
    e(f"s_mul_hi_u32 {s('T7')}, {s('T8')}, 12")
    e(f"s_mul_i32 {s('T8')}, {s('T8')}, 12")
    e(f"s_add_u32 {s('T9')}, {s('T9')}, {s('T7')}")
    e(f"s_add_u32 {s('T4')}, {s('OUT')}, {s('T8')}")
    e(f"s_addc_u32 {s('T5')}, {s('OUT1')}, {s('T9')}")
    e("s_nop 7")
    e("s_nop 7")
    e(f"global_store_dwordx3 v{V_PIXOFF}, a[{A_RGB}:{A_RGB + 2}], {s2('T4')}")      # (never counted: see the module text)
    e(f"s_mov_b64 exec, {s2('EX')}")
    e(f"{skip}:")
    # ---- next tile
    b.wait_vm(("Q5", 3))                       # (the next tile's skip rows, requested ten blocks ago: keeps the loop's state simple)
    b.stamp(16)
    b.stamp(17, real=True)
    e(f"s_add_u32 {s('TILE')}, {s('TILE')}, 1")
    e(f"s_mov_b32 {s('FG')}, {s('FGN')}")
    e(f"s_mov_b32 {s('PG')}, {s('PGN')}")
    e(f"s_cmp_lt_u32 {s('TILE')}, {s('TEND')}")
    e("s_cbranch_scc0 S2LF_END")
    head()
    # the loop's state: the next tile's pieces (tagged + NPOS), its tables' requests and its first two A quads
    b.vm = [(t[0], t[1] - NPOS) if t[0] == "D" else t for t in b.vm]
    b.lds = [(t[0], t[1] - NPOS * 4) for t in b.lds]
    assert b.vm == vm_body, (b.vm[:6], vm_body[:6], len(b.vm), len(vm_body))
    assert TRACE or b.lds == lds_body, (b.lds, lds_body)      # (a trace build has waited for everything: the body's waits are stricter than needed)
    e("s_branch S2LF_BODY")
    e("S2LF_END:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")      # run-ahead requests must land before the workgroup's LDS is released
    return b


OPERANDS = """      :
      : [tile0] "s"(tile0), [tile_end] "s"(tile_end), [nfg] "s"(nfg), [fg0] "s"(fg0), [pg0] "s"(pg0), [hw] "s"(a.hw), [wave] "s"(wave), [ldsw] "s"(ldsw),
        [wb] "s"(wb), [wout] "s"(wout), [p0] "s"(a.p0t), [p5] "s"(a.p5t), [q0] "s"(a.q0), [q5] "s"(a.q5), [out] "s"(a.out), [bias] "s"(biasp),
        [lds0] "s"(lds0), [lane] "v"(lane)TRACE_OPERAND
"""


def main(path):
    b = generate()
    lines = b.L
    clob = [f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)] + [f"s{r}" for r in range(36, S_LAST + 1)]
    clob += ["vcc", "scc", "memory"]
    out = ["// GENERATED by csrc/gen_render_fs_body.py -- do not edit; the generator is the source.", "asm volatile("]
    out += [f'    "{x}\\n\\t"' for x in lines]
    out.append(OPERANDS.rstrip("\n").replace("TRACE_OPERAND", ', [trace] "s"(g_trace)' if TRACE else ""))
    out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return len(lines)


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"render body, feature-split: {main(os.path.join(d, 'render_fs_body.inc'))} instructions")
