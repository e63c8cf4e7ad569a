"""Generates conv16_body.inc: the body of s2l::conv16_asm_kernel (csrc/conv16.hip) -- the split-bf16 3x3 convolution of the U-Net's
inference speed mode as ONE fixed-register gfx950 assembly text (the renderer's recipe: one persistent workgroup per CU, one wave per
SIMD, the chunk loop owned instruction by instruction).

Arithmetic = conv3x3_split_kernel's (csrc/unet.hip), operation for operation and in its order, so the outputs are THE SAME BITS
(tests/test_gpu_unet_kernels.py, tools/cmp_conv16.py): fp32 activations split on the way into LDS as hi = bf16(x),
lo = bf16(x - hi); the BatchNorm-folded weights pre-split by s2l_unet_pack16x3 ([cout/64][cin/16][tap 9][part 2][block 2]
[lane 64][8]); per 16-channel chunk and tap three products on v_mfma_f32_32x32x16_bf16, smallest first: W_lo x_hi, W_hi x_lo,
W_hi x_hi; accumulators start at the bias; ReLU; fp32 NHWC out.

Shape of the work (v4):
  * tile = 32 rows x 16 columns x 64 output channels, FOUR waves (one per SIMD); wave w owns rows 8 w .. 8 w + 7 as four N-blocks
    of 2 rows x 16 pixels, and both 32-channel M-blocks: 8 accumulator blocks of 16 AGPRs; per tap 4 + 8 ds_read_b128 (into AGPRs:
    both operand sets live there) feed 24 MFMAs of 32 cycles: 0.5 operand reads per MFMA (the C++ kernel: eight waves each reading
    all of a chunk's weights, 0.67);
  * why 32 x 16: three earlier forms of this kernel used 16 x 16 tiles and were bound by the TEXTURE PATH -- per chunk and CU
    36 KiB of weight DMA + 20 KiB of halo loads (+ 64 KiB of stores per tile) at ~16 B/clk is as long as the chunk's 3 456 MFMA
    cycles; ablation builds (S2L_C16_EXP) without any ONE of the three ran 15 % faster, without all three 25 %.  The taller tile
    halves the weight bytes per MFMA and trims the halo overlap (1.27 -> 1.20): -30 % texture-path work per MFMA;
  * TWO LDS buffers of (halo 34 x 18 pixels x 64 B [hi 16 ch | lo 16 ch], the 16-byte segment index XORed with (column >> 2) & 3:
    the sixteen pixels of an operand read cover all 64 banks) + (weights 36 KiB, by LDS-DMA: nine 1-KiB pieces per wave);
  * a workgroup's chunks form ONE stream across its tiles, staged one chunk ahead: during chunk g the weights of g + 1 arrive by
    LDS-DMA, its halo values (fp32, ten 16-byte loads per lane) are requested behind taps 0..2 and split + written to the other
    buffer behind taps 6..8 (16 VALU + 2 ds_write per quad; a filler beside a bf16 MFMA costs about half of what it costs alone);
    ONE barrier per chunk.  (Requesting two chunks ahead, v2 / v3, bought nothing: the loads' cost is issue, not latency.)
  * a tile's end: the accumulators leave through this wave's 4 KiB of LDS (buffer 1's halo area is idle then) so that a store
    instruction writes eight whole 128-byte lines (as the MFMA leaves them, 16 bytes to each of 64 lines: measured 6 x the cost);
    ReLU on the way; the next tile's bias comes from a table in LDS straight into the AGPRs.

Register map (per wave): a0-127 acc[mb][nb][16] | a128-175, a176-223 two operand sets (A[part][mb] 4 x 4, B[part][nb] 8 x 4) |
v0-39 staged halo quads | v40.. per-lane constants and temporaries (names below).
Scalars: s16.. (names below); kernel arguments are loaded from the kernarg segment (struct Conv16Args, csrc/conv16.h)."""
import os
import sys

TILE_H = 32
HALO_ROWS = TILE_H + 2
HALO_BYTES, W_BYTES = HALO_ROWS * 18 * 64, 9 * 2 * 2 * 64 * 16  # 39 168 + 36 864
BUF = HALO_BYTES + W_BYTES                                      # 76 032
BIAS_OFF = 2 * BUF                                              # 256 floats
LDS_BYTES = BIAS_OFF + 1024
STG_OFF = BUF                                                   # store staging = the start of buffer 1's halo area: 4 waves x 4 KiB
NQ = 10                                                         # halo quads per lane: 34 * 18 * 4 = 2448 = 9 * 256 + 144
CONST_WORDS = 34                                                # per lane, from the C++ prologue: whi[10], hrc[10], bofs[6], swa[4], sra[4]
STORE_MOD = os.environ.get("S2L_C16_STORE_MOD", "")
EXP = int(os.environ.get("S2L_C16_EXP", "0"))                  # ablation builds (results wrong): 1 no stores, 2 no halo loads, 4 no weight DMA, 8 no commit

# ---- registers
A_ACC = 0
A_OPS = (128, 176)              # operand set: A[pt][mb] at + (pt * 2 + mb) * 4, B[pt][nb] at + 16 + (pt * 4 + nb) * 4
A_LAST = 223
V_PIN = 0                       # 10 quads x 4
V_WHI = (40, 50)                # LDS addresses, one set per buffer (DS immediates are 16 bits)
V_WLO = (60, 70)
V_BOFS = (80, 86)
V_AOFS = (92, 93)
V_HRC = 94                      # 10: halo (row << 8 | col) of the quads
V_PIX = 104                     # 10: pixel index inside the frame of the quads of the STAGING stream's tile (0 where invalid)
V_MSK = 114                     # 10: all ones / zero, same tile
V_VOFF = 124                    # 10: byte offset of the quads from the chunk's source pointer
V_LANE, V_TID, V_HH16, V_C4X16, V_DMA, V_BIASA, V_L3, V_L7X16 = 134, 135, 136, 137, 138, 139, 140, 141
V_T = 144                       # temporaries 144..175
V_SOFF = 176                    # 16: store byte offsets [nb][j] of the tile that just ended (-1: no store for this lane)
V_SWA = 192                     # 4: staging write addresses per register quad rq
V_SRA = 196                     # 4: staging read addresses per store j
V_LAST = 199


def _scalars(first, singles, pairs, skip=(32, 33)):
    """names -> SGPR numbers (s32 / s33 are reserved by the compiler on this target: never on the clobber list)"""
    m, r = {}, first
    for n in pairs:
        while (r & 1) or r in skip or (r + 1) in skip:
            r += 1
        m[n], m[n + "1"] = r, r + 1
        r += 2
    for n in singles:
        while r in skip:
            r += 1
        m[n] = r
        r += 1
    return m


# two streams walk the workgroup's tiles: compute ("" prefix) and staging, one chunk ahead (N)
S = _scalars(16, singles=("CA CB COUT H W TILESX TILESY NCT NCH CHA WAVE LDS0 RELU CBYTES LDSW T0 T1 T2 T3 "
                          "TX TY CT FR X0 Y0 CC NTL LEFT NTX NTY NCT_ NFR NC NLEFT SX0 SY0").split(),
             pairs=("KARG", "INA", "INB", "W16", "WB", "BIAS", "OUT", "SRC", "WCH", "WCH1_", "WCH2_", "FRA", "FRB", "OUTF", "EX", "M9", "TA"))
S_LAST = max(S.values())
assert S_LAST <= 100, S_LAST

# byte offsets of the fields of struct Conv16Args (csrc/conv16.hip static_asserts them)
ARG = {"inA": 0, "inB": 8, "w16": 16, "bias": 24, "out": 32, "CA": 40, "CB": 44, "cout": 48, "H": 52, "W": 56, "tiles_x": 60,
       "tiles_y": 64, "n_ct": 68, "relu": 72}


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
        return f"S2LK_{stem}_{self.nlabel}"

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
        self.e("s_waitcnt lgkmcnt(0)")
        self.lds = []

    def emit_group(self, g):
        for it in g:
            if isinstance(it, tuple) and it[0] == "wait":
                self.wait_lds(it[1])
            elif isinstance(it, tuple):
                self.lds_op(it[0], it[1])
            else:
                self.e(it)

    # ---- registers
    @staticmethod
    def acc(mb, nb):
        b = A_ACC + (mb * 4 + nb) * 16
        return f"a[{b}:{b + 15}]"

    @staticmethod
    def opa(os_, pt, mb):
        return A_OPS[os_] + (pt * 2 + mb) * 4

    @staticmethod
    def opb(os_, pt, nb):
        return A_OPS[os_] + 16 + (pt * 4 + nb) * 4

    # ---- operand reads of tap t from buffer `buf` into operand set os_: 12 (text, tag), in the order the MFMAs need them:
    # A lo (2), B hi (4) [first product], B lo (4) [second], A hi (2) [third]
    def tap_reads(self, t, buf, os_):
        dy, dx = t // 3, t % 3
        out = []
        for kind in range(4):
            for blk in range(2 if kind in (0, 3) else 4):
                if kind in (0, 3):
                    pt = 1 if kind == 0 else 0
                    r = self.opa(os_, pt, blk)
                    off = ((t * 2 + pt) * 2 + blk) * 1024      # (V_AOFS points at the buffer's weight area)
                    out.append((f"ds_read_b128 a[{r}:{r + 3}], v{V_AOFS[buf]} offset:{off}", ("R", t, kind, blk)))
                else:
                    pt = 0 if kind == 1 else 1
                    r = self.opb(os_, pt, blk)
                    off = (2 * blk + dy) * 18 * 64
                    out.append((f"ds_read_b128 a[{r}:{r + 3}], v{V_BOFS[buf] + dx * 2 + pt} offset:{off}", ("R", t, kind, blk)))
        return out

    def tap_mfmas(self, os_, sprinkle):
        """24 MFMAs of one tap on operand set os_; sprinkle: 24 lists of items tucked behind MFMA m"""
        m = 0
        for g in range(3):
            pa, pb = (1 if g == 0 else 0), (1 if g == 1 else 0)
            for mb in range(2):
                for nb in range(4):
                    a, b = self.opa(os_, pa, mb), self.opb(os_, pb, nb)
                    self.e(f"v_mfma_f32_32x32x16_f16 {self.acc(mb, nb)}, a[{a}:{a + 3}], a[{b}:{b + 3}], {self.acc(mb, nb)}")
                    self.emit_group(sprinkle[m])
                    m += 1

    # ---- streams
    def next_coords(self, p, ct_name):
        """the stream with prefix p moves to its next tile (x fastest, then y, channel tile, frame) -- or stays on its last one"""
        e = self.e
        stay = self.label("stay")
        e(f"s_cmp_lt_u32 {s(p + 'LEFT')}, 2")
        e(f"s_cbranch_scc1 {stay}")
        e(f"s_sub_u32 {s(p + 'LEFT')}, {s(p + 'LEFT')}, 1")
        e(f"s_add_u32 {s(p + 'TX')}, {s(p + 'TX')}, 1")
        e(f"s_cmp_eq_u32 {s(p + 'TX')}, {s('TILESX')}")
        e(f"s_cselect_b32 {s(p + 'TX')}, 0, {s(p + 'TX')}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s(p + 'TY')}, {s(p + 'TY')}, {s('T0')}")
        e(f"s_cmp_eq_u32 {s(p + 'TY')}, {s('TILESY')}")
        e(f"s_cselect_b32 {s(p + 'TY')}, 0, {s(p + 'TY')}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s(ct_name)}, {s(ct_name)}, {s('T0')}")
        e(f"s_cmp_eq_u32 {s(ct_name)}, {s('NCT')}")
        e(f"s_cselect_b32 {s(ct_name)}, 0, {s(ct_name)}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s(p + 'FR')}, {s(p + 'FR')}, {s('T0')}")
        e(f"{stay}:")

    def geometry(self):
        """per-lane geometry of the staging stream's tile at (SX0, SY0): pixel index and validity mask of the ten quads"""
        e = self.e
        for i in range(NQ):
            r, c, t = V_T, V_T + 1, V_T + 2
            e(f"v_lshrrev_b32 v{r}, 8, v{V_HRC + i}")
            e(f"v_and_b32 v{c}, 0xff, v{V_HRC + i}")
            e(f"v_add_u32 v{r}, {s('SY0')}, v{r}")                     # gy + 1
            e(f"v_add_u32 v{c}, {s('SX0')}, v{c}")                     # gx + 1
            e(f"v_subrev_u32 v{r}, 1, v{r}")                            # gy (wraps below 0: fails the unsigned test)
            e(f"v_subrev_u32 v{c}, 1, v{c}")
            e(f"v_cmp_gt_u32 vcc, {s('H')}, v{r}")
            e(f"v_cmp_gt_u32 {s2('TA')}, {s('W')}, v{c}")
            e(f"s_and_b64 vcc, vcc, {s2('TA')}")
            if i == NQ - 1:
                e(f"s_and_b64 vcc, vcc, {s2('M9')}")
            e(f"v_mul_lo_u32 v{t}, v{r}, {s('W')}")
            e(f"v_add_u32 v{t}, v{t}, v{c}")
            e(f"v_cndmask_b32 v{V_PIX + i}, 0, v{t}, vcc")
            e(f"v_cndmask_b32 v{V_MSK + i}, 0, -1, vcc")

    def staging_tile_setup(self):
        """geometry registers, frame bases FRA / FRB and the weight base WB for the staging stream's tile (NTX, NTY, NCT_, NFR)"""
        e = self.e
        e(f"s_lshl_b32 {s('SX0')}, {s('NTX')}, 4")
        e(f"s_mul_i32 {s('SY0')}, {s('NTY')}, {TILE_H}")
        self.geometry()
        e(f"s_mul_i32 {s('T0')}, {s('H')}, {s('W')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('NFR')}")               # pixels before this frame (< 2^31: the launcher checks)
        for fr, src, c in (("FRA", "INA", "CA"), ("FRB", "INB", "CB")):
            e(f"s_mul_i32 {s('T1')}, {s(c)}, 4")
            e(f"s_mul_hi_u32 {s('T3')}, {s('T0')}, {s('T1')}")
            e(f"s_mul_i32 {s('T2')}, {s('T0')}, {s('T1')}")
            e(f"s_add_u32 {s(fr)}, {s(src)}, {s('T2')}")
            e(f"s_addc_u32 {s(fr + '1')}, {s(src + '1')}, {s('T3')}")
        e(f"s_mul_i32 {s('T0')}, {s('NCT_')}, {s('NCH')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {W_BYTES}")
        e(f"s_add_u32 {s('WB')}, {s('W16')}, {s('T0')}")
        e(f"s_addc_u32 {s('WB1')}, {s('W161')}, 0")

    def staging_source(self):
        """for chunk NC of the staging tile: SRC (channel 0 of the chunk at pixel 0 of the frame), CBYTES (bytes per pixel of that
        tensor), WCH / WCH1_ / WCH2_ (the chunk's weights, + 4 KiB, + 8 KiB)"""
        e = self.e
        e(f"s_cmp_lt_u32 {s('NC')}, {s('CHA')}")                       # chunk from A?
        e(f"s_cselect_b64 {s2('SRC')}, {s2('FRA')}, {s2('FRB')}")
        e(f"s_cselect_b32 {s('CBYTES')}, {s('CA')}, {s('CB')}")
        e(f"s_cselect_b32 {s('T0')}, 0, {s('CHA')}")
        e(f"s_sub_u32 {s('T0')}, {s('NC')}, {s('T0')}")                # chunk index inside its tensor
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 6")                        # * 16 channels * 4 bytes
        e(f"s_add_u32 {s('SRC')}, {s('SRC')}, {s('T0')}")
        e(f"s_addc_u32 {s('SRC1')}, {s('SRC1')}, 0")
        e(f"s_lshl_b32 {s('CBYTES')}, {s('CBYTES')}, 2")
        e(f"s_mul_i32 {s('T0')}, {s('NC')}, {W_BYTES}")
        e(f"s_add_u32 {s('WCH')}, {s('WB')}, {s('T0')}")
        e(f"s_addc_u32 {s('WCH1')}, {s('WB1')}, 0")
        e(f"s_add_u32 {s('WCH1_')}, {s('WCH')}, 4096")
        e(f"s_addc_u32 {s('WCH1_1')}, {s('WCH1')}, 0")
        e(f"s_add_u32 {s('WCH2_')}, {s('WCH')}, 8192")
        e(f"s_addc_u32 {s('WCH2_1')}, {s('WCH1')}, 0")

    def advance_staging(self):
        e = self.e
        same = self.label("nsame")
        e(f"s_add_u32 {s('NC')}, {s('NC')}, 1")
        e(f"s_cmp_lt_u32 {s('NC')}, {s('NCH')}")
        e(f"s_cbranch_scc1 {same}")
        e(f"s_mov_b32 {s('NC')}, 0")
        self.next_coords("N", "NCT_")
        self.staging_tile_setup()
        e(f"{same}:")

    # ---- instruction groups
    def dma_items(self, wbuf):
        """nine 1-KiB pieces of the staging chunk's weights -> weight area of buffer wbuf"""
        items = []
        for grp in range(3):
            base = ("WCH", "WCH1_", "WCH2_")[grp]
            for j in range(4 if grp < 2 else 1):
                g = []
                if j == 0:
                    g += [f"s_add_u32 m0, {s('LDSW')}, {wbuf * BUF + grp * 4096}", "s_nop 0"]
                if not EXP & 4:
                    g += [f"global_load_lds_dwordx4 v{V_DMA}, {s2(base)} offset:{1024 * j}"]
                items.append(g)
        return items

    def halo_items(self):
        """ten 16-byte loads of the staging chunk's halo values"""
        items = []
        for i in range(NQ):
            g = [f"v_mul_lo_u32 v{V_VOFF + i}, v{V_PIX + i}, {s('CBYTES')}", f"v_add_u32 v{V_VOFF + i}, v{V_VOFF + i}, v{V_C4X16}"]
            if i == NQ - 1:
                g += [f"s_mov_b64 exec, {s2('M9')}"]
            if not EXP & 2:
                g += [f"global_load_dwordx4 v[{V_PIN + 4 * i}:{V_PIN + 4 * i + 3}], v{V_VOFF + i}, {s2('SRC')}"]
            if i == NQ - 1:
                g += ["s_mov_b64 exec, -1"]
            items.append(g)
        return items

    def commit_items(self, buf):
        """split + LDS write of the staged quads into the halo area of buffer buf"""
        items = []
        for i in range(NQ):
            p = V_PIN + 4 * i
            h, l, t = V_T, V_T + 2, V_T + 4
            if i % 2:
                h, l, t = V_T + 8, V_T + 10, V_T + 12
            g1 = [f"v_and_b32 v{p + j}, v{p + j}, v{V_MSK + i}" for j in range(4)]
            # parts as IEEE halves, round toward zero (csrc/unet.hip: pack_f16x2): hi = f16(x), lo = f16(x - hi)
            g2 = [f"v_cvt_pkrtz_f16_f32 v{h}, v{p}, v{p + 1}", f"v_cvt_pkrtz_f16_f32 v{h + 1}, v{p + 2}, v{p + 3}",
                  f"v_cvt_f32_f16 v{t}, v{h}", f"v_cvt_f32_f16_sdwa v{t + 1}, v{h} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"]
            g3 = [f"v_cvt_f32_f16 v{t + 2}, v{h + 1}", f"v_cvt_f32_f16_sdwa v{t + 3}, v{h + 1} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
                  f"v_sub_f32 v{p}, v{p}, v{t}", f"v_sub_f32 v{p + 1}, v{p + 1}, v{t + 1}"]
            g4 = [f"v_sub_f32 v{p + 2}, v{p + 2}, v{t + 2}", f"v_sub_f32 v{p + 3}, v{p + 3}, v{t + 3}",
                  f"v_cvt_f16_f32 v{p}, v{p}", f"v_cvt_f16_f32 v{p + 1}, v{p + 1}"]      # the lo parts round to nearest (unbiased)
            g4b = [f"v_cvt_f16_f32 v{p + 2}, v{p + 2}", f"v_cvt_f16_f32 v{p + 3}, v{p + 3}",
                   f"v_pack_b32_f16 v{l}, v{p}, v{p + 1}", f"v_pack_b32_f16 v{l + 1}, v{p + 2}, v{p + 3}"]
            wr = []
            if i == NQ - 1:
                wr += [f"s_mov_b64 exec, {s2('M9')}"]
            wr += [(f"ds_write_b64 v{V_WHI[buf] + i}, v[{h}:{h + 1}]", ("W", i, 0)),
                   (f"ds_write_b64 v{V_WLO[buf] + i}, v[{l}:{l + 1}]", ("W", i, 1))]
            if i == NQ - 1:
                wr += ["s_mov_b64 exec, -1"]
            items += [g1, g2, g3, g4, g4b, wr] if not EXP & 8 else []
        return items

    def store_tile(self):
        """the stores of the tile that just ended (offsets V_SOFF, base OUTF: set by tile_end), ReLU on the way.  A block
        (mb, nb) = 32 channels x 32 pixels leaves through this wave's 4 KiB of LDS staging: transposed, lane l writes channel quad
        l & 7 of pixel 8 j + (l >> 3): eight whole 128-byte lines per store instruction."""
        e = self.e
        ta, tb = V_T, V_T + 16
        for nb in range(4):
            for mb in range(2):
                blk = nb * 2 + mb
                a0 = A_ACC + (mb * 4 + nb) * 16
                for rq in range(4):
                    for j in range(4):
                        e(f"v_accvgpr_read_b32 v{ta + 4 * rq + j}, a{a0 + 4 * rq + j}")
                    for j in range(4):
                        e(f"v_max_f32 v{ta + 4 * rq + j}, {s('RELU')}, v{ta + 4 * rq + j}")
                    self.lds_op(f"ds_write_b128 v{V_SWA + rq}, v[{ta + 4 * rq}:{ta + 4 * rq + 3}]", ("SW", blk, rq))
                for j in range(4):
                    self.lds_op(f"ds_read_b128 v[{tb + 4 * j}:{tb + 4 * j + 3}], v{V_SRA + j}", ("SR", blk, j))
                for j in range(4):
                    self.wait_lds(("SR", blk, j))
                    e(f"v_cmp_ne_u32 vcc, -1, v{V_SOFF + nb * 4 + j}")
                    e("s_nop 0")
                    e("s_and_b64 exec, vcc, exec")
                    if not EXP & 1:
                        e(f"global_store_dwordx4 v{V_SOFF + nb * 4 + j}, v[{tb + 4 * j}:{tb + 4 * j + 3}], {s2('OUTF')} offset:{mb * 128}" + STORE_MOD)
                    e("s_mov_b64 exec, -1")
        self.wait_all_lds()

    # ---- one chunk of the compute stream: reads buffer p, stages the next chunk into buffer p ^ 1
    def chunk(self, p):
        e = self.e
        self.staging_source()
        dma = self.dma_items(p ^ 1)
        req = self.halo_items()
        com = self.commit_items(p ^ 1)
        for text, tag in self.tap_reads(0, p, 0):        # operands of tap 0 (exposed after the barrier)
            self.lds_op(text, tag)
        for t in range(9):
            os_ = t & 1
            nxt = self.tap_reads(t + 1, p, os_ ^ 1) if t + 1 < 9 else []
            sprinkle = [[] for _ in range(24)]
            for m, rd in enumerate(nxt):
                sprinkle[m].append(rd)
            # the chunk's vector-memory instructions: 9 DMA pieces behind the second half of tap 0, the 10 halo requests behind taps
            # 1 and 2.  Measured alternatives (16 frames 500 x 500, whole forward): everything one per MFMA from tap 0 on 7.53 ms; one
            # every 4 / 6 / 8 / 10 MFMAs over taps 0..4 7.24 - 7.32 ms; this placement 7.05 - 7.09 ms.
            if t == 0:
                for k, g in enumerate(dma):
                    sprinkle[12 + k].extend(g)
            if t in (1, 2):
                for k, g in enumerate(req[(t - 1) * 5:t * 5]):
                    sprinkle[12 + 2 * k].extend(g)
            if t >= 6 and com:                           # the 60 commit groups behind taps 6..8: 20 each
                start = [0, 20, 40][t - 6]
                for k, g in enumerate(com[start:start + 20]):
                    sprinkle[3 + k].extend(g)
            if t == 6:
                e("s_waitcnt vmcnt(0)")                  # the staged values (requested four taps ago) and the weight pieces have landed
            self.wait_lds(("R", t, 3, 1))
            self.tap_mfmas(os_, sprinkle)
        self.wait_all_lds()
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        self.advance_staging()

    def bias_init(self):
        """acc = bias of (CT, mb, channels (r & 3) + 8 (r >> 2) + 4 hh), straight into the AGPRs of the four N-blocks"""
        e = self.e
        e(f"s_lshl_b32 {s('T0')}, {s('CT')}, 8")                        # CT * 64 floats
        e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_BIASA}")
        for mb in range(2):
            for nb in range(4):
                for rq in range(4):
                    b = A_ACC + (mb * 4 + nb) * 16 + 4 * rq
                    self.lds_op(f"ds_read_b128 a[{b}:{b + 3}], v{V_T} offset:{mb * 128 + rq * 32}", ("BI", mb, nb, rq))
                if nb % 2:
                    self.wait_all_lds()

    def tile_end(self):
        """the finished tile's store state: OUTF and, per N-block and store j, this lane's byte offset (pixel 8 j + (lane >> 3) of
        the block, channel quad lane & 7) or -1 outside the image; the stores; then the compute stream moves on"""
        e = self.e
        e(f"s_mul_i32 {s('T0')}, {s('H')}, {s('W')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('FR')}")
        e(f"s_lshl_b32 {s('T1')}, {s('COUT')}, 2")
        e(f"s_mul_hi_u32 {s('T3')}, {s('T0')}, {s('T1')}")
        e(f"s_mul_i32 {s('T2')}, {s('T0')}, {s('T1')}")
        e(f"s_add_u32 {s('OUTF')}, {s('OUT')}, {s('T2')}")
        e(f"s_addc_u32 {s('OUTF1')}, {s('OUT1')}, {s('T3')}")
        e(f"s_lshl_b32 {s('T0')}, {s('CT')}, 8")
        e(f"s_add_u32 {s('OUTF')}, {s('OUTF')}, {s('T0')}")
        e(f"s_addc_u32 {s('OUTF1')}, {s('OUTF1')}, 0")
        e(f"s_lshl_b32 {s('X0')}, {s('TX')}, 4")
        e(f"s_mul_i32 {s('Y0')}, {s('TY')}, {TILE_H}")
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 3")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('Y0')}")                # first row of this wave
        for half in range(2):                                           # columns 8 half + (lane >> 3)
            e(f"v_add_u32 v{V_T + half}, {s('X0')}, v{V_L3}")
            if half:
                e(f"v_add_u32 v{V_T + half}, 8, v{V_T + half}")
        for nb in range(4):
            for j in range(4):
                row, half = 2 * nb + (j >> 1), j & 1
                d = V_SOFF + nb * 4 + j
                e(f"s_add_u32 {s('T2')}, {s('T0')}, {row}")             # gy (wave-uniform)
                e(f"s_cmp_lt_u32 {s('T2')}, {s('H')}")
                e(f"s_cselect_b64 {s2('TA')}, -1, 0")
                e(f"v_cmp_gt_u32 vcc, {s('W')}, v{V_T + half}")
                e(f"s_and_b64 vcc, vcc, {s2('TA')}")
                e(f"s_mul_i32 {s('T3')}, {s('T2')}, {s('W')}")
                e(f"v_add_u32 v{V_T + 2}, {s('T3')}, v{V_T + half}")    # pixel index
                e(f"v_mul_lo_u32 v{V_T + 2}, v{V_T + 2}, {s('T1')}")    # * COUT * 4
                e(f"v_add_u32 v{V_T + 2}, v{V_T + 2}, v{V_L7X16}")
                e(f"v_cndmask_b32 v{d}, -1, v{V_T + 2}, vcc")
        e("s_nop 7")
        e("s_nop 7")                                                  # (MFMA results -> v_accvgpr_read)
        self.store_tile()
        e("s_barrier")                                                # (the staging area is buffer 1's halo region: nobody may commit the next
        self.next_coords("", "CT")                                    #  tile's chunk 1 into it while another wave still stores)


def generate():
    b = Body()
    e = b.e
    # ================= prologue
    e(f"s_mov_b64 {s2('KARG')}, %[karg]")
    for dst, src in (("WAVE", "wave"), ("LDS0", "lds0"), ("TX", "tx0"), ("TY", "ty0"), ("CT", "ct0"), ("FR", "fr0"), ("NTL", "ntl")):
        e(f"s_mov_b32 {s(dst)}, %[{src}]")
    for dst, field in (("INA", "inA"), ("INB", "inB"), ("W16", "w16"), ("BIAS", "bias"), ("OUT", "out")):
        e(f"s_load_dwordx2 {s2(dst)}, {s2('KARG')}, {ARG[field]}")
    for dst, field in (("CA", "CA"), ("CB", "CB"), ("COUT", "cout"), ("H", "H"), ("W", "W"), ("TILESX", "tiles_x"), ("TILESY", "tiles_y"),
                       ("NCT", "n_ct"), ("T3", "relu")):
        e(f"s_load_dword {s(dst)}, {s2('KARG')}, {ARG[field]}")
    e(f"v_mov_b32 v{V_TID}, %[tid]")
    e(f"v_and_b32 v{V_LANE}, 63, v{V_TID}")
    # per-lane constants: the C++ prologue left CONST_WORDS words per lane at the start of LDS ([word][256 threads])
    e(f"v_lshlrev_b32 v{V_T}, 2, v{V_TID}")
    e(f"v_add_u32 v{V_T}, {s('LDS0')}, v{V_T}")
    for i in range(NQ):
        e(f"ds_read_b32 v{V_WHI[0] + i}, v{V_T} offset:{1024 * i}")
    e("s_waitcnt lgkmcnt(0)")
    for i in range(NQ):
        e(f"ds_read_b32 v{V_HRC + i}, v{V_T} offset:{1024 * (10 + i)}")
    e("s_waitcnt lgkmcnt(0)")
    for i in range(6):
        e(f"ds_read_b32 v{V_BOFS[0] + i}, v{V_T} offset:{1024 * (20 + i)}")
    for i in range(4):
        e(f"ds_read_b32 v{V_SWA + i}, v{V_T} offset:{1024 * (26 + i)}")
        e(f"ds_read_b32 v{V_SRA + i}, v{V_T} offset:{1024 * (30 + i)}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                                    # (everybody has read its constants: the buffers may be written)
    for i in range(NQ):
        e(f"v_xor_b32 v{V_WLO[0] + i}, 32, v{V_WHI[0] + i}")
        e(f"v_add_u32 v{V_WHI[1] + i}, {BUF}, v{V_WHI[0] + i}")
        e(f"v_add_u32 v{V_WLO[1] + i}, {BUF}, v{V_WLO[0] + i}")
    for i in range(6):
        e(f"v_add_u32 v{V_BOFS[1] + i}, {BUF}, v{V_BOFS[0] + i}")
    e(f"v_lshlrev_b32 v{V_T}, 4, v{V_LANE}")                           # lane * 16
    e(f"v_lshrrev_b32 v{V_HH16}, 5, v{V_LANE}")
    e(f"v_lshlrev_b32 v{V_HH16}, 4, v{V_HH16}")                        # hh * 16 bytes
    e(f"v_lshrrev_b32 v{V_L3}, 3, v{V_LANE}")
    e(f"v_and_b32 v{V_L7X16}, 7, v{V_LANE}")
    e(f"v_lshlrev_b32 v{V_L7X16}, 4, v{V_L7X16}")
    e(f"v_and_b32 v{V_C4X16}, 3, v{V_TID}")
    e(f"v_lshlrev_b32 v{V_C4X16}, 4, v{V_C4X16}")                      # (tid & 3) * 16 bytes
    e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, 9216")
    e(f"v_add_u32 v{V_DMA}, {s('T0')}, v{V_T}")                         # this lane's 16 B of the wave's nine 1-KiB pieces (global offset)
    e(f"s_add_u32 {s('T1')}, {s('LDS0')}, {HALO_BYTES}")
    e(f"v_add_u32 v{V_AOFS[0]}, {s('T1')}, v{V_T}")                      # A reads: the buffer's weight area + lane * 16
    e(f"v_add_u32 v{V_AOFS[1]}, {BUF}, v{V_AOFS[0]}")
    e(f"s_add_u32 {s('LDSW')}, {s('LDS0')}, {HALO_BYTES}")
    e(f"s_add_u32 {s('LDSW')}, {s('LDSW')}, {s('T0')}")                 # LDS address of this wave's weight pieces in buffer 0
    e(f"s_add_u32 {s('T0')}, {s('LDS0')}, {BIAS_OFF}")
    e(f"v_add_u32 v{V_BIASA}, {s('T0')}, v{V_HH16}")                    # bias table + hh * 16
    # quad 9 exists for threads 0..143: waves 0 and 1, lanes 0..15 of wave 2
    e(f"s_cmp_lt_u32 {s('WAVE')}, 2")
    e(f"s_cselect_b64 {s2('M9')}, -1, 0")
    e(f"s_cmp_eq_u32 {s('WAVE')}, 2")
    e(f"s_cselect_b64 {s2('TA')}, 0xffff, 0")
    e(f"s_or_b64 {s2('M9')}, {s2('M9')}, {s2('TA')}")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_add_u32 {s('NCH')}, {s('CA')}, {s('CB')}")
    e(f"s_lshr_b32 {s('NCH')}, {s('NCH')}, 4")
    e(f"s_lshr_b32 {s('CHA')}, {s('CA')}, 4")
    e(f"s_cmp_eq_u32 {s('T3')}, 0")
    e(f"s_cselect_b32 {s('RELU')}, 0xff800000, 0")                    # lower bound of the stores' v_max: -inf (linear) or 0
    # the bias table -> LDS: thread t (< 64 * NCT) copies bias[t]
    e(f"s_lshl_b32 {s('T0')}, {s('NCT')}, 6")
    e(f"v_cmp_gt_u32 vcc, {s('T0')}, v{V_TID}")
    e(f"v_lshlrev_b32 v{V_T}, 2, v{V_TID}")
    e(f"v_mov_b32 v{V_T + 1}, 0")
    e(f"s_mov_b64 {s2('EX')}, exec")
    e("s_and_b64 exec, exec, vcc")
    e(f"global_load_dword v{V_T + 1}, v{V_T}, {s2('BIAS')}")
    e(f"s_mov_b64 exec, {s2('EX')}")
    e("s_waitcnt vmcnt(0)")
    e(f"s_add_u32 {s('T0')}, {s('LDS0')}, {BIAS_OFF}")
    e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_T}")
    e(f"ds_write_b32 v{V_T}, v{V_T + 1}")
    # the staging stream starts on the workgroup's first tile
    e(f"s_mov_b32 {s('LEFT')}, {s('NTL')}")
    e(f"s_mov_b32 {s('NLEFT')}, {s('NTL')}")
    for n_, c_ in (("NTX", "TX"), ("NTY", "TY"), ("NCT_", "CT"), ("NFR", "FR")):
        e(f"s_mov_b32 {s(n_)}, {s(c_)}")
    e(f"s_mov_b32 {s('NC')}, 0")
    b.staging_tile_setup()
    # chunk 0 -> buffer 0, all exposed (once per workgroup)
    b.staging_source()
    for g in b.dma_items(0):
        b.emit_group(g)
    for g in b.halo_items():
        b.emit_group(g)
    e("s_waitcnt vmcnt(0)")
    for g in b.commit_items(0):
        b.emit_group(g)
    b.advance_staging()
    b.wait_all_lds()
    e("s_barrier")

    # ================= tile loop
    e("S2LK_TILE:")
    b.bias_init()
    e(f"s_mov_b32 {s('CC')}, 0")
    e("S2LK_PAIR:")
    b.chunk(0)
    b.chunk(1)
    e(f"s_add_u32 {s('CC')}, {s('CC')}, 2")
    e(f"s_cmp_lt_u32 {s('CC')}, {s('NCH')}")
    e("s_cbranch_scc1 S2LK_PAIR")
    b.tile_end()
    e(f"s_sub_u32 {s('NTL')}, {s('NTL')}, 1")
    e(f"s_cmp_gt_u32 {s('NTL')}, 0")
    e("s_cbranch_scc1 S2LK_TILE")
    e("s_waitcnt vmcnt(0)")
    return b.L


OPERANDS = """      :
      : [karg] "s"(karg), [wave] "s"(wave), [lds0] "s"(lds0), [tx0] "s"(tx0), [ty0] "s"(ty0), [ct0] "s"(ct0), [fr0] "s"(fr0), [ntl] "s"(ntl),
        [tid] "v"(tid)
"""


def main(objdir):
    lines = generate()
    clob = [f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)] + [f"s{r}" for r in range(16, S_LAST + 1) if r not in (32, 33)]
    clob += ["vcc", "scc", "memory"]
    out = ["// GENERATED by csrc/gen_conv16_body.py -- do not edit; the generator is the source.", "asm volatile("]
    out += [f'    "{x}\\n\\t"' for x in lines]
    out.append(OPERANDS.rstrip("\n"))
    out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
    with open(os.path.join(objdir, "conv16_body.inc"), "w") as f:
        f.write("\n".join(out) + "\n")
    return len(lines)


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"conv16 body: {main(d)} instructions")
