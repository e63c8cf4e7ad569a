"""Generates conv16_body.inc: the body of s2l::conv16_asm_kernel (csrc/conv16.hip) -- the split-bf16 3x3 convolution of the U-Net's
inference speed mode as ONE fixed-register gfx950 assembly text (the renderer's recipe: one persistent workgroup per CU, one wave per
SIMD, the chunk loop owned instruction by instruction).

Arithmetic = conv3x3_split_kernel's (csrc/unet.hip), operation for operation and in its order, so the outputs are THE SAME BITS
(tests/test_gpu_unet_kernels.py): fp32 activations split on the way into LDS as hi = bf16(x), lo = bf16(x - hi); the
BatchNorm-folded weights pre-split by s2l_unet_pack16x3 ([cout/64][cin/16][tap 9][part 2][block 2][lane 64][8]); per 16-channel
chunk and tap three products on v_mfma_f32_32x32x16_bf16, smallest first: W_lo x_hi, W_hi x_lo, W_hi x_hi; accumulators start at
the bias; ReLU; fp32 NHWC out.

What differs from the C++ kernel is the shape of the work (it was bound by LDS reads: eight waves each reading all of a chunk's
weights, 0.67 operand reads per MFMA, matrix pipe busy 0.66):
  * tile = 16 x 16 pixels x 64 output channels, FOUR waves (one per SIMD); wave w owns rows 4 w .. 4 w + 3 as two N-blocks of
    2 rows x 16 pixels, and both 32-channel M-blocks: 4 accumulator blocks of 16 registers; per tap 4 + 4 ds_read_b128 feed 12
    MFMAs of 32 cycles: 21 B/clk per wave, 85 of the LDS's 128 B/clk per CU;
  * TWO buffers of (halo 18 x 18 pixels x 64 B [hi 16 ch | lo 16 ch], the 16-byte segment index XORed with (column >> 2) & 3:
    the sixteen pixels of an operand read cover all 64 banks) + (weights 36 KiB, by LDS-DMA: nine 1-KiB pieces per wave);
  * during chunk c: the weights of c + 1 are requested behind the first MFMAs, the halo values of c + 1 (fp32, six 16-byte loads per
    lane) too; behind taps 5..8 they are split and written to the other buffer (16 VALU + 2 ds_write per quad, two per MFMA slot:
    a filler beside a bf16 MFMA costs about half of what it costs alone); ONE barrier per chunk;
  * the next TILE's first chunk is staged during the current tile's last chunk (the geometry registers are free by then), so only a
    workgroup's very first chunk is exposed;
  * a tile's end: ReLU, the accumulators leave through 16 global_store_dwordx4 per lane (32 bytes per pixel and instruction);
    the next tile's bias comes from a table in LDS.

Register map (per wave): v0-63 acc[mb][nb][16] | v64-95, v96-127 two operand sets (A[part][mb], B[part][nb]: 8 x 4) |
v128-151 staged halo quads of the next chunk | v152.. per-lane constants and temporaries (names below).
Scalars: s8.. (names below); kernel arguments are loaded from the kernarg segment (struct Conv16Args, csrc/conv16.hip)."""
import os
import sys

HALO_BYTES, W_BYTES = 18 * 18 * 64, 9 * 2 * 2 * 64 * 16        # 20 736 + 36 864
BUF = HALO_BYTES + W_BYTES                                      # 57 600
BIAS_OFF = 2 * BUF                                              # 256 floats
LDS_BYTES = BIAS_OFF + 1024
NQ = 6                                                          # halo quads per lane: 18 * 18 * 4 = 1296 = 5 * 256 + 16
EXP = int(os.environ.get("S2L_C16_EXP", "0"))                  # ablation builds (results wrong): 1 no stores, 2 no halo loads, 4 no weight DMA, 8 no commit

# ---- vector registers
V_ACC = 0
V_OPS = (64, 96)                # operand set: A[pt][mb] at + (pt * 2 + mb) * 4, B[pt][nb] at + 16 + (pt * 2 + nb) * 4
V_PIN = 128                     # 6 quads x 4
# LDS addresses, one set per buffer (DS immediates are 16 bits: buffer 1 starts at 57 600)
V_WHI = (152, 171)              # 6: write address of the quads' hi halves
V_WLO = (158, 177)              # 6: ... lo halves (= hi ^ 32)
V_BOFS = (164, 183)             # 6: B-read addresses [dx][pt]
V_AOFS = (170, 189)             # A-read address (LDS base + halo + lane * 16 folded into the immediate: base + lane * 16)
V_HRC = 190                     # 6: halo (row << 8 | col) of the quads
V_PIX = 196                     # 6: pixel index inside the frame of the quads (0 where invalid)
V_MSK = 202                     # 6: all ones / zero
V_VOFF = 208                    # 6: byte offset of the quads from the chunk's source pointer
V_LANE, V_COL, V_ROW2, V_HH16, V_C4X16, V_DMA, V_TID = 214, 215, 216, 217, 218, 219, 220
V_T = 222                       # temporaries 222..241 (even: 64-bit tuples must be aligned)
V_SOFF = 242                    # 2: store byte offsets per N-block
V_LAST = 243
CONST_WORDS = 18                # per lane, written to LDS by the C++ prologue: whi[6] (absolute, buffer 0), hrc[6], bofs[6] (absolute, buffer 0)

def _scalars(first, singles, pairs):
    m, r = {}, first
    for n in pairs:
        r += r & 1
        m[n], m[n + "1"] = r, r + 1
        r += 2
    for n in singles:
        m[n] = r
        r += 1
    return m


S = _scalars(16, singles="CA CB COUT H W TILESX TILESY NCT NTL NCH CHA WAVE LDS0 TX TY CT FR X0 Y0 SX0 SY0 CC RELU T0 T1 T2 T3 CBYTES NTX NTY NCTN NFR LDSW TGT".split(),
             pairs=("KARG", "INA", "INB", "W16", "WB", "BIAS", "OUT", "SRC", "WCH", "WCH1_", "WCH2_", "FRA", "FRB", "OUTF", "EX", "M5", "TA", "TB"))
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

    # ---- registers
    @staticmethod
    def acc(mb, nb):
        b = V_ACC + (mb * 2 + nb) * 16
        return f"v[{b}:{b + 15}]"

    @staticmethod
    def opa(os_, pt, mb):
        b = V_OPS[os_] + (pt * 2 + mb) * 4
        return b

    @staticmethod
    def opb(os_, pt, nb):
        b = V_OPS[os_] + 16 + (pt * 2 + nb) * 4
        return b

    # ---- operand reads of tap t from buffer `buf` into operand set os_: list of 8 (text, tag), in the order the MFMAs need them:
    # A lo, B hi (first product), B lo (second), A hi (third)
    def tap_reads(self, t, buf, os_):
        dy, dx = t // 3, t % 3
        out = []
        for kind in range(4):
            for blk in range(2):
                if kind in (0, 3):
                    pt = 1 if kind == 0 else 0
                    r = self.opa(os_, pt, blk)
                    off = HALO_BYTES + ((t * 2 + pt) * 2 + blk) * 1024
                    out.append((f"ds_read_b128 v[{r}:{r + 3}], v{V_AOFS[buf]} offset:{off}", ("R", t, kind, blk)))
                else:
                    pt = 0 if kind == 1 else 1
                    r = self.opb(os_, pt, blk)
                    off = (2 * blk + dy) * 18 * 64
                    out.append((f"ds_read_b128 v[{r}:{r + 3}], v{V_BOFS[buf] + dx * 2 + pt} offset:{off}", ("R", t, kind, blk)))
        return out

    def tap_mfmas(self, os_, sprinkle):
        """12 MFMAs of one tap on operand set os_; sprinkle: list of 12 lists of items tucked behind MFMA m (item: text or (text, tag))"""
        for m in range(12):
            g, mb, nb = m >> 2, (m >> 1) & 1, m & 1
            pa, pb = (1 if g == 0 else 0), (1 if g == 1 else 0)
            a, b = self.opa(os_, pa, mb), self.opb(os_, pb, nb)
            self.e(f"v_mfma_f32_32x32x16_bf16 {self.acc(mb, nb)}, v[{a}:{a + 3}], v[{b}:{b + 3}], {self.acc(mb, nb)}")
            for it in sprinkle[m]:
                if isinstance(it, tuple):
                    self.lds_op(it[0], it[1])
                else:
                    self.e(it)

    # ---- staging of one chunk: requests (global loads + weight DMA) and the split + LDS commit
    def chunk_source(self):
        """scalar set-up for the chunk S[TGT] (0 .. NCH-1) of the tile whose frame bases are FRA / FRB and whose channel tile is the
        one WB was set for: SRC = pointer to channel 0 of the chunk at pixel 0 of the frame, CBYTES = bytes per pixel of that tensor,
        WCH / WCH1_ / WCH2_ = the chunk's weights (+ 4 KiB, + 8 KiB)."""
        e = self.e
        e(f"s_cmp_lt_u32 {s('TGT')}, {s('CHA')}")                      # chunk from A?
        e(f"s_cselect_b64 {s2('SRC')}, {s2('FRA')}, {s2('FRB')}")
        e(f"s_cselect_b32 {s('CBYTES')}, {s('CA')}, {s('CB')}")
        e(f"s_cselect_b32 {s('T0')}, 0, {s('CHA')}")
        e(f"s_sub_u32 {s('T0')}, {s('TGT')}, {s('T0')}")               # chunk index inside its tensor
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 6")                        # * 16 channels * 4 bytes
        e(f"s_add_u32 {s('SRC')}, {s('SRC')}, {s('T0')}")
        e(f"s_addc_u32 {s('SRC1')}, {s('SRC1')}, 0")
        e(f"s_lshl_b32 {s('CBYTES')}, {s('CBYTES')}, 2")
        e(f"s_mul_i32 {s('T0')}, {s('TGT')}, {W_BYTES}")
        e(f"s_add_u32 {s('WCH')}, {s('WB')}, {s('T0')}")
        e(f"s_addc_u32 {s('WCH1')}, {s('WB1')}, 0")
        e(f"s_add_u32 {s('WCH1_')}, {s('WCH')}, 4096")
        e(f"s_addc_u32 {s('WCH1_1')}, {s('WCH1')}, 0")
        e(f"s_add_u32 {s('WCH2_')}, {s('WCH')}, 8192")
        e(f"s_addc_u32 {s('WCH2_1')}, {s('WCH1')}, 0")

    def request_items(self, buf):
        """the chunk's requests as a list of instruction groups (each group goes behind one MFMA): 6 x (offset, load) and 9 DMA pieces"""
        items = []
        for i in range(NQ):
            g = [f"v_mul_lo_u32 v{V_VOFF + i}, v{V_PIX + i}, {s('CBYTES')}", f"v_add_u32 v{V_VOFF + i}, v{V_VOFF + i}, v{V_C4X16}"]
            if i == NQ - 1:
                g += [f"s_mov_b64 exec, {s2('M5')}"]
            if not EXP & 2:
                g += [f"global_load_dwordx4 v[{V_PIN + 4 * i}:{V_PIN + 4 * i + 3}], v{V_VOFF + i}, {s2('SRC')}"]
            if i == NQ - 1:
                g += ["s_mov_b64 exec, -1"]
            items.append(g)
        for grp in range(3):
            base = ("WCH", "WCH1_", "WCH2_")[grp]
            for j in range(4 if grp < 2 else 1):
                g = []
                if j == 0:
                    g += [f"s_add_u32 m0, {s('LDSW')}, {buf * BUF + grp * 4096}", "s_nop 0"]
                if not EXP & 4:
                    g += [f"global_load_lds_dwordx4 v{V_DMA}, {s2(base)} offset:{1024 * j}"]
                items.append(g)
        return items

    def commit_items(self, buf):
        """split + LDS write of the six staged quads into buffer `buf`: a list of small instruction groups"""
        items = []
        for i in range(NQ):
            p = V_PIN + 4 * i
            h, l, t = V_T, V_T + 2, V_T + 4
            if i % 2:
                h, l, t = V_T + 8, V_T + 10, V_T + 12
            g1 = [f"v_and_b32 v{p + j}, v{p + j}, v{V_MSK + i}" for j in range(4)]
            g2 = [f"v_cvt_pk_bf16_f32 v{h}, v{p}, v{p + 1}", f"v_cvt_pk_bf16_f32 v{h + 1}, v{p + 2}, v{p + 3}",
                  f"v_lshlrev_b32 v{t}, 16, v{h}", f"v_and_b32 v{t + 1}, 0xffff0000, v{h}"]
            g3 = [f"v_lshlrev_b32 v{t + 2}, 16, v{h + 1}", f"v_and_b32 v{t + 3}, 0xffff0000, v{h + 1}",
                  f"v_sub_f32 v{p}, v{p}, v{t}", f"v_sub_f32 v{p + 1}, v{p + 1}, v{t + 1}"]
            g4 = [f"v_sub_f32 v{p + 2}, v{p + 2}, v{t + 2}", f"v_sub_f32 v{p + 3}, v{p + 3}, v{t + 3}",
                  f"v_cvt_pk_bf16_f32 v{l}, v{p}, v{p + 1}", f"v_cvt_pk_bf16_f32 v{l + 1}, v{p + 2}, v{p + 3}"]
            wr = []
            if i == NQ - 1:
                wr += [f"s_mov_b64 exec, {s2('M5')}"]
            wr += [(f"ds_write_b64 v{V_WHI[buf] + i}, v[{h}:{h + 1}]", ("W", i, 0)),
                   (f"ds_write_b64 v{V_WLO[buf] + i}, v[{l}:{l + 1}]", ("W", i, 1))]
            if i == NQ - 1:
                wr += ["s_mov_b64 exec, -1"]
            items += [g1, g2, g3, g4, wr] if not EXP & 8 else []
        return items

    def emit_group(self, g):
        for it in g:
            if isinstance(it, tuple):
                self.lds_op(it[0], it[1])
            else:
                self.e(it)

    def stage_now(self, buf):
        """prologue form: request, wait, commit -- everything exposed (a workgroup's first chunk only)"""
        self.chunk_source()
        for g in self.request_items(buf):
            self.emit_group(g)
        self.e("s_waitcnt vmcnt(0)")
        for g in self.commit_items(buf):
            self.emit_group(g)

    def geometry(self):
        """per-lane staging geometry of the tile at (SX0, SY0): pixel index and validity mask of the six quads"""
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
                e(f"s_and_b64 vcc, vcc, {s2('M5')}")
            e(f"v_mul_lo_u32 v{t}, v{r}, {s('W')}")
            e(f"v_add_u32 v{t}, v{t}, v{c}")
            e(f"v_cndmask_b32 v{V_PIX + i}, 0, v{t}, vcc")
            e(f"v_cndmask_b32 v{V_MSK + i}, 0, -1, vcc")

    def frame_bases(self, fr_name):
        """FRA / FRB = inA / inB + frame * H * W * C * 4 of the tile being STAGED"""
        e = self.e
        e(f"s_mul_i32 {s('T0')}, {s('H')}, {s('W')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s(fr_name)}")                # pixels before this frame (< 2^31: the launcher checks)
        for fr, src, c in (("FRA", "INA", "CA"), ("FRB", "INB", "CB")):
            e(f"s_mul_i32 {s('T1')}, {s(c)}, 4")
            e(f"s_mul_hi_u32 {s('T3')}, {s('T0')}, {s('T1')}")
            e(f"s_mul_i32 {s('T2')}, {s('T0')}, {s('T1')}")
            e(f"s_add_u32 {s(fr)}, {s(src)}, {s('T2')}")
            e(f"s_addc_u32 {s(fr + '1')}, {s(src + '1')}, {s('T3')}")

    def tile_weights(self, ct_name):
        """WB = w16 + channel tile * NCH * W_BYTES of the tile being STAGED"""
        e = self.e
        e(f"s_mul_i32 {s('T0')}, {s(ct_name)}, {s('NCH')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {W_BYTES}")
        e(f"s_add_u32 {s('WB')}, {s('W16')}, {s('T0')}")
        e(f"s_addc_u32 {s('WB1')}, {s('W161')}, 0")

    def next_tile_coords(self):
        """(NTX, NTY, NCTN, NFR) = the tile after (TX, TY, CT, FR): x fastest, then y, channel tile, frame"""
        e = self.e
        e(f"s_add_u32 {s('NTX')}, {s('TX')}, 1")
        e(f"s_cmp_eq_u32 {s('NTX')}, {s('TILESX')}")
        e(f"s_cselect_b32 {s('NTX')}, 0, {s('NTX')}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s('NTY')}, {s('TY')}, {s('T0')}")
        e(f"s_cmp_eq_u32 {s('NTY')}, {s('TILESY')}")
        e(f"s_cselect_b32 {s('NTY')}, 0, {s('NTY')}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s('NCTN')}, {s('CT')}, {s('T0')}")
        e(f"s_cmp_eq_u32 {s('NCTN')}, {s('NCT')}")
        e(f"s_cselect_b32 {s('NCTN')}, 0, {s('NCTN')}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s('NFR')}, {s('FR')}, {s('T0')}")

    # ---- one chunk: reads buffer `buf`; stage: None, or "next" = request + commit the chunk S[TGT] into the other buffer
    def chunk(self, buf, stage):
        e = self.e
        if stage:
            self.chunk_source()                          # (scalar; for the chunk S[TGT] of the staging tile)
        req = self.request_items(buf ^ 1) if stage else []
        com = self.commit_items(buf ^ 1) if stage else []
        # operands of tap 0 (exposed after the barrier; 8 reads)
        for text, tag in self.tap_reads(0, buf, 0):
            self.lds_op(text, tag)
        for t in range(9):
            os_ = t & 1
            nxt = self.tap_reads(t + 1, buf, os_ ^ 1) if t + 1 < 9 else []
            sprinkle = [[] for _ in range(12)]
            for m, rd in enumerate(nxt):
                sprinkle[m].append(rd)
            if t == 0 and req:                           # the 15 request groups: 8 behind tap 0, 7 behind tap 1
                for k, g in enumerate(req[:8]):
                    sprinkle[k + 2].extend(g)
            if t == 1 and req:
                for k, g in enumerate(req[8:]):
                    sprinkle[k + 2].extend(g)
            if t >= 5 and com:                           # the 30 commit groups behind taps 5..8: 8 + 8 + 7 + 7
                per = [8, 8, 7, 7][t - 5]
                start = [0, 8, 16, 23][t - 5]
                for k, g in enumerate(com[start:start + per]):
                    sprinkle[k + 2].extend(g)
            if t == 5 and com:
                e("s_waitcnt vmcnt(0)")                  # the staged values (requested ~ 2 k cycles ago) and the weight pieces have landed
            # this tap's operands: the newest 8 reads (issued behind the previous tap) -- wait for all of them
            self.wait_lds(("R", t, 3, 1))
            self.tap_mfmas(os_, sprinkle)
        self.wait_all_lds()
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")

    def bias_init(self):
        """acc = bias of (CT, mb, channels (r & 3) + 8 (r >> 2) + 4 hh): 8 reads for nb = 0, copied to nb = 1"""
        e = self.e
        e(f"s_lshl_b32 {s('T0')}, {s('CT')}, 8")                        # CT * 64 floats
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDS0')}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {BIAS_OFF}")
        e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_HH16}")
        for mb in range(2):
            for rq in range(4):
                b = V_ACC + (mb * 2) * 16 + 4 * rq
                self.lds_op(f"ds_read_b128 v[{b}:{b + 3}], v{V_T} offset:{mb * 128 + rq * 32}", ("BI", mb, rq))
        self.wait_all_lds()
        for mb in range(2):
            for r in range(16):
                e(f"v_mov_b32 v{V_ACC + (mb * 2 + 1) * 16 + r}, v{V_ACC + (mb * 2) * 16 + r}")

    def epilogue(self):
        """ReLU (RELU = 0.0 or -inf as the lower bound) and the stores of the tile at (TX, TY, CT, FR)"""
        e = self.e
        for r in range(64):
            e(f"v_max_f32 v{V_ACC + r}, {s('RELU')}, v{V_ACC + r}")
        # OUTF = out + ((FR * H * W) * COUT + CT * 64) * 4
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
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 2")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('Y0')}")                # first row of this wave
        e(f"v_add_u32 v{V_T + 1}, {s('X0')}, v{V_COL}")                # gx
        e(f"v_cmp_gt_u32 {s2('TB')}, {s('W')}, v{V_T + 1}")
        e(f"s_mov_b64 {s2('EX')}, exec")
        for nb in range(2):
            e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_ROW2}")               # gy of N-block 0: row0 + (n >> 4)
            if nb:
                e(f"v_add_u32 v{V_T}, 2, v{V_T}")
            e(f"v_cmp_gt_u32 vcc, {s('H')}, v{V_T}")
            e(f"s_and_b64 vcc, vcc, {s2('TB')}")
            e(f"v_mul_lo_u32 v{V_T + 2}, v{V_T}, {s('W')}")
            e(f"v_add_u32 v{V_T + 2}, v{V_T + 2}, v{V_T + 1}")          # pixel index
            e(f"v_mul_lo_u32 v{V_SOFF + nb}, v{V_T + 2}, {s('T1')}")    # * COUT * 4
            e(f"v_add_u32 v{V_SOFF + nb}, v{V_SOFF + nb}, v{V_HH16}")
            e("s_nop 1")
            e("s_and_b64 exec, exec, vcc")
            for mb in range(2):
                for rq in range(4):
                    b = V_ACC + (mb * 2 + nb) * 16 + 4 * rq
                    if not EXP & 1:
                        e(f"global_store_dwordx4 v{V_SOFF + nb}, v[{b}:{b + 3}], {s2('OUTF')} offset:{mb * 128 + rq * 32}")
            e(f"s_mov_b64 exec, {s2('EX')}")
        e("s_nop 2")


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
        e(f"ds_read_b32 v{V_HRC + i}, v{V_T} offset:{1024 * (6 + i)}")
        e(f"ds_read_b32 v{V_BOFS[0] + i}, v{V_T} offset:{1024 * (12 + i)}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                                    # (everybody has read its constants: the buffers may be written)
    for i in range(NQ):
        e(f"v_xor_b32 v{V_WLO[0] + i}, 32, v{V_WHI[0] + i}")
        e(f"v_add_u32 v{V_WHI[1] + i}, {BUF}, v{V_WHI[0] + i}")
        e(f"v_add_u32 v{V_WLO[1] + i}, {BUF}, v{V_WLO[0] + i}")
        e(f"v_add_u32 v{V_BOFS[1] + i}, {BUF}, v{V_BOFS[0] + i}")
    e(f"v_lshlrev_b32 v{V_T}, 4, v{V_LANE}")                           # lane * 16
    e(f"v_and_b32 v{V_COL}, 15, v{V_LANE}")
    e(f"v_bfe_u32 v{V_ROW2}, v{V_LANE}, 4, 1")                         # (n >> 4) & 1
    e(f"v_lshrrev_b32 v{V_HH16}, 5, v{V_LANE}")
    e(f"v_lshlrev_b32 v{V_HH16}, 4, v{V_HH16}")                        # hh * 16 bytes
    e(f"v_and_b32 v{V_C4X16}, 3, v{V_TID}")
    e(f"v_lshlrev_b32 v{V_C4X16}, 4, v{V_C4X16}")                      # (tid & 3) * 16 bytes
    e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, 9216")
    e(f"v_add_u32 v{V_DMA}, {s('T0')}, v{V_T}")                         # this lane's 16 B of the wave's nine 1-KiB pieces (global offset)
    e(f"v_add_u32 v{V_AOFS[0]}, {s('LDS0')}, v{V_T}")                   # A reads: LDS base + lane * 16 (+ the halo region as an immediate)
    e(f"v_add_u32 v{V_AOFS[1]}, {BUF}, v{V_AOFS[0]}")
    e(f"s_add_u32 {s('LDSW')}, {s('LDS0')}, {HALO_BYTES}")
    e(f"s_add_u32 {s('LDSW')}, {s('LDSW')}, {s('T0')}")                 # LDS address of this wave's weight pieces in buffer 0
    e(f"s_cmp_eq_u32 {s('WAVE')}, 0")                                  # quad 5 exists for lanes 0..15 of wave 0 only
    e(f"s_cselect_b64 {s2('M5')}, 0xffff, 0")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_add_u32 {s('NCH')}, {s('CA')}, {s('CB')}")
    e(f"s_lshr_b32 {s('NCH')}, {s('NCH')}, 4")
    e(f"s_lshr_b32 {s('CHA')}, {s('CA')}, 4")
    e(f"s_cmp_eq_u32 {s('T3')}, 0")
    e(f"s_cselect_b32 {s('RELU')}, 0xff800000, 0")                    # lower bound of the epilogue's v_max: -inf (linear) or 0
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
    # first tile: geometry, pointers, chunk 0 staged synchronously
    e(f"s_lshl_b32 {s('X0')}, {s('TX')}, 4")
    e(f"s_lshl_b32 {s('Y0')}, {s('TY')}, 4")
    e(f"s_mov_b32 {s('SX0')}, {s('X0')}")
    e(f"s_mov_b32 {s('SY0')}, {s('Y0')}")
    b.geometry()
    b.frame_bases("FR")
    b.tile_weights("CT")
    e(f"s_mov_b32 {s('TGT')}, 0")
    b.stage_now(0)
    b.wait_all_lds()
    e("s_barrier")

    # ================= tile loop
    e("S2LK_TILE:")
    b.bias_init()
    e(f"s_mov_b32 {s('CC')}, 0")
    e("S2LK_PAIR:")
    # chunk CC (even) reads buffer 0 and stages CC + 1 of this tile into buffer 1
    e(f"s_add_u32 {s('TGT')}, {s('CC')}, 1")
    b.chunk(0, "next")
    # chunk CC + 1 reads buffer 1; it stages CC + 2 -- or, at the tile's end, chunk 0 of the NEXT tile -- into buffer 0
    last = b.label("lastpair")
    join = b.label("staged")
    e(f"s_add_u32 {s('TGT')}, {s('CC')}, 2")
    e(f"s_cmp_lt_u32 {s('TGT')}, {s('NCH')}")
    e(f"s_cbranch_scc1 {join}")
    # the tile ends with this chunk: the staging state (geometry registers, FRA / FRB / WB) moves on to the next tile -- or stays on
    # this one when it is the workgroup's last (valid addresses, unused data)
    b.next_tile_coords()
    e(f"s_cmp_gt_u32 {s('NTL')}, 1")
    for n_, c_ in (("NTX", "TX"), ("NTY", "TY"), ("NCTN", "CT"), ("NFR", "FR")):
        e(f"s_cselect_b32 {s(n_)}, {s(n_)}, {s(c_)}")
    e(f"s_lshl_b32 {s('SX0')}, {s('NTX')}, 4")
    e(f"s_lshl_b32 {s('SY0')}, {s('NTY')}, 4")
    b.geometry()
    b.frame_bases("NFR")
    b.tile_weights("NCTN")
    e(f"s_mov_b32 {s('TGT')}, 0")
    e(f"{join}:")
    b.chunk(1, "next")
    e(f"s_add_u32 {s('CC')}, {s('CC')}, 2")
    e(f"s_cmp_lt_u32 {s('CC')}, {s('NCH')}")
    e("s_cbranch_scc1 S2LK_PAIR")
    b.epilogue()
    # next tile
    e(f"s_mov_b32 {s('TX')}, {s('NTX')}")
    e(f"s_mov_b32 {s('TY')}, {s('NTY')}")
    e(f"s_mov_b32 {s('CT')}, {s('NCTN')}")
    e(f"s_mov_b32 {s('FR')}, {s('NFR')}")
    e(f"s_lshl_b32 {s('X0')}, {s('TX')}, 4")
    e(f"s_lshl_b32 {s('Y0')}, {s('TY')}, 4")
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
    clob = [f"v{r}" for r in range(0, V_LAST + 1)] + [f"s{r}" for r in range(16, S_LAST + 1)] + ["vcc", "scc", "memory"]
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
