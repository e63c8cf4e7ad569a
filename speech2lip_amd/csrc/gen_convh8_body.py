"""Generates convh8_body.inc: the EIGHT-wave form of the half-width convolution (gen_convh_body.py has the design and the arithmetic, which
are unchanged: the outputs are the same bits).  Same 32 x 16-pixel tile x 64 output channels, same LDS shapes and requests; what differs
is the wave shape: 8 waves = TWO per SIMD, wave w owns rows 4 w .. 4 w + 3 as two N-blocks (accumulators 2 x 2 x 16 = 64 AGPRs, two operand
sets of 32: 128 AGPRs + <= 128 VGPRs), 8 MFMAs per tap behind 4 + 4 operand reads.  Why: with one wave per SIMD every stall of that wave --
the barrier, the first tap after it, the tile's epilogue -- idles the matrix pipe (busy 0.53 of the cycles, profiles/r04b); a second wave
on the SIMD issues its MFMAs meanwhile.  Price: a third more operand bytes from LDS (which has the headroom: 0.39 of the cycles).

Register map (per wave): a0-63 acc[mb][nb][16] | a64-95, a96-127 two operand sets (A[ks][mb] 4 x 4, B[ks][nb] 4 x 4) | VGPRs: names below."""
import os
import sys

TILE_H = 32
HALO_ROWS = TILE_H + 2
HALO_BYTES, W_BYTES = HALO_ROWS * 18 * 64, 9 * 2 * 2 * 64 * 16  # 39 168 + 36 864
N_SLOTS = HALO_BYTES // 16                                      # 2 448
BUF = HALO_BYTES + W_BYTES                                      # 76 032
BIAS_OFF = 2 * BUF                                              # 256 floats
STAT_OFF = BIAS_OFF + 1024                                       # per-wave statistics of a tile: [8 waves][64 channels][sum, sum of squares] floats
NORM_OFF = STAT_OFF + 4096                                       # (the normalising form) two tables [scale 128 | shift 128] floats, by frame parity
LDS_BYTES = NORM_OFF + 2048
STG_OFF = BUF                                                   # store staging = the start of buffer 1's halo area: 8 waves x 4 KiB
NW, NB = 8, 2                                                   # waves, N-blocks (2 rows x 16 pixels) per wave
NI = 5                                                          # halo DMA instructions per wave (wave w: slots 320 w ..)
CONST_WORDS = 23                                                # per lane, from the C++ prologue: hrc[5], bofs[6], swa[8], sra[4]
WITH_RELU = False                                               # main() generates both: convh8_body.inc (linear) and convh8r_body.inc (max(0, .))
WITH_NORM = False                                               # ... and convh8n_body.inc: the input tensor is a PRE-BatchNorm tensor z; the chunk's halo tile is
                                                                # normalised in LDS -- a = bf16(max(fma(z, scale, shift), 0)), bn_relu_h_kernel's expression -- before it is read
PFX = "S2L8"                                                    # label prefix (two bodies in one translation unit)
EXP = int(os.environ.get("S2L_CH_EXP", "0"))                   # ablation builds (results wrong): 1 no stores, 2 no halo DMA, 4 no weight DMA, 16 no gate loads, 32 no B operand reads, 64 no A operand reads, 128 no bias init, 512 no epilogue, 1024 no per-chunk barrier, 4096 / 8192 PRICING of BatchNorm's normalise + ReLU inside the consuming convolution (norm_items), 16384 the halo tile read as one contiguous 40-KiB block (access-pattern pricing)

# ---- registers
A_ACC = 0
A_OPS = (64, 96)                # operand set: A[ks][mb] at + (ks * 2 + mb) * 4, B[ks][nb] at + 16 + (ks * 2 + nb) * 4
A_LAST = 127
V_VOFF = 0                      # 5: byte offset of this lane's piece of DMA instruction i from the TILE's source pointer (a constant)
V_HRC = 5                       # 5: col | row << 8 | logical segment << 16, or negative: no such slot
V_VALID, V_ZMASK = 10, 11       # bit i: the piece is inside the image / is a halo slot outside it (zeroed); staging tile
V_SLOT = (12, 13)               # LDS address of slot (320 wave + lane) per buffer
V_DMA2 = 14                     # the wave's extra weight piece (waves 0..3)
V_Z = 16                        # 4 zeros
V_BOFS = (20, 26)
V_AOFS = (32, 33)
V_LANE, V_TID, V_HH16, V_DMA, V_BIASA, V_L3, V_L7X16, V_ONES, V_FFFF, V_INR = 34, 35, 36, 37, 38, 39, 40, 41, 42, 43
V_VS = 44                       # byte offset of this lane's 16 bytes in a store from the row's pointer (lane * 16)
V_SWA = 46                      # 8: staging write addresses per piece mb * 4 + rq
V_SRA = 54                      # 4: staging read addresses per store j
V_T = 60                        # temporaries 60..87 (the store read-backs share the accumulator temporaries)
V_G = 88                        # 32: gate pieces [nb][j][4]; a launch that leaves the tile's statistics has no gate: the same registers then hold
V_SH, V_SS, V_SQ, V_SA, V_SZ = V_G, V_G + 16, V_G + 17, V_G + 18, V_G + 22   # 16 loaded pixel pairs | sum | sum of squares | 4 staging addresses | 6 temporaries
V_LAST = 123
# backward-statistics mode (stage 1 of the NEXT layer's BatchNorm backward inside an input-gradient convolution: STATP != 0 and GATE != 0, the gate
# pointer then being that layer's z): the gate registers hold z pieces, so its persistent registers live above them and its transients in the
# accumulator temporaries the store staging is done with (V_T + 16 ..)
V_BSS, V_BSQ, V_BSC = (58, 59), (120, 121), 122                  # sum g | sum g z: (even, odd pixels) | (scale, shift) of this lane's channel: even-aligned pairs
V_BSA = (15, 45, V_T + 26, V_T + 27)                             # 4 staging addresses: holes of the map above, and two of the gating temporaries (tm: a
                                                                 # backward-statistics launch never runs the gating code, nothing else touches them inside store_tile)
ZSTG_DELTA = HALO_BYTES         # the z staging = the weight area of buffer 1 (spent at the tile's end: such launches reload their weights every tile)
# the normalising form has no gate: its work registers live where the gate pieces would (the statistics block, which runs at the tile's end,
# uses the same ones), and five more hold each piece's constant offset into a chunk's table row
V_ND, V_NC, V_NU, V_NA, V_NT = V_G, V_G + 4, V_G + 20, V_G + 28, V_G + 29
V_NSEG = 120                    # 5: (logical segment of this lane's slot in piece i) * 32 bytes
V_LAST_NORM = 124


def _scalars(first, singles, pairs, skip=(32, 33)):
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


S = _scalars(16, singles=("CA CB COUT H W TILESX TILESY NCT NCH CHA CHB WAVE LDS0 RELU HW64 HW LDSW LDSH T0 T1 T2 T3 "
                          "TX TY CT FR X0 Y0 CC NTL LEFT NTX NTY NCT_ NFR NC NLEFT SX0 SY0 RS GY0 LDSW2").split(),
             pairs=("KARG", "INA", "INB", "W16", "WB", "BIAS", "OUT", "GATE", "SRC", "WCH", "FRA", "FRB", "OUTF", "GATEF",
                    "EX", "TA", "TB", "XM0", "ROWB"))
for _n, _r in (("STATP", 98), ("STATP1", 99), ("NCOLS", 100), ("WLD", 101)):      # the statistics' output pointer, the tile's valid columns, weight loads left
    assert _r not in S.values()
    S[_n] = _r
S_LAST = max(S.values())
assert S_LAST <= 101, S_LAST
# the normalising form has no gate input: the gate's scalars carry its table pointer, the staged chunk's table row and the table's frame
S_NORM = {"NORMP": S["GATE"], "NORMP1": S["GATE1"], "NB": S["GATEF"], "NPREV": S["GATEF1"]}

# byte offsets of the fields of struct ConvHArgs (csrc/convh.hip static_asserts them)
ARG = {"inA": 0, "inB": 8, "w16": 16, "bias": 24, "out": 32, "gate": 40, "CA": 48, "CB": 52, "cout": 56, "H": 60, "W": 64, "tiles_x": 68,
       "tiles_y": 72, "n_ct": 76, "relu": 80, "stat": 88, "norm": 96, "bz": 104, "bst": 112}
# (the bias pointer is needed in the prologue only: afterwards its scalars carry bst, the per-frame rows [scale C | shift C | ...] of the layer
#  whose gradient statistics a backward-statistics launch leaves)
S_BST = {"BST": S["BIAS"], "BST1": S["BIAS1"]}


def _sreg(n):
    return S_NORM[n] if n in S_NORM else S_BST[n] if n in S_BST else S[n]


def s(n):
    return f"s{_sreg(n)}"


def s2(n):
    r = _sreg(n)
    return f"s[{r}:{r + 1}]"


class Body:
    pfx = None                                                      # label prefix; None: the module's PFX (gen_convhx_body.py sets its own)

    def __init__(self):
        self.L, self.lds, self.nlabel = [], [], 0

    def e(self, t):
        self.L.append(t)

    def label(self, stem):
        self.nlabel += 1
        return f"{self.pfx or PFX}_{stem}_{self.nlabel}"

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
        b = A_ACC + (mb * NB + nb) * 16
        return f"a[{b}:{b + 15}]"

    @staticmethod
    def opa(os_, ks, mb):
        return A_OPS[os_] + (ks * 2 + mb) * 4

    @staticmethod
    def opb(os_, ks, nb):
        return A_OPS[os_] + 16 + (ks * NB + nb) * 4

    # ---- operand reads of tap t from buffer `buf` into operand set os_: 12 (text, tag) in the order the MFMAs use them
    def tap_reads(self, t, buf, os_):
        dy, dx = t // 3, t % 3
        out = []
        for ks in range(2):
            for mb in range(2):
                r = self.opa(os_, ks, mb)
                off = ((t * 2 + ks) * 2 + mb) * 1024      # (V_AOFS points at the buffer's weight area)
                if not (EXP & 64 and t > 0):
                    out.append((f"ds_read_b128 a[{r}:{r + 3}], v{V_AOFS[buf]} offset:{off}", ("R", t, ks, 0, mb)))
            for nb in range(NB):
                r = self.opb(os_, ks, nb)
                off = (2 * nb + dy) * 18 * 64
                if not (EXP & 32 and (t > 0 or nb < NB - 1 or ks < 1)):
                    out.append((f"ds_read_b128 a[{r}:{r + 3}], v{V_BOFS[buf] + dx * 2 + ks} offset:{off}", ("R", t, ks, 1, nb)))
        return out

    def tap_mfmas(self, os_, sprinkle):
        """8 MFMAs of one tap on operand set os_; sprinkle: 8 lists of items tucked behind MFMA m"""
        m = 0
        for ks in range(2):
            for mb in range(2):
                for nb in range(NB):
                    a, b = self.opa(os_, ks, mb), self.opb(os_, ks, nb)
                    self.e(f"v_mfma_f32_32x32x16_bf16 {self.acc(mb, nb)}, a[{a}:{a + 3}], a[{b}:{b + 3}], {self.acc(mb, nb)}")
                    self.emit_group(sprinkle[m])
                    m += 1

    # ---- streams
    def next_coords(self, p, ct_name):
        """the stream with prefix p moves to its next tile (channel tile fastest, then x, y, frame: the channel tiles of one position
        read the same halo planes back to back, the second time from L2) -- or stays on its last one"""
        e = self.e
        stay = self.label("stay")
        e(f"s_cmp_lt_u32 {s(p + 'LEFT')}, 2")
        e(f"s_cbranch_scc1 {stay}")
        e(f"s_sub_u32 {s(p + 'LEFT')}, {s(p + 'LEFT')}, 1")
        e(f"s_add_u32 {s(ct_name)}, {s(ct_name)}, 1")
        e(f"s_cmp_eq_u32 {s(ct_name)}, {s('NCT')}")
        e(f"s_cselect_b32 {s(ct_name)}, 0, {s(ct_name)}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s(p + 'TX')}, {s(p + 'TX')}, {s('T0')}")
        e(f"s_cmp_eq_u32 {s(p + 'TX')}, {s('TILESX')}")
        e(f"s_cselect_b32 {s(p + 'TX')}, 0, {s(p + 'TX')}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s(p + 'TY')}, {s(p + 'TY')}, {s('T0')}")
        e(f"s_cmp_eq_u32 {s(p + 'TY')}, {s('TILESY')}")
        e(f"s_cselect_b32 {s(p + 'TY')}, 0, {s(p + 'TY')}")
        e(f"s_cselect_b32 {s('T0')}, 1, 0")
        e(f"s_add_u32 {s(p + 'FR')}, {s(p + 'FR')}, {s('T0')}")
        e(f"{stay}:")

    def geometry(self):
        """which of this lane's ten pieces of the staging stream's tile at (SX0, SY0) are inside the image (V_VALID) and which are halo
        slots outside it (V_ZMASK).  A tile whose halo lies inside the image (most): every slot is valid."""
        e = self.e
        r, c, t2, t3 = V_T, V_T + 1, V_T + 3, V_T + 4
        slow, done = self.label("border"), self.label("geo")
        e(f"s_cmp_eq_u32 {s('SX0')}, 0")
        e(f"s_cbranch_scc1 {slow}")
        e(f"s_cmp_eq_u32 {s('SY0')}, 0")
        e(f"s_cbranch_scc1 {slow}")
        e(f"s_add_u32 {s('T0')}, {s('SX0')}, 16")
        e(f"s_cmp_ge_u32 {s('T0')}, {s('W')}")
        e(f"s_cbranch_scc1 {slow}")
        e(f"s_add_u32 {s('T0')}, {s('SY0')}, {TILE_H}")
        e(f"s_cmp_ge_u32 {s('T0')}, {s('H')}")
        e(f"s_cbranch_scc1 {slow}")
        e(f"v_mov_b32 v{V_VALID}, v{V_INR}")
        e(f"v_mov_b32 v{V_ZMASK}, 0")
        e(f"s_branch {done}")
        e(f"{slow}:")
        e(f"v_mov_b32 v{V_VALID}, 0")
        e(f"v_mov_b32 v{V_ZMASK}, 0")
        for i in range(NI):
            e(f"v_bfe_u32 v{r}, v{V_HRC + i}, 8, 8")
            e(f"v_and_b32 v{c}, 0xff, v{V_HRC + i}")
            e(f"v_add_u32 v{r}, {s('SY0')}, v{r}")                     # gy + 1
            e(f"v_add_u32 v{c}, {s('SX0')}, v{c}")                     # gx + 1
            e(f"v_subrev_u32 v{r}, 1, v{r}")                            # gy (wraps below 0: fails the unsigned test)
            e(f"v_subrev_u32 v{c}, 1, v{c}")
            e(f"v_cmp_gt_u32 vcc, {s('H')}, v{r}")
            e(f"v_cmp_gt_u32 {s2('TA')}, {s('W')}, v{c}")
            e(f"v_cmp_le_i32 {s2('TB')}, 0, v{V_HRC + i}")             # the slot exists
            e(f"s_and_b64 vcc, vcc, {s2('TA')}")
            e(f"s_and_b64 vcc, vcc, {s2('TB')}")
            e(f"v_mov_b32 v{t2}, {1 << i}")
            e(f"v_cndmask_b32 v{t3}, 0, v{t2}, vcc")
            e(f"v_or_b32 v{V_VALID}, v{V_VALID}, v{t3}")
            e(f"s_andn2_b64 vcc, {s2('TB')}, vcc")
            e(f"v_cndmask_b32 v{t3}, 0, v{t2}, vcc")
            e(f"v_or_b32 v{V_ZMASK}, v{V_ZMASK}, v{t3}")
        e(f"{done}:")

    def staging_tile_setup(self):
        """validity masks, the tensors' bases FRA / FRB at the TILE's origin (halo pixel (0, 0): may lie before the frame) and the weight
        base WB for the staging stream's tile (NTX, NTY, NCT_, NFR)"""
        e = self.e
        e(f"s_lshl_b32 {s('SX0')}, {s('NTX')}, 4")
        e(f"s_mul_i32 {s('SY0')}, {s('NTY')}, {TILE_H}")
        self.geometry()
        e(f"s_sub_u32 {s('T1')}, {s('SY0')}, 1")
        e(f"s_mul_i32 {s('T1')}, {s('T1')}, {s('W')}")
        e(f"s_add_u32 {s('T1')}, {s('T1')}, {s('SX0')}")
        e(f"s_sub_u32 {s('T1')}, {s('T1')}, 1")                         # (SY0 - 1) W + SX0 - 1 >= -(W + 1): the halo origin inside a plane
        for fr, src, ch in (("FRA", "INA", "CHA"), ("FRB", "INB", "CHB")):
            e(f"s_mul_i32 {s('T0')}, {s('NFR')}, {s(ch)}")              # planes of this tensor before the frame
            e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('HW')}")             # pixels (< 2^31: the launcher checks)
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T1')}")             # signed
            e(f"s_ashr_i32 {s('T3')}, {s('T0')}, 26")                   # * 64 bytes as a signed 64-bit number
            e(f"s_lshl_b32 {s('T2')}, {s('T0')}, 6")
            e(f"s_add_u32 {s(fr)}, {s(src)}, {s('T2')}")
            e(f"s_addc_u32 {s(fr + '1')}, {s(src + '1')}, {s('T3')}")
        e(f"s_mul_i32 {s('T0')}, {s('NCT_')}, {s('NCH')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {W_BYTES}")
        e(f"s_add_u32 {s('WB')}, {s('W16')}, {s('T0')}")
        e(f"s_addc_u32 {s('WB1')}, {s('W161')}, 0")

    def staging_source(self):
        """for chunk NC of the staging tile: SRC (the chunk's 32-channel plane at the tile's halo origin) and WCH (the chunk's weights)"""
        e = self.e
        e(f"s_cmp_lt_u32 {s('NC')}, {s('CHA')}")                       # chunk from A?
        e(f"s_cselect_b64 {s2('SRC')}, {s2('FRA')}, {s2('FRB')}")
        e(f"s_cselect_b32 {s('T0')}, 0, {s('CHA')}")
        e(f"s_sub_u32 {s('T0')}, {s('NC')}, {s('T0')}")                # plane index inside its tensor
        e(f"s_mul_hi_u32 {s('T1')}, {s('T0')}, {s('HW64')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('HW64')}")
        e(f"s_add_u32 {s('SRC')}, {s('SRC')}, {s('T0')}")
        e(f"s_addc_u32 {s('SRC1')}, {s('SRC1')}, {s('T1')}")
        e(f"s_mul_i32 {s('T0')}, {s('NC')}, {W_BYTES}")
        e(f"s_add_u32 {s('WCH')}, {s('WB')}, {s('T0')}")
        e(f"s_addc_u32 {s('WCH1')}, {s('WB1')}, 0")

    def advance_staging(self):
        e = self.e
        same = self.label("nsame")
        e(f"s_add_u32 {s('NC')}, {s('NC')}, 1")
        e(f"s_cmp_lt_u32 {s('NC')}, {s('NCH')}")
        e(f"s_cbranch_scc1 {same}")
        e(f"s_mov_b32 {s('NC')}, 0")
        self.next_coords("N", "NCT_")
        self.staging_tile_setup()
        if WITH_NORM:
            keep = self.label("nsamefr")
            e(f"s_cmp_eq_u32 {s('NFR')}, {s('NPREV')}")
            e(f"s_cbranch_scc1 {keep}")
            self.norm_table_load()
            e(f"{keep}:")
        e(f"{same}:")

    # ---- instruction groups
    def dma_items(self, wbuf):
        """this wave's 1-KiB pieces of the staging chunk's 36 KiB of weights -> weight area of buffer wbuf: pieces 4 w .. 4 w + 3 and, for
        waves 0..3, piece 32 + w.  WLD counts the loads still to do: a layer of TWO chunks and ONE channel tile (64 -> 64 channels) finds
        chunk 0's weights in buffer 0 and chunk 1's in buffer 1 on every tile after the first -- the store staging aliases a HALO area -- so
        its requests stop after the first two (a fifth of those launches' LDS-DMA bytes per tile); every other layer keeps loading."""
        items = []
        for j in range(4):
            g = []
            if j == 0:
                g += [f"s_add_u32 m0, {s('LDSW')}, {wbuf * BUF}", "s_nop 0"]
            if not EXP & 4:
                lab = self.label("wres")
                g += [f"s_cmp_eq_u32 {s('WLD')}, 0", f"s_cbranch_scc1 {lab}",
                      f"global_load_lds_dwordx4 v{V_DMA}, {s2('WCH')} offset:{1024 * j}", f"{lab}:"]
            items.append(g)
        skip = self.label("nox")
        g = [f"s_cmp_gt_u32 {s('WAVE')}, 3", f"s_cbranch_scc1 {skip}", f"s_add_u32 m0, {s('LDSW2')}, {wbuf * BUF}", "s_nop 0"]
        if not EXP & 4:
            g += [f"s_cmp_eq_u32 {s('WLD')}, 0", f"s_cbranch_scc1 {skip}", f"global_load_lds_dwordx4 v{V_DMA2}, {s2('WCH')}"]
        g += [f"{skip}:", f"s_cmp_eq_u32 {s('WLD')}, 0", f"s_cselect_b32 {s('T0')}, 0, 1", f"s_sub_u32 {s('WLD')}, {s('WLD')}, {s('T0')}"]
        items.append(g)
        return items

    def halo_items(self, buf):
        """five LDS-DMA instructions of the staging chunk's halo tile -> halo area of buffer buf, and the zeroes of its slots outside the image"""
        items = []
        t = V_T + 8
        for i in range(NI):
            g = [f"s_add_u32 m0, {s('LDSH')}, {buf * BUF + 1024 * i}",
                 f"v_bfe_u32 v{t}, v{V_VALID}, {i}, 1", f"v_cmp_eq_u32 vcc, 1, v{t}", "s_nop 0", "s_mov_b64 exec, vcc"]
            if not EXP & 2:
                g += [f"global_load_lds_dwordx4 v{V_VOFF + i}, {s2('SRC')}"]
            g += ["s_mov_b64 exec, -1", f"v_bfe_u32 v{t}, v{V_ZMASK}, {i}, 1", f"v_cmp_eq_u32 vcc, 1, v{t}", "s_nop 0", "s_mov_b64 exec, vcc",
                  (f"ds_write_b128 v{V_SLOT[buf]}, v[{V_Z}:{V_Z + 3}] offset:{1024 * i}", ("Z", i)), "s_mov_b64 exec, -1"]
            items.append(g)
        return items

    def row_exec(self, row, mb, base, first):
        """exec = the lanes of the store of (row, M-block) inside the image; ROWB = base + row * RS + mb * HW64"""
        e = self.e
        if first:
            e(f"s_mov_b64 {s2('ROWB')}, {s2(base)}")
        elif mb == 0:
            e(f"s_sub_u32 {s('ROWB')}, {s('ROWB')}, {s('HW64')}")        # back to M-block 0, one row down
            e(f"s_subb_u32 {s('ROWB1')}, {s('ROWB1')}, 0")
            e(f"s_add_u32 {s('ROWB')}, {s('ROWB')}, {s('RS')}")
            e(f"s_addc_u32 {s('ROWB1')}, {s('ROWB1')}, 0")
        else:
            e(f"s_add_u32 {s('ROWB')}, {s('ROWB')}, {s('HW64')}")
            e(f"s_addc_u32 {s('ROWB1')}, {s('ROWB1')}, 0")
        e(f"s_add_u32 {s('T2')}, {s('GY0')}, {row}")
        e(f"s_cmp_lt_u32 {s('T2')}, {s('H')}")
        e(f"s_cselect_b64 exec, {s2('XM0')}, 0")

    def gate_loads(self):
        """the gate's eight pieces of this lane (the compute tile's store positions), if the launch has a gate and this is the tile's
        last pair of chunks"""
        e = self.e
        skip = self.label("nogate")
        e(f"s_add_u32 {s('T0')}, {s('CC')}, 2")
        e(f"s_cmp_lt_u32 {s('T0')}, {s('NCH')}")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_cmp_eq_u64 {s2('GATE')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        noz = self.label("nozero")
        e(f"s_cmp_eq_u64 {s2('STATP')}, 0")                             # backward statistics multiply by z: the pieces outside the image (not loaded)
        e(f"s_cbranch_scc1 {noz}")                                      # must not hold stale bits (0 x Inf)
        for k in range(16 * NB):
            e(f"v_mov_b32 v{V_G + k}, 0")
        e(f"{noz}:")
        for k in range(4 * NB):
            nb, j = k >> 2, k & 3
            self.row_exec(2 * nb + (j >> 1), j & 1, "GATEF", k == 0)
            if not EXP & 16:
                e(f"global_load_dwordx4 v[{V_G + 4 * k}:{V_G + 4 * k + 3}], v{V_VS}, {s2('ROWB')}")
        e("s_mov_b64 exec, -1")
        # backward statistics: the gate is z; lane c also fetches scale / shift of channel 64 CT + c of frame FR (the ReLU mask is fma(z, scale, shift) > 0)
        e(f"s_cmp_eq_u64 {s2('STATP')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_lshl_b32 {s('T0')}, {s('FR')}, 11")                       # 512 floats per frame
        e(f"s_add_u32 {s('TA')}, {s('BST')}, {s('T0')}")
        e(f"s_addc_u32 {s('TA1')}, {s('BST1')}, 0")
        e(f"s_lshl_b32 {s('T0')}, {s('CT')}, 8")
        e(f"s_add_u32 {s('TA')}, {s('TA')}, {s('T0')}")
        e(f"s_addc_u32 {s('TA1')}, {s('TA1')}, 0")
        e(f"s_lshl_b32 {s('T0')}, {s('COUT')}, 2")
        e(f"s_add_u32 {s('TB')}, {s('TA')}, {s('T0')}")
        e(f"s_addc_u32 {s('TB1')}, {s('TA1')}, 0")
        e(f"v_lshlrev_b32 v{V_BSC}, 2, v{V_LANE}")
        e(f"global_load_dword v{V_BSC + 1}, v{V_BSC}, {s2('TB')}")
        e(f"global_load_dword v{V_BSC}, v{V_BSC}, {s2('TA')}")          # (overwrites its own address last)
        e(f"{skip}:")

    def store_tile(self):
        """the stores of the tile that just ended (bases OUTF / GATEF, mask XM0, GY0: set by tile_begin).  An N-block's 64 channels x
        32 pixels leave through this wave's 4 KiB of LDS staging as bf16 ([M-block][pixel][32 channels]): lane (n, hh) writes the 8
        bytes of channels 8 rq + 4 hh .. + 3 of pixel n of M-block mb; lane l reads piece l & 3 of pixel 16 r + (l >> 2): a store
        instruction writes one row of the tile in one 32-channel plane, 1 KiB contiguous.  Gate (<= 0 -> 0) on the packed halves."""
        e = self.e
        ta, tc, tb, tm = V_T, V_T + 16, V_T, V_T + 24      # (the read-backs land in the accumulator temporaries, which are spent by then)
        for nb in range(NB):
            for mb in range(2):
                a0 = A_ACC + (mb * NB + nb) * 16
                for rq in range(4):
                    for j in range(4):
                        e(f"v_accvgpr_read_b32 v{ta + 4 * rq + j}, a{a0 + 4 * rq + j}")
                    if WITH_RELU:
                        for j in range(4):
                            e(f"v_max_f32 v{ta + 4 * rq + j}, {s('RELU')}, v{ta + 4 * rq + j}")
                    e(f"v_cvt_pk_bf16_f32 v{tc + 2 * rq}, v{ta + 4 * rq}, v{ta + 4 * rq + 1}")
                    e(f"v_cvt_pk_bf16_f32 v{tc + 2 * rq + 1}, v{ta + 4 * rq + 2}, v{ta + 4 * rq + 3}")
                    self.lds_op(f"ds_write_b64 v{V_SWA + mb * 4 + rq}, v[{tc + 2 * rq}:{tc + 2 * rq + 1}]", ("SW", nb, mb, rq))
            for j in range(4):
                self.lds_op(f"ds_read_b128 v[{tb + 4 * j}:{tb + 4 * j + 3}], v{V_SRA + j}", ("SR", nb, j))
            if not WITH_NORM:
                nogate = self.label("ng")
                e(f"s_cmp_eq_u64 {s2('GATE')}, 0")
                e(f"s_cbranch_scc1 {nogate}")
                e(f"s_cmp_lg_u64 {s2('STATP')}, 0")                   # backward statistics: the "gate" is z, nothing is masked here
                e(f"s_cbranch_scc1 {nogate}")
                self.wait_lds(("SR", nb, 3))
                for j in range(4):
                    for d in range(4):
                        g, m, x = V_G + 16 * nb + 4 * j + d, tm + d, tb + 4 * j + d
                        e(f"v_pk_max_i16 v{m}, v{g}, v{V_Z}")
                        e(f"v_pk_min_u16 v{m}, v{m}, v{V_ONES}")
                        e(f"v_pk_mul_lo_u16 v{m}, v{m}, v{V_FFFF}")
                        e(f"v_and_b32 v{x}, v{x}, v{m}")
                e(f"{nogate}:")
            self.wait_all_lds()                               # (either path: the read-backs have arrived)
            self.stats_block(nb)
            for j in range(4):
                self.row_exec(2 * nb + (j >> 1), j & 1, "OUTF", nb == 0 and j == 0)
                if not EXP & 1:
                    e(f"global_store_dwordx4 v{V_VS}, v[{tb + 4 * j}:{tb + 4 * j + 3}], {s2('ROWB')}")
            e("s_mov_b64 exec, -1")

    # ---- per-channel statistics of the tile (the train-mode chain: BatchNorm's batch statistics without a pass over the stored tensor)
    def stats_block(self, nb):
        """STATP != 0: accumulate, per channel, the sum and the sum of squares of THIS N-block's stored (bf16-rounded) values over its
        pixels inside the image.  The values are the ones the store staging holds ([M-block][pixel 32][32 channels] bf16, swizzled):
        lane c reads channel c (M-block c >> 5) of every pixel -- two bytes each -- and adds them in pixel order with a 0 / 1 weight per
        pixel (rows >= H, columns >= W).  After the second N-block the
        pair (sum, sum of squares) of the wave's 4 rows x 16 columns goes to LDS (STAT_OFF); tile_end adds the eight waves' pairs."""
        e = self.e
        skip = self.label("nostat")
        e(f"s_cmp_eq_u64 {s2('STATP')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        if not WITH_NORM:
            bwd = self.label("bstat")
            e(f"s_cmp_lg_u64 {s2('GATE')}, 0")
            e(f"s_cbranch_scc1 {bwd}")
        t0, t1, t2, t3, t4, t5 = (V_SZ + k for k in range(6))
        if nb == 0:
            e(f"v_mov_b32 v{V_SS}, 0")
            e(f"v_mov_b32 v{V_SQ}, 0")
            e(f"v_and_b32 v{t0}, 31, v{V_LANE}")                        # channel inside its M-block
            e(f"v_lshrrev_b32 v{t1}, 3, v{t0}")                         # its 16-byte piece of a pixel
            e(f"v_and_b32 v{t2}, 7, v{t0}")
            e(f"v_lshlrev_b32 v{t2}, 1, v{t2}")                         # byte inside the piece
            e(f"v_lshrrev_b32 v{t3}, 5, v{V_LANE}")
            e(f"v_lshlrev_b32 v{t3}, 11, v{t3}")                        # M-block * 2048
            e(f"v_add_u32 v{t2}, v{t2}, v{t3}")
            e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 12")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDS0')}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {STG_OFF}")
            e(f"v_add_u32 v{t2}, {s('T0')}, v{t2}")
            for q in range(4):                                         # pixel p's pieces are swizzled by (p >> 1) & 3
                e(f"v_xor_b32 v{t3}, {q}, v{t1}")
                e(f"v_lshlrev_b32 v{t3}, 4, v{t3}")
                e(f"v_add_u32 v{V_SA + q}, v{t2}, v{t3}")
            e(f"s_sub_u32 {s('NCOLS')}, {s('W')}, {s('X0')}")            # valid columns of the tile (X0 < W)
            e(f"s_min_u32 {s('NCOLS')}, {s('NCOLS')}, 16")
        for k, dst in ((0, "T2"), (1, "T3")):                          # the N-block's two rows: weight 1.0 inside the image
            e(f"s_add_u32 {s('T1')}, {s('GY0')}, {2 * nb + k}")
            e(f"s_cmp_lt_u32 {s('T1')}, {s('H')}")
            e(f"s_cselect_b32 {s(dst)}, 1.0, 0")

        # one pixel per register (zero-extended 16-bit reads): two d16 loads into the halves of ONE register may not be in flight together --
        # the second merges with the register's value at issue time and the first half is lost (measured: only odd pixels were counted)
        ring = 14                                                      # reads in flight (lgkmcnt counts to 15)

        def read(p):
            self.lds_op(f"ds_read_u16 v{V_SH + p % ring}, v{V_SA + ((p >> 1) & 3)} offset:{64 * p}", ("ST", nb, p))
        for p in range(ring):
            read(p)
        for p in range(32):
            self.wait_lds(("ST", nb, p))
            e(f"s_cmp_gt_u32 {s('NCOLS')}, {p & 15}")
            e(f"s_cselect_b32 {s('T0')}, {s('T2' if p < 16 else 'T3')}, 0")
            e(f"v_lshlrev_b32 v{t0}, 16, v{V_SH + p % ring}")
            e(f"v_fmac_f32 v{V_SS}, {s('T0')}, v{t0}")
            e(f"v_mul_f32 v{t1}, {s('T0')}, v{t0}")
            e(f"v_fmac_f32 v{V_SQ}, v{t1}, v{t0}")
            if p + ring < 32:
                read(p + ring)
        if nb == NB - 1:
            e(f"v_lshlrev_b32 v{t0}, 3, v{V_LANE}")
            e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 9")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDS0')}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {STAT_OFF}")
            e(f"v_add_u32 v{t0}, {s('T0')}, v{t0}")
            e(f"v_mov_b32 v{t2}, v{V_SS}")
            e(f"v_mov_b32 v{t3}, v{V_SQ}")
            self.lds_op(f"ds_write_b64 v{t0}, v[{t2}:{t3}]", ("STW",))
            self.wait_all_lds()
        if not WITH_NORM:
            self.wait_all_lds()
            e(f"s_branch {skip}")
            e(f"{bwd}:")
            self.bstats_block(nb)
        e(f"{skip}:")

    def bstats_block(self, nb):
        """Backward statistics (stage 1 of BatchNorm's backward for the layer whose output gradient this input-gradient convolution produces):
        per channel, over the N-block's pixels inside the image, sum g' and sum g' z with g' = (fma(z, scale, shift) > 0 ? g : 0) -- bn_bwd_reduce_h_kernel's
        expressions on the STORED (bf16-rounded) g.  z's tile arrived in the gate registers in the stores' layout; written to a second staging area
        (buffer 1's weight area) at the read-backs' addresses it has the layout of the g staging, so lane c reads channel c of both: two pixels per
        step, each bf16 straight into the upper half of a zeroed register (ds_read_u16_d16_hi: the fp32 value), packed fp32 arithmetic on the pair.
        Tiles wholly inside the image (T1 = all ones) skip the per-pixel validity scalars.  The pair of sums leaves through the same LDS slots and
        stats_reduce as the forward statistics; bn_bwd_finalize_kernel forms invstd (sum g z - mean sum g) from the tiles' raw sums."""
        e = self.e
        work = V_G + 16 * nb                                            # this N-block's z pieces are spent once staged: ring + temporaries live there
        t0, t1, t2, t3 = work, work + 1, work + 2, work + 3
        tf = work + 12                                                 # the fma results of a pixel pair (even-aligned)
        for j in range(4):
            self.lds_op(f"ds_write_b128 v{V_SRA + j}, v[{V_G + 16 * nb + 4 * j}:{V_G + 16 * nb + 4 * j + 3}] offset:{ZSTG_DELTA}", ("ZW", nb, j))
        # T1 = bit p set: pixel p (row p >> 4, column p & 15) of the N-block is inside the image
        if nb == 0:
            e(f"s_sub_u32 {s('NCOLS')}, {s('W')}, {s('X0')}")
            e(f"s_min_u32 {s('NCOLS')}, {s('NCOLS')}, 16")
        e(f"s_lshl_b32 {s('T0')}, 1, {s('NCOLS')}")
        e(f"s_sub_u32 {s('T0')}, {s('T0')}, 1")                          # NCOLS ones
        e(f"s_mov_b32 {s('T1')}, 0")
        for k in range(2):
            e(f"s_add_u32 {s('T2')}, {s('GY0')}, {2 * nb + k}")
            e(f"s_cmp_lt_u32 {s('T2')}, {s('H')}")
            e(f"s_cselect_b32 {s('T2')}, {s('T0')}, 0")
            if k:
                e(f"s_lshl_b32 {s('T2')}, {s('T2')}, 16")
            e(f"s_or_b32 {s('T1')}, {s('T1')}, {s('T2')}")
        self.wait_all_lds()                                            # (z staged -- its registers are free now; this wave reads only its own 4 KiB)
        if nb == 0:
            for r in V_BSS + V_BSQ:
                e(f"v_mov_b32 v{r}, 0")
            e(f"v_and_b32 v{t0}, 31, v{V_LANE}")                        # (the addresses of stats_block, in this mode's registers)
            e(f"v_lshrrev_b32 v{t1}, 3, v{t0}")
            e(f"v_and_b32 v{t2}, 7, v{t0}")
            e(f"v_lshlrev_b32 v{t2}, 1, v{t2}")
            e(f"v_lshrrev_b32 v{t3}, 5, v{V_LANE}")
            e(f"v_lshlrev_b32 v{t3}, 11, v{t3}")
            e(f"v_add_u32 v{t2}, v{t2}, v{t3}")
            e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 12")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDS0')}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {STG_OFF}")
            e(f"v_add_u32 v{t2}, {s('T0')}, v{t2}")
            for q in range(4):
                e(f"v_xor_b32 v{t3}, {q}, v{t1}")
                e(f"v_lshlrev_b32 v{t3}, 4, v{t3}")
                e(f"v_add_u32 v{V_BSA[q]}, v{t2}, v{t3}")
        for r in range(12):                                            # the ring's lower halves stay zero (d16_hi writes the upper ones; a masked g is 0 or g)
            e(f"v_mov_b32 v{work + r}, 0")
        slots = 3

        def read(q):
            r = work + 4 * (q % slots)
            for h in range(2):
                pix = 2 * q + h
                a = V_BSA[(pix >> 1) & 3]
                if EXP & 65536:      # pricing (results wrong): the statistics loop without its LDS reads
                    continue
                self.lds_op(f"ds_read_u16_d16_hi v{r + h}, v{a} offset:{64 * pix}", ("BG", nb, pix))
                self.lds_op(f"ds_read_u16_d16_hi v{r + 2 + h}, v{a} offset:{64 * pix + ZSTG_DELTA}", ("BZ", nb, pix))

        def loop(check):
            for q in range(slots):
                read(q)
            for q in range(16):
                r = work + 4 * (q % slots)
                self.wait_lds(("BZ", nb, 2 * q + 1))
                if EXP & 131072:      # pricing (results wrong): ... without its arithmetic
                    if q + slots < 16:
                        read(q + slots)
                    continue
                if check:
                    e(f"s_bitcmp1_b32 {s('T1')}, {2 * q}")
                    e(f"s_cselect_b64 {s2('TA')}, -1, 0")
                    e(f"s_bitcmp1_b32 {s('T1')}, {2 * q + 1}")
                    e(f"s_cselect_b64 {s2('TB')}, -1, 0")
                e(f"v_pk_fma_f32 v[{tf}:{tf + 1}], v[{r + 2}:{r + 3}], v[{V_BSC}:{V_BSC + 1}], v[{V_BSC}:{V_BSC + 1}] op_sel:[0,0,1] op_sel_hi:[1,0,1]")
                for h, m in ((0, "TA"), (1, "TB")):
                    e(f"v_cmp_lt_f32 vcc, 0, v{tf + h}")
                    if check:
                        e(f"s_and_b64 vcc, vcc, {s2(m)}")
                    e(f"v_cndmask_b32 v{r + h}, 0, v{r + h}, vcc")
                e(f"v_pk_add_f32 v[{V_BSS[0]}:{V_BSS[1]}], v[{V_BSS[0]}:{V_BSS[1]}], v[{r}:{r + 1}]")
                e(f"v_pk_fma_f32 v[{V_BSQ[0]}:{V_BSQ[1]}], v[{r}:{r + 1}], v[{r + 2}:{r + 3}], v[{V_BSQ[0]}:{V_BSQ[1]}]")
                if q + slots < 16:
                    read(q + slots)

        slow, done = self.label("bedge"), self.label("bdone")
        e(f"s_cmp_lg_u32 {s('T1')}, -1")
        e(f"s_cbranch_scc1 {slow}")
        saved = list(self.lds)
        loop(False)
        self.wait_all_lds()
        e(f"s_branch {done}")
        e(f"{slow}:")
        self.lds = saved
        loop(True)
        self.wait_all_lds()
        e(f"{done}:")
        if nb == NB - 1:
            e(f"v_lshlrev_b32 v{t0}, 3, v{V_LANE}")
            e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 9")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDS0')}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {STAT_OFF}")
            e(f"v_add_u32 v{t0}, {s('T0')}, v{t0}")
            e(f"v_add_f32 v{V_BSS[0]}, v{V_BSS[0]}, v{V_BSS[1]}")       # even + odd pixels
            e(f"v_add_f32 v{V_BSS[1]}, v{V_BSQ[0]}, v{V_BSQ[1]}")
            self.lds_op(f"ds_write_b64 v{t0}, v[{V_BSS[0]}:{V_BSS[1]}]", ("STW",))
        self.wait_all_lds()

    def stats_reduce(self):
        """after the tile's barrier: wave w adds, for channels 8 w .. 8 w + 7 (lanes 0..7), the eight waves' pairs in wave order and stores
        them to stat[((frame * tiles + ty * tiles_x + tx) * 2 + {0, 1}) * cout + 64 ct + channel] (channel_stats_h_kernel's partial layout
        with block = tile)"""
        e = self.e
        skip = self.label("nored")
        e(f"s_cmp_eq_u64 {s2('STATP')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        t0 = V_SZ
        e("s_mov_b64 exec, 0xff")
        e(f"v_lshlrev_b32 v{t0}, 3, v{V_LANE}")
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 6")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDS0')}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {STAT_OFF}")
        e(f"v_add_u32 v{t0}, {s('T0')}, v{t0}")
        for k in range(8):
            self.lds_op(f"ds_read_b64 v[{V_SH + 2 * k}:{V_SH + 2 * k + 1}], v{t0} offset:{512 * k}", ("SRD", k))
        # the tile's place in the partial buffer (64-bit frame offset)
        e(f"s_mul_i32 {s('T0')}, {s('TILESX')}, {s('TILESY')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('COUT')}")
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 3")                        # bytes per frame: tiles * 2 * cout * 4
        e(f"s_mul_hi_u32 {s('T1')}, {s('FR')}, {s('T0')}")
        e(f"s_mul_i32 {s('T0')}, {s('FR')}, {s('T0')}")
        e(f"s_add_u32 {s('TA')}, {s('STATP')}, {s('T0')}")
        e(f"s_addc_u32 {s('TA1')}, {s('STATP1')}, {s('T1')}")
        e(f"s_mul_i32 {s('T0')}, {s('TY')}, {s('TILESX')}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('TX')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('COUT')}")
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 1")
        e(f"s_lshl_b32 {s('T1')}, {s('CT')}, 6")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T1')}")
        e(f"s_lshl_b32 {s('T1')}, {s('WAVE')}, 3")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T1')}")
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 2")
        e(f"s_add_u32 {s('TA')}, {s('TA')}, {s('T0')}")
        e(f"s_addc_u32 {s('TA1')}, {s('TA1')}, 0")
        e(f"v_lshlrev_b32 v{t0}, 2, v{V_LANE}")
        self.wait_all_lds()
        for k in range(1, 8):                                          # fixed order: wave 0 + wave 1 + ... + wave 7
            e(f"v_add_f32 v{V_SH}, v{V_SH}, v{V_SH + 2 * k}")
            e(f"v_add_f32 v{V_SH + 1}, v{V_SH + 1}, v{V_SH + 2 * k + 1}")
        e(f"global_store_dword v{t0}, v{V_SH}, {s2('TA')}")
        e(f"s_lshl_b32 {s('T0')}, {s('COUT')}, 2")
        e(f"s_add_u32 {s('TA')}, {s('TA')}, {s('T0')}")
        e(f"s_addc_u32 {s('TA1')}, {s('TA1')}, 0")
        e(f"global_store_dword v{t0}, v{V_SH + 1}, {s2('TA')}")
        e("s_mov_b64 exec, -1")
        e(f"{skip}:")

    # ---- the normalising form (WITH_NORM): BatchNorm's a = relu(z * scale + shift) of the PRODUCING layer, applied to the staged halo tile in LDS
    def norm_table_load(self):
        """The staging stream's frame NFR has its own statistics (every frame is a group): scale[0..CA) | shift[0..CA) of that frame (the
        row st[NFR * 512 ..] bn_finalize_groups_kernel wrote: scale at [c], shift at [CA + c]) -> LDS table NFR & 1 as [scale 128 | shift 128]
        floats.  EVERY wave fetches and writes the whole kilobyte itself and waits for its own writes: whoever reads the table later finds its own
        copy landed (the other waves write the same values), so no barrier is needed; the parity keeps the table of the frame that is still being
        normalised by a slower... no: by this same wave one chunk earlier -- untouched.  Rare (a workgroup crosses a frame boundary a few times
        per launch); its latency is exposed."""
        e = self.e
        t0, t1, d = V_NT, V_NA, V_ND
        e(f"s_lshl_b32 {s('T0')}, {s('NFR')}, 11")                     # 512 floats per frame
        e(f"s_add_u32 {s('TA')}, {s('NORMP')}, {s('T0')}")
        e(f"s_addc_u32 {s('TA1')}, {s('NORMP1')}, 0")
        e(f"s_lshl_b32 {s('T1')}, {s('CA')}, 2")                        # the shifts start CA floats into the row
        e(f"v_and_b32 v{t0}, 31, v{V_LANE}")
        e(f"v_lshlrev_b32 v{t0}, 4, v{t0}")                             # lanes 0..31: 16 bytes of scale each, lanes 32..63: of shift
        e(f"v_mov_b32 v{t1}, {s('T1')}")
        e(f"v_cmp_lt_u32 vcc, 31, v{V_LANE}")
        e(f"v_cndmask_b32 v{t1}, 0, v{t1}, vcc")
        e(f"v_add_u32 v{t0}, v{t0}, v{t1}")
        e(f"global_load_dwordx4 v[{d}:{d + 3}], v{t0}, {s2('TA')}")
        e(f"s_and_b32 {s('T0')}, {s('NFR')}, 1")
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 10")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDS0')}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {NORM_OFF}")
        e(f"v_add_u32 v{t0}, {s('T0')}, v{V_VS}")                       # + lane * 16: [scale 128 | shift 128]
        e("s_waitcnt vmcnt(0)")
        e(f"ds_write_b128 v{t0}, v[{d}:{d + 3}]")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_mov_b32 {s('NPREV')}, {s('NFR')}")

    def norm_row(self):
        """NB = the LDS address of the staged chunk's 32 scales (its shifts: + 512): table NFR & 1, row NC"""
        e = self.e
        e(f"s_and_b32 {s('NB')}, {s('NFR')}, 1")
        e(f"s_lshl_b32 {s('NB')}, {s('NB')}, 10")
        e(f"s_lshl_b32 {s('T0')}, {s('NC')}, 7")
        e(f"s_add_u32 {s('NB')}, {s('NB')}, {s('T0')}")
        e(f"s_add_u32 {s('NB')}, {s('NB')}, {s('LDS0')}")
        e(f"s_add_u32 {s('NB')}, {s('NB')}, {NORM_OFF}")

    def norm_real_items(self, buf):
        """This wave's own five 1-KiB pieces of the staged chunk's halo tile (buffer buf), once they have landed: per 16-byte slot (8 channels
        of one pixel) a ds_read_b128, its 8 scales + 8 shifts (4 ds_read_b128 of the table row at the slot's logical segment), 8 unpacks,
        8 v_fma_f32, 8 v_max_f32 with 0, 4 v_cvt_pk_bf16_f32 (round to nearest even) and the ds_write_b128 back -- under the lanes whose slot is
        a pixel INSIDE the image: a padding pixel stays the zero the halo requests wrote (the convolution pads the ACTIVATION with zeros, not z),
        a lane without a slot writes nothing.  25 groups for the sprinkle lists."""
        d, c, u, a, t = V_ND, V_NC, V_NU, V_NA, V_NT
        items = []
        for i in range(NI):
            slot = f"v{V_SLOT[buf]}"
            g0 = [f"v_add_u32 v{a}, {s('NB')}, v{V_NSEG + i}",
                  (f"ds_read_b128 v[{d}:{d + 3}], {slot} offset:{1024 * i}", ("N", i, 0)),
                  (f"ds_read_b128 v[{c}:{c + 3}], v{a}", ("N", i, 1)), (f"ds_read_b128 v[{c + 4}:{c + 7}], v{a} offset:16", ("N", i, 2)),
                  (f"ds_read_b128 v[{c + 8}:{c + 11}], v{a} offset:512", ("N", i, 3)), (f"ds_read_b128 v[{c + 12}:{c + 15}], v{a} offset:528", ("N", i, 4))]
            g1 = [("wait", ("N", i, 4))]
            for k in range(4):      # a dword = channels 2 k (low half) and 2 k + 1 (high half)
                g1 += [f"v_lshlrev_b32 v{u + 2 * k}, 16, v{d + k}", f"v_and_b32 v{u + 2 * k + 1}, 0xffff0000, v{d + k}"]
            g2 = [f"v_fma_f32 v{u + k}, v{u + k}, v{c + k}, v{c + 8 + k}" for k in range(8)]
            g3 = [f"v_max_f32 v{u + k}, 0, v{u + k}" for k in range(8)]
            g4 = [f"v_cvt_pk_bf16_f32 v{d + k}, v{u + 2 * k}, v{u + 2 * k + 1}" for k in range(4)]
            g4 += [f"v_bfe_u32 v{t}, v{V_VALID}, {i}, 1", f"v_cmp_eq_u32 vcc, 1, v{t}", "s_nop 0", "s_mov_b64 exec, vcc",
                   (f"ds_write_b128 {slot}, v[{d}:{d + 3}] offset:{1024 * i}", ("N", i, 9)), "s_mov_b64 exec, -1"]
            items += [g0, g1, g2, g3, g4]
        return items

    # ---- PRICING ONLY (S2L_CH_EXP & 4096 / 8192): what would a = relu(z * scale + shift) cost INSIDE the convolution that consumes it?
    def norm_items(self, buf):
        """The work of normalising this wave's own five 1-KiB pieces of the staging chunk's halo tile in place (buffer buf), once they
        have landed: per 16-byte slot one ds_read_b128, the 8 + 8 per-channel constants (4 ds_read_b128 of a table -- here: arbitrary LDS
        words), 8 unpacks, 8 fma, 8 max, 4 packs, one ds_write_b128.  Arithmetic on whatever the slots hold: the results are wrong, the
        instruction mix and its LDS traffic are what the real thing would issue.  Returns 25 groups for the sprinkle lists."""
        d, c, u = V_T, V_T + 4, V_T + 20                               # data 4 | constants 16 | unpacked 8
        items = []
        for i in range(NI):
            a = f"v{V_SLOT[buf]}"
            g0 = [(f"ds_read_b128 v[{d}:{d + 3}], {a} offset:{1024 * i}", ("N", i, 0))]
            for k in range(4):
                g0.append((f"ds_read_b128 v[{c + 4 * k}:{c + 4 * k + 3}], v{V_BIASA} offset:{16 * k}", ("N", i, 1 + k)))
            g1 = [("wait", ("N", i, 4))]
            for k in range(4):
                g1 += [f"v_lshlrev_b32 v{u + 2 * k}, 16, v{d + k}", f"v_and_b32 v{u + 2 * k + 1}, 0xffff0000, v{d + k}"]
            # (identity arithmetic of the same instruction classes: the operands the MFMAs see -- and with them the power-governed clock --
            #  stay those of the plain kernel; with real constants from an all-zero table the tile became zeros and the kernel 8 % FASTER)
            g2 = [f"v_fma_f32 v{u + k}, v{u + k}, 1.0, 0" for k in range(8)] + [f"v_max_f32 v{c + k}, v{c + k}, v{c + 8 + k}" for k in range(2)]
            g3 = [f"v_max_f32 v{u + k}, v{u + k}, v{u + k}" for k in range(8)]
            g4 = [f"v_cvt_pk_bf16_f32 v{d + k}, v{u + 2 * k}, v{u + 2 * k + 1}" for k in range(4)]
            g4.append((f"ds_write_b128 {a}, v[{d}:{d + 3}] offset:{1024 * i}", ("N", i, 9)))
            items += [g0, g1, g2, g3, g4]
        return items

    # ---- one chunk of the compute stream: reads buffer p, stages the next chunk into buffer p ^ 1
    def chunk(self, p):
        e = self.e
        self.staging_source()
        dma = self.dma_items(p ^ 1)
        hal = self.halo_items(p ^ 1)
        if EXP & 8192:      # pricing, pessimistic placement: the chunk's own buffer is normalised right behind the barrier, by all waves at once
            for g in self.norm_items(p):
                self.emit_group(g)
            self.wait_all_lds()
            e("s_barrier")
        norm = self.norm_items(p ^ 1) if EXP & 4096 else []
        if WITH_NORM:
            self.norm_row()
            norm = self.norm_real_items(p ^ 1)
        for text, tag in self.tap_reads(0, p, 0):        # operands of tap 0 (exposed after the barrier)
            self.lds_op(text, tag)
        for t in range(9):
            os_ = t & 1
            nxt = self.tap_reads(t + 1, p, os_ ^ 1) if t + 1 < 9 else []
            sprinkle = [[] for _ in range(8)]
            for m, rd in enumerate(nxt):
                sprinkle[m].append(rd)
            # m0 belongs to one group at a time: the weight pieces behind tap 0, the halo requests behind taps 1..2
            if t == 0:
                for k, g in enumerate(dma):
                    sprinkle[3 + k].extend(g)
            if t in (1, 2):
                for k, g in enumerate(hal[:3] if t == 1 else hal[3:]):
                    sprinkle[2 + 2 * k].extend(g)
            if t == 4 and p == 1 and not WITH_NORM:
                self.gate_loads()
            if norm and t == 6:      # pricing, optimistic placement: this wave's own pieces have landed (it waits for them), the work hides
                e("s_waitcnt vmcnt(0)")      # behind the MFMAs of the last three taps
            if norm and t >= 6:
                for m in range(8):
                    if norm:
                        sprinkle[m].extend(norm.pop(0))
                if t == 8:
                    while norm:
                        sprinkle[7].extend(norm.pop(0))
            if EXP & 96:
                self.wait_all_lds()
            else:
                self.wait_lds(("R", t, 1, 1, NB - 1))
            self.tap_mfmas(os_, sprinkle)
        self.wait_all_lds()
        if not EXP & 32768:      # (pricing, results racy: nobody waits for the chunk's requests -- what is left of their cost is bandwidth / power, not latency)
            e("s_waitcnt vmcnt(0)")
        if not EXP & 1024:
            e("s_barrier")
        self.advance_staging()

    def bias_init(self):
        """acc = bias of (CT, mb, channels (r & 3) + 8 (r >> 2) + 4 hh), straight into the AGPRs of the four N-blocks"""
        e = self.e
        if EXP & 128:
            return
        if EXP & 256:                                                   # zeros by register writes instead of the LDS table
            for r in range(32 * NB):
                e(f"v_accvgpr_write_b32 a{A_ACC + r}, 0")
            return
        e(f"s_lshl_b32 {s('T0')}, {s('CT')}, 8")                        # CT * 64 floats
        e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_BIASA}")
        for mb in range(2):
            for nb in range(NB):
                for rq in range(4):
                    b = A_ACC + (mb * NB + nb) * 16 + 4 * rq
                    self.lds_op(f"ds_read_b128 a[{b}:{b + 3}], v{V_T} offset:{mb * 128 + rq * 32}", ("BI", mb, nb, rq))
                if nb % 2:
                    self.wait_all_lds()

    def tile_begin(self):
        """the compute tile's store state: OUTF / GATEF = the address of (plane 2 CT of frame FR, first row of this wave, column X0),
        GY0 = that row, XM0 = the lanes whose column (X0 + (lane >> 2)) is inside the image"""
        e = self.e
        e(f"s_lshl_b32 {s('X0')}, {s('TX')}, 4")
        e(f"s_mul_i32 {s('Y0')}, {s('TY')}, {TILE_H}")
        e(f"s_lshl_b32 {s('GY0')}, {s('WAVE')}, 2")                     # 4 rows per wave
        e(f"s_add_u32 {s('GY0')}, {s('GY0')}, {s('Y0')}")
        e(f"s_lshr_b32 {s('T0')}, {s('COUT')}, 5")                      # planes per frame
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('FR')}")
        e(f"s_lshl_b32 {s('T1')}, {s('CT')}, 1")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T1')}")                 # plane index of (FR, CT, M-block 0)
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('HW')}")
        e(f"s_mul_i32 {s('T1')}, {s('GY0')}, {s('W')}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T1')}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('X0')}")                 # pixel index (< 2^31: the launcher checks)
        e(f"s_lshr_b32 {s('T3')}, {s('T0')}, 26")
        e(f"s_lshl_b32 {s('T2')}, {s('T0')}, 6")
        for dst, src in (("OUTF", "OUT"),) if WITH_NORM else (("OUTF", "OUT"), ("GATEF", "GATE")):
            e(f"s_add_u32 {s(dst)}, {s(src)}, {s('T2')}")
            e(f"s_addc_u32 {s(dst + '1')}, {s(src + '1')}, {s('T3')}")
        e(f"v_lshrrev_b32 v{V_T}, 2, v{V_LANE}")
        e(f"v_add_u32 v{V_T}, {s('X0')}, v{V_T}")
        e(f"v_cmp_gt_u32 {s2('XM0')}, {s('W')}, v{V_T}")

    def tile_end(self):
        e = self.e
        e("s_nop 7")
        e("s_nop 7")                                                  # (MFMA results -> v_accvgpr_read)
        if not EXP & 512:
            self.store_tile()
        e("s_barrier")                                                # (the staging area is buffer 1's halo region: nobody may stage the next
        self.stats_reduce()                                           #  tile's chunk 1 into it while another wave still stores)
        self.next_coords("", "CT")


def emit_prologue(b, with_gate=True):
    """everything before the tile loop: arguments, per-lane constants, the bias table, chunk 0 of the first tile staged into buffer 0
    (shared with csrc/gen_convhx_body.py, whose register map is this one's without the gate pieces)"""
    e = b.e
    PFX_ = b.pfx or PFX
    e(f"s_mov_b64 {s2('KARG')}, %[karg]")
    for dst, src in (("WAVE", "wave"), ("LDS0", "lds0"), ("TX", "tx0"), ("TY", "ty0"), ("CT", "ct0"), ("FR", "fr0"), ("NTL", "ntl")):
        e(f"s_mov_b32 {s(dst)}, %[{src}]")
    for dst, field in (("INA", "inA"), ("INB", "inB"), ("W16", "w16"), ("BIAS", "bias"), ("OUT", "out"),
                       ("NORMP", "norm") if WITH_NORM else ("GATE", "gate"), ("STATP", "stat")):
        e(f"s_load_dwordx2 {s2(dst)}, {s2('KARG')}, {ARG[field]}")
    for dst, field in (("CA", "CA"), ("CB", "CB"), ("COUT", "cout"), ("H", "H"), ("W", "W"), ("TILESX", "tiles_x"), ("TILESY", "tiles_y"),
                       ("NCT", "n_ct"), ("T3", "relu")):
        e(f"s_load_dword {s(dst)}, {s2('KARG')}, {ARG[field]}")
    e(f"v_mov_b32 v{V_TID}, %[tid]")
    e(f"v_and_b32 v{V_LANE}, 63, v{V_TID}")
    # per-lane constants: the C++ prologue left CONST_WORDS words per lane at the start of LDS ([word][512 threads])
    e(f"v_lshlrev_b32 v{V_T}, 2, v{V_TID}")
    e(f"v_add_u32 v{V_T}, {s('LDS0')}, v{V_T}")
    for i in range(NI):
        e(f"ds_read_b32 v{V_HRC + i}, v{V_T} offset:{2048 * i}")
    for i in range(6):
        e(f"ds_read_b32 v{V_BOFS[0] + i}, v{V_T} offset:{2048 * (NI + i)}")
    e("s_waitcnt lgkmcnt(0)")
    for i in range(8):
        e(f"ds_read_b32 v{V_SWA + i}, v{V_T} offset:{2048 * (NI + 6 + i)}")
    for i in range(4):
        e(f"ds_read_b32 v{V_SRA + i}, v{V_T} offset:{2048 * (NI + 14 + i)}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                                    # (everybody has read its constants: the buffers may be written)
    for i in range(6):
        e(f"v_add_u32 v{V_BOFS[1] + i}, {BUF}, v{V_BOFS[0] + i}")
    e(f"v_lshlrev_b32 v{V_T}, 4, v{V_LANE}")                           # lane * 16
    e(f"v_lshrrev_b32 v{V_HH16}, 5, v{V_LANE}")
    e(f"v_lshlrev_b32 v{V_HH16}, 4, v{V_HH16}")                        # hh * 16 bytes
    e(f"v_lshrrev_b32 v{V_L3}, 3, v{V_LANE}")
    e(f"v_and_b32 v{V_L7X16}, 7, v{V_LANE}")
    e(f"v_lshlrev_b32 v{V_L7X16}, 4, v{V_L7X16}")
    e(f"v_mov_b32 v{V_ONES}, 0x00010001")
    e(f"v_mov_b32 v{V_FFFF}, -1")
    for k in range(4):
        e(f"v_mov_b32 v{V_Z + k}, 0")
    for k in range(16 * NB if with_gate else 0):
        e(f"v_mov_b32 v{V_G + k}, 0x3f803f80")                          # no gate: every half passes
    e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 12")
    e(f"v_add_u32 v{V_DMA}, {s('T0')}, v{V_T}")                         # this lane's 16 B of the wave's four 1-KiB weight pieces 4 w .. (global offset)
    e(f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 10")
    e(f"s_add_u32 {s('T2')}, {s('T2')}, 32768")
    e(f"v_add_u32 v{V_DMA2}, {s('T2')}, v{V_T}")                        # ... and of piece 32 + w (waves 0..3)
    e(f"s_add_u32 {s('LDSW2')}, {s('LDS0')}, {HALO_BYTES}")
    e(f"s_add_u32 {s('LDSW2')}, {s('LDSW2')}, {s('T2')}")
    e(f"s_add_u32 {s('T1')}, {s('LDS0')}, {HALO_BYTES}")
    e(f"v_add_u32 v{V_AOFS[0]}, {s('T1')}, v{V_T}")                      # A reads: the buffer's weight area + lane * 16
    e(f"v_add_u32 v{V_AOFS[1]}, {BUF}, v{V_AOFS[0]}")
    e(f"s_add_u32 {s('LDSW')}, {s('LDS0')}, {HALO_BYTES}")
    e(f"s_add_u32 {s('LDSW')}, {s('LDSW')}, {s('T0')}")                 # LDS address of this wave's weight pieces in buffer 0
    e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, {NI * 1024}")
    e(f"s_add_u32 {s('LDSH')}, {s('LDS0')}, {s('T0')}")                 # LDS address of this wave's halo slots in buffer 0
    e(f"v_add_u32 v{V_SLOT[0]}, {s('LDSH')}, v{V_T}")
    e(f"v_add_u32 v{V_SLOT[1]}, {BUF}, v{V_SLOT[0]}")
    e(f"s_add_u32 {s('T0')}, {s('LDS0')}, {BIAS_OFF}")
    e(f"v_add_u32 v{V_BIASA}, {s('T0')}, v{V_HH16}")                    # bias table + hh * 16
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_add_u32 {s('NCH')}, {s('CA')}, {s('CB')}")
    e(f"s_lshr_b32 {s('NCH')}, {s('NCH')}, 5")
    e(f"s_lshr_b32 {s('CHA')}, {s('CA')}, 5")
    e(f"s_lshr_b32 {s('CHB')}, {s('CB')}, 5")
    # the pieces' constant byte offsets from the tile's origin inside a 32-channel plane, and which of the ten exist
    e(f"v_mov_b32 v{V_INR}, 0")
    for i in range(NI):
        r, c, t = V_T, V_T + 1, V_T + 2
        e(f"v_bfe_u32 v{r}, v{V_HRC + i}, 8, 8")
        e(f"v_and_b32 v{c}, 0xff, v{V_HRC + i}")
        e(f"v_mul_lo_u32 v{r}, v{r}, {s('W')}")
        e(f"v_add_u32 v{r}, v{r}, v{c}")
        e(f"v_lshlrev_b32 v{r}, 6, v{r}")
        e(f"v_bfe_u32 v{c}, v{V_HRC + i}, 16, 2")
        e(f"v_lshlrev_b32 v{c}, 4, v{c}")
        e(f"v_add_u32 v{r}, v{r}, v{c}")
        e(f"v_cmp_le_i32 vcc, 0, v{V_HRC + i}")
        e(f"v_cndmask_b32 v{V_VOFF + i}, 0, v{r}, vcc")
        e(f"v_mov_b32 v{t}, {1 << i}")
        e(f"v_cndmask_b32 v{t}, 0, v{t}, vcc")
        e(f"v_or_b32 v{V_INR}, v{V_INR}, v{t}")
    if EXP & 16384:      # pricing (results wrong): the halo tile's 40 KiB as ONE contiguous block at the tile's origin instead of 34 strided rows
        for i in range(NI):
            e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, {NI * 1024}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {1024 * i}")
            e(f"s_add_u32 {s('T1')}, {s('W')}, 1")                      # (the tile's halo origin may lie (W + 1) pixels before the tensor)
            e(f"s_lshl_b32 {s('T1')}, {s('T1')}, 6")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T1')}")
            e(f"v_lshlrev_b32 v{V_VOFF + i}, 4, v{V_LANE}")
            e(f"v_add_u32 v{V_VOFF + i}, {s('T0')}, v{V_VOFF + i}")
    if WITH_NORM:
        for i in range(NI):                                            # the slot's logical segment (channels 8 seg .. + 7 of the chunk) * 32 bytes
            e(f"v_bfe_u32 v{V_NSEG + i}, v{V_HRC + i}, 16, 2")
            e(f"v_lshlrev_b32 v{V_NSEG + i}, 5, v{V_NSEG + i}")
    e(f"s_mul_i32 {s('HW')}, {s('H')}, {s('W')}")
    e(f"s_lshl_b32 {s('HW64')}, {s('HW')}, 6")                          # bytes of one 32-channel plane (< 2^31: the launcher checks)
    e(f"s_lshl_b32 {s('RS')}, {s('W')}, 6")                             # bytes between rows of a plane
    e(f"v_lshlrev_b32 v{V_VS}, 4, v{V_LANE}")                           # a store's lane offset: column lane >> 2, piece lane & 3
    e(f"s_cmp_eq_u32 {s('T3')}, 0")
    e(f"s_cselect_b32 {s('RELU')}, 0xff800000, 0")                    # lower bound of the epilogue's v_max: -inf (linear) or 0
    # the bias table -> LDS: thread t (< 64 * NCT) copies bias[t] (no bias: zeros)
    e(f"s_lshl_b32 {s('T0')}, {s('NCT')}, 6")
    e(f"v_cmp_gt_u32 vcc, {s('T0')}, v{V_TID}")
    e(f"v_lshlrev_b32 v{V_T}, 2, v{V_TID}")
    e(f"v_mov_b32 v{V_T + 1}, 0")
    e(f"s_mov_b64 {s2('EX')}, exec")
    e(f"s_cmp_eq_u64 {s2('BIAS')}, 0")
    e(f"s_cbranch_scc1 {PFX_}_NOBIAS")
    e("s_and_b64 exec, exec, vcc")
    e(f"global_load_dword v{V_T + 1}, v{V_T}, {s2('BIAS')}")
    e(f"s_mov_b64 exec, {s2('EX')}")
    e("s_waitcnt vmcnt(0)")
    e(f"{PFX_}_NOBIAS:")
    e(f"s_add_u32 {s('T0')}, {s('LDS0')}, {BIAS_OFF}")
    e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_T}")
    e(f"ds_write_b32 v{V_T}, v{V_T + 1}")
    # the staging stream starts on the workgroup's first tile
    e(f"s_mov_b32 {s('LEFT')}, {s('NTL')}")
    e(f"s_mov_b32 {s('NLEFT')}, {s('NTL')}")
    for n_, c_ in (("NTX", "TX"), ("NTY", "TY"), ("NCT_", "CT"), ("NFR", "FR")):
        e(f"s_mov_b32 {s(n_)}, {s(c_)}")
    e(f"s_mov_b32 {s('NC')}, 0")
    # weight loads left: 2 for a layer of two chunks and one channel tile (its weights then stay in the two buffers), else unbounded
    e(f"s_cmp_eq_u32 {s('NCH')}, 2")
    e(f"s_cselect_b32 {s('T0')}, 1, 0")
    e(f"s_cmp_eq_u32 {s('NCT')}, 1")
    e(f"s_cselect_b32 {s('T1')}, 1, 0")
    e(f"s_and_b32 {s('T0')}, {s('T0')}, {s('T1')}")
    e(f"s_cmp_eq_u32 {s('T0')}, 1")
    e(f"s_cselect_b32 {s('WLD')}, 2, 0x7fffffff")
    if with_gate:      # backward statistics stage z in buffer 1's weight area at every tile's end: nothing persists there
        e(f"s_cmp_eq_u64 {s2('STATP')}, 0")
        e(f"s_cselect_b32 {s('T0')}, {s('WLD')}, 0x7fffffff")
        e(f"s_cmp_eq_u64 {s2('GATE')}, 0")
        e(f"s_cselect_b32 {s('WLD')}, {s('WLD')}, {s('T0')}")
        e(f"s_load_dwordx2 {s2('BST')}, {s2('KARG')}, {ARG['bst']}")      # (the bias table is in LDS: its pointer's scalars are free)
        e("s_waitcnt lgkmcnt(0)")
    b.staging_tile_setup()
    # chunk 0 -> buffer 0, all exposed (once per workgroup)
    b.staging_source()
    for g in b.dma_items(0):
        b.emit_group(g)
    for g in b.halo_items(0):
        b.emit_group(g)
    if WITH_NORM:      # the first frame's table, then chunk 0's own normalisation (exposed, once per workgroup)
        b.wait_all_lds()
        b.norm_table_load()
        b.norm_row()
        e("s_waitcnt vmcnt(0)")
        for g in b.norm_real_items(0):
            b.emit_group(g)
    b.advance_staging()
    b.wait_all_lds()
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")


def generate():
    b = Body()
    e = b.e
    emit_prologue(b, with_gate=not WITH_NORM)
    # ================= tile loop
    e(f"{PFX}_TILE:")
    b.tile_begin()
    b.bias_init()
    e(f"s_mov_b32 {s('CC')}, 0")
    e(f"{PFX}_PAIR:")
    b.chunk(0)
    b.chunk(1)
    e(f"s_add_u32 {s('CC')}, {s('CC')}, 2")
    e(f"s_cmp_lt_u32 {s('CC')}, {s('NCH')}")
    e(f"s_cbranch_scc1 {PFX}_PAIR")
    b.tile_end()
    e(f"s_sub_u32 {s('NTL')}, {s('NTL')}, 1")
    e(f"s_cmp_gt_u32 {s('NTL')}, 0")
    e(f"s_cbranch_scc1 {PFX}_TILE")
    e("s_waitcnt vmcnt(0)")
    return b.L


OPERANDS = """      :
      : [karg] "s"(karg), [wave] "s"(wave), [lds0] "s"(lds0), [tx0] "s"(tx0), [ty0] "s"(ty0), [ct0] "s"(ct0), [fr0] "s"(fr0), [ntl] "s"(ntl),
        [tid] "v"(tid)
"""


def main(objdir):
    global WITH_RELU, WITH_NORM, PFX
    n = 0
    for relu, norm, pfx, name in ((False, False, "S2L8", "convh8_body.inc"), (True, False, "S2L9", "convh8r_body.inc"), (False, True, "S2LN", "convh8n_body.inc")):
        WITH_RELU, WITH_NORM, PFX = relu, norm, pfx
        lines = generate()
        clob = [f"v{r}" for r in range(0, (V_LAST_NORM if norm else V_LAST) + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)] + [f"s{r}" for r in range(16, S_LAST + 1) if r not in (32, 33)]
        clob += ["vcc", "scc", "memory"]
        out = ["// GENERATED by csrc/gen_convh8_body.py -- do not edit; the generator is the source.", "asm volatile("]
        out += [f'    "{x}\\n\\t"' for x in lines]
        out.append(OPERANDS.rstrip("\n"))
        out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
        with open(os.path.join(objdir, name), "w") as f:
            f.write("\n".join(out) + "\n")
        n = n or len(lines)
    WITH_RELU, WITH_NORM, PFX = False, False, "S2L8"
    return n


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"convh8 body: {main(d)} instructions")
