"""Generates conv_body_<variant>.inc: the body of s2l::conv3x3_asm_kernel<variant> (csrc/unet.hip) as ONE fixed-register gfx950
assembly text, like gen_render_body.py does for the renderer and for the same reason: with one wave per SIMD a wave's VALU
instructions never overlap its own MFMAs, so the chunk loop must be nothing but MFMAs, LDS reads, memory-instruction issues and
scalar code.

The convolution is the implicit GEMM of conv3x3_kernel (same tile, same packed weights, same LDS layouts, same accumulation
order per output: chunk, tap, k-step -- results are bit-identical):
  * tile = 16x16 pixels x 64 output channels; wave w owns tile rows 4w..4w+3 (g = 0..3), lane (q = lane >> 4, px = lane & 15);
    D[channel 16 mb + 4 q + r][pixel px of row g] in acc[mb][g][r] (64 VGPRs, C/D in VGPRs);
  * input channels go through LDS 16 at a time: halo tile [18*18][16] fp32 (20 736 B) + the chunk's weights in A-operand order
    [9 taps][4 mb][64 lanes][4] (36 864 B); TWO such buffers, filled by LDS-DMA (global_load_lds_dwordx4: 6 + 9 instructions per
    wave and chunk), one instruction behind every 12th MFMA of the chunk before: the texture path takes 64 cycles per 1-KiB
    instruction of any wave of the CU, and a burst stalls the issuing wave -- and with it its MFMAs;
  * one s_barrier per chunk, placed BEFORE the last tap's MFMAs: behind it the first tap's operands of the next chunk are read
    (into the operand set the last tap does not use) while those 64 MFMAs run, so a chunk boundary exposes no LDS latency;
  * out-of-image halo pixels: the DMA lanes are masked off (exec) and the buffer's halo region is zero-filled by ds_write first
    (border tiles only; interior tiles take a scalar-only set-up: the lanes' halo offsets are launch constants, the tile's
    origin lives in the scalar base address);
  * a workgroup is persistent (one per CU) and walks a contiguous range of tiles (frame, channel tile, y, x; x fastest); the
    first chunk of the next tile is fetched during the last chunk of the current one;
  * end of a tile: ReLU writes the accumulators into 64 staging registers; the 16 (+ 8 pooled) stores are issued behind MFMAs of
    the NEXT tile's first chunk (16 bytes per lane and cycle is all the store path takes: 4 k cycles per tile if exposed).

Variants: "fwd" (bias, ReLU), "fwd_pool" (+ the 2x2 max-pooled copy), "fwd_out" (+ the network's 1x1 output convolution
64 -> 3 on the ReLU'd tile, conv3x3_kernel<true>'s arithmetic in its order; the 64-channel activation itself is stored only
when the caller keeps it for a backward pass), "lin" (no bias, no ReLU; an optional gate tensor of the output's shape zeroes the
result where it is <= 0: the input-gradient convolutions of the backward passes -- fed with the transposed, tap-mirrored chunks --
and the raw convolutions of the train-mode forward).  Inputs A | B (virtual concat) with CA % 16 == 0, CB in
(0, CA) and an even number of chunks, so that a tile starts on buffer 0 and ends on buffer 1."""
import os
import sys

EVERY = int(os.environ.get("S2L_CONV_EVERY", "12"))   # one group of memory instructions behind every EVERY-th MFMA of a chunk
EVERY_FIRST = int(os.environ.get("S2L_CONV_EVERY_FIRST", "10"))   # ... of a tile's first chunk (DMA and the stores of the tile before)
TRACE = os.environ.get("S2L_CONV_TRACE") == "1"      # experiment builds only (tools/trace_conv_asm.py): per-tile timestamps
HALO_BYTES, W_BYTES = 18 * 18 * 64, 9 * 4 * 64 * 16
BUF = HALO_BYTES + W_BYTES           # 57 600

# ---- vector registers
V_ACC = 0                      # 64: acc[(mb*4+g)*4 + r]
V_OPS = (64, 96)               # two operand sets: A quads of mb 0..3 (16 regs), then B quads of g 0..3 (16 regs)
V_LANE16, V_PART16, V_PX, V_Q16 = 128, 129, 130, 131
V_ABASE = (132, 133)           # A-read base of buffer 0 / 1
V_BBASE = (134, 135)           # B-read base of buffer 0 / 1
V_HOFF = 136                   # 6: halo voffset per DMA instruction, relative to the tile's first halo pixel (launch constants)
V_ZBASE = (142, 143)           # zero-fill LDS address of buffer 0 / 1 (this wave's six KiB of the halo region)
V_WOFF = 144                   # 3: weight DMA voffsets
V_PROW, V_PCOL = 147, 153      # 6 + 6: halo pixel row / column of this lane for DMA instruction i
V_GX = 159                     # this lane's image column in the tile that just ended
V_ZERO = 160                   # 4 zeros
V_SOFF = 164                   # 4: store offsets per g of the tile that just ended
V_POFF = 168                   # 2: pooled store offsets per row pair
V_T = 172                      # temporaries 172..187
V_OWB, V_X16, V_X32 = 188, 189, 190      # fwd_out: LDS address of this lane's output-weight quads; 4 * (lane ^ 16), 4 * (lane ^ 32)
V_STAGE = 192                  # 64: ReLU(acc) of the tile that just ended, until its stores are issued
V_LAST = 255
A_BIAS = 0                     # AGPRs 0..15: bias quad per mb of the tile being fetched / started


def _scalar_map(first, singles, pairs, skip=(32, 33)):
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


S = _scalar_map(8,
                singles="LDSB CA CB COUT H W TILESX TILESY NCT TILE TEND NCH CC WAVE TX TY CT FRAME NTX NTY NCTN NFR X0 Y0 BORDER "
                        "CHB T0 T1 T2 T3 FO OB1 OB2".split(),
                pairs=("P0", "INA", "INB", "WB", "BIAS", "OUT", "POOL", "M0_", "M1_", "M2_", "M3_", "M4_", "M5_", "VALID2", "VALID3",
                       "COLOK", "SRC", "WCH", "OUTF", "POOLF", "TOFF", "SM0", "SM1", "SM2", "SM3", "PM0", "PM1")
                      + (("TRACE",) if TRACE else ()))
S["GATE"], S["GATE1"], S["GATEF"], S["GATEF1"], S["LASTF"], S["LASTF1"] = S["POOL"], S["POOL1"], S["POOLF"], S["POOLF1"], S["PM0"], S["PM01"]   # (lin)
S["OB0"] = S["CB"]                                  # (fwd_out: the output bias; CB is dead once NCH is known)
S["KARG"], S["KARG1"] = S["P0"], S["P01"]          # the kernel-argument pointer is dead once the arguments are loaded
S["TS"], S["TS1"], S["TA"], S["TA1"] = S["COLOK"], S["COLOK1"], S["P0"], S["P01"]   # (trace builds: both are dead at the trace points)
S_LAST = max(S.values())
assert S_LAST <= 101, S_LAST


def s(n):
    return f"s{S[n]}"


def s2(n):
    return f"s[{S[n]}:{S[n] + 1}]"


class Body:
    def __init__(self, variant):
        self.variant, self.pool, self.fuse, self.lin = variant, variant == "fwd_pool", variant == "fwd_out", variant == "lin"
        self.L, self.lds, self.nlabel = [], [], 0

    def e(self, t):
        self.L.append(t)

    def label(self, stem):
        self.nlabel += 1
        return f"S2LC_{self.variant}_{stem}_{self.nlabel}"

    def lds_op(self, text, tag):
        self.e(text)
        self.lds.append(tag)

    def wait_lds(self, tag):
        if tag not in self.lds:
            return
        newer = len(self.lds) - 1 - self.lds.index(tag)
        assert newer <= 15
        self.e(f"s_waitcnt lgkmcnt({newer})")
        self.lds = self.lds[len(self.lds) - newer:] if newer else []

    def trace(self, slot, plus_cc=False):
        """experiment builds: wave 0 stores s_memtime to trace[TILE][slot (+ CC)] (24 slots of 8 bytes per tile)"""
        if not TRACE:
            return
        e = self.e
        skip = self.label("notrace")
        e(f"s_memtime {s2('TS')}")
        e(f"s_cmp_eq_u64 {s2('TRACE')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_cmp_eq_u32 {s('WAVE')}, 0")
        e(f"s_cbranch_scc0 {skip}")
        e(f"s_mul_i32 {s('TA')}, {s('TILE')}, 192")
        e(f"s_add_u32 {s('TA')}, {s('TA')}, {8 * slot}")
        if plus_cc:
            e(f"s_lshl_b32 {s('TA1')}, {s('CC')}, 3")
            e(f"s_add_u32 {s('TA')}, {s('TA')}, {s('TA1')}")
        e(f"s_add_u32 {s('TA')}, {s('TRACE')}, {s('TA')}")
        e(f"s_addc_u32 {s('TA1')}, {s('TRACE1')}, 0")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_store_dwordx2 {s2('TS')}, {s2('TA')}, 0")
        e("s_waitcnt lgkmcnt(0)")
        e(f"{skip}:")

    @staticmethod
    def acc(mb, g):
        b = V_ACC + (mb * 4 + g) * 4
        return f"v[{b}:{b + 3}]"

    @staticmethod
    def stage(mb, g):
        return V_STAGE + (mb * 4 + g) * 4

    # ------------------------------------------------------------------ set-up of the tile whose chunks are fetched next
    def fetch_setup(self, tx, ty, ct, frame):
        """TOFF = byte offset of the tile's first halo pixel (y0 - 1, x0 - 1) in an input tensor (it may lie before the frame:
        only lanes of existing pixels are enabled), SRC = input A + TOFF, WCH = the tile's first weight chunk, the DMA masks
        M_i and BORDER, the bias quads (AGPRs; waited for with the chunk's DMA)"""
        e = self.e
        e(f"s_lshl_b32 {s('X0')}, {s(tx)}, 4")
        e(f"s_lshl_b32 {s('Y0')}, {s(ty)}, 4")
        e(f"s_sub_u32 {s('T2')}, {s('Y0')}, 1")
        e(f"s_sub_u32 {s('T3')}, {s('X0')}, 1")
        e(f"s_mul_i32 {s('T0')}, {s('T2')}, {s('W')}")              # two's complement throughout: (y0 - 1) W + (x0 - 1) >= -W - 1
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T3')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('CA')}")
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 2")
        e(f"s_ashr_i32 {s('T1')}, {s('T0')}, 31")
        e(f"s_mul_hi_u32 {s('TOFF1')}, {s('FO')}, {s(frame)}")
        e(f"s_mul_i32 {s('TOFF')}, {s('FO')}, {s(frame)}")
        e(f"s_add_u32 {s('TOFF')}, {s('TOFF')}, {s('T0')}")
        e(f"s_addc_u32 {s('TOFF1')}, {s('TOFF1')}, {s('T1')}")
        e(f"s_add_u32 {s('SRC')}, {s('INA')}, {s('TOFF')}")
        e(f"s_addc_u32 {s('SRC1')}, {s('INA1')}, {s('TOFF1')}")
        # WCH = w + ct * NCH * 36864
        e(f"s_mul_i32 {s('T0')}, {s(ct)}, {s('NCH')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {W_BYTES}")
        e(f"s_add_u32 {s('WCH')}, {s('WB')}, {s('T0')}")
        e(f"s_addc_u32 {s('WCH1')}, {s('WB1')}, 0")
        if not self.lin:      # the tile's bias quads
            e(f"s_lshl_b32 {s('T0')}, {s(ct)}, 8")
            e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_Q16}")
            for mb in range(4):
                e(f"global_load_dwordx4 a[{A_BIAS + 4 * mb}:{A_BIAS + 4 * mb + 3}], v{V_T}, {s2('BIAS')} offset:{64 * mb}")
        # interior tile (the whole 18x18 halo lies in the image): every existing halo pixel is fetched
        border, done = self.label("border"), self.label("masks")
        e(f"s_cmp_eq_u32 {s(tx)}, 0")
        e(f"s_cbranch_scc1 {border}")
        e(f"s_cmp_eq_u32 {s(ty)}, 0")
        e(f"s_cbranch_scc1 {border}")
        e(f"s_add_u32 {s('T0')}, {s('X0')}, 17")
        e(f"s_cmp_gt_u32 {s('T0')}, {s('W')}")
        e(f"s_cbranch_scc1 {border}")
        e(f"s_add_u32 {s('T0')}, {s('Y0')}, 17")
        e(f"s_cmp_gt_u32 {s('T0')}, {s('H')}")
        e(f"s_cbranch_scc1 {border}")
        e(f"s_mov_b32 {s('BORDER')}, 0")
        for i in range(6):
            e(f"s_mov_b64 {s2(f'M{i}_')}, " + ("-1" if i < 2 else s2("VALID2") if i == 2 else s2("VALID3")))
        e(f"s_branch {done}")
        e(f"{border}:")
        e(f"s_mov_b32 {s('BORDER')}, 1")
        for i in range(6):
            gy, gx = V_T, V_T + 1
            e(f"v_add_u32 v{gy}, {s('T2')}, v{V_PROW + i}")
            e(f"v_add_u32 v{gx}, {s('T3')}, v{V_PCOL + i}")
            e(f"v_cmp_gt_u32 vcc, {s('H')}, v{gy}")              # unsigned compare: also rejects -1
            e(f"s_mov_b64 {s2('P0')}, vcc")
            e(f"v_cmp_gt_u32 vcc, {s('W')}, v{gx}")
            e(f"s_and_b64 {s2('P0')}, {s2('P0')}, vcc")
            e(f"v_cmp_gt_u32 vcc, 18, v{V_PROW + i}")             # the pixel exists: p < 324  <=>  prow < 18
            e(f"s_and_b64 {s2(f'M{i}_')}, {s2('P0')}, vcc")
        e(f"{done}:")

    def advance_in_tile(self):
        """the chunk after chunk CC of the same tile: next 16 channels (64 bytes further), or the first 16 of input B"""
        e = self.e
        same = self.label("sameinput")
        e(f"s_add_u32 {s('T0')}, {s('CC')}, 1")
        e(f"s_add_u32 {s('SRC')}, {s('SRC')}, 64")
        e(f"s_addc_u32 {s('SRC1')}, {s('SRC1')}, 0")
        e(f"s_cmp_lg_u32 {s('T0')}, {s('CHB')}")
        e(f"s_cbranch_scc1 {same}")
        e(f"s_add_u32 {s('SRC')}, {s('INB')}, {s('TOFF')}")        # CB == CA: the halo offsets stay valid
        e(f"s_addc_u32 {s('SRC1')}, {s('INB1')}, {s('TOFF1')}")
        e(f"{same}:")
        e(f"s_add_u32 {s('WCH')}, {s('WCH')}, {W_BYTES}")
        e(f"s_addc_u32 {s('WCH1')}, {s('WCH1')}, 0")

    # ------------------------------------------------------------------ DMA of one chunk into buffer b
    def dma_groups(self, b):
        """the 15 LDS-DMA instructions of a chunk, each a self-contained group (sets m0 / exec itself)"""
        groups = []
        for i in range(6):      # halo: KiB (wave*6 + i) of the buffer's halo region; waves 0..2 six each, wave 3 only i < 3
            groups.append([f"s_mul_i32 {s('T0')}, {s('WAVE')}, {6 * 1024}",
                           f"s_add_u32 {s('T0')}, {s('T0')}, {b * BUF + i * 1024}",
                           f"s_add_u32 m0, {s('T0')}, {s('LDSB')}",
                           f"s_mov_b64 exec, {s2(f'M{i}_')}",
                           f"global_load_lds_dwordx4 v{V_HOFF + i}, {s2('SRC')}",
                           "s_mov_b64 exec, -1"])
        for gi in range(3):     # weights: this wave's 9 KiB = 4 + 4 + 1
            for k in range(4 if gi < 2 else 1):
                groups.append([f"s_mul_i32 {s('T0')}, {s('WAVE')}, 9216",
                               f"s_add_u32 {s('T0')}, {s('T0')}, {b * BUF + HALO_BYTES + gi * 4096}",
                               f"s_add_u32 m0, {s('T0')}, {s('LDSB')}",
                               "s_nop 0",
                               f"global_load_lds_dwordx4 v{V_WOFF + gi}, {s2('WCH')} offset:{1024 * k}"])
        return groups

    def zero_fill(self, b):
        """border tiles: zero this wave's part of buffer b's halo region (before the masked DMA lands on it)"""
        e = self.e
        skip = self.label("interior")
        e(f"s_cmp_eq_u32 {s('BORDER')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        for i in range(6):
            if i == 2:
                e(f"s_mov_b64 exec, {s2('VALID2')}")
            if i == 3:
                e("s_mov_b64 exec, -1")
                e(f"s_cmp_eq_u32 {s('WAVE')}, 3")          # wave 3 has only 21 - 18 = 3 KiB of the 20.25
                e(f"s_cbranch_scc1 {skip}_w3")
            e(f"ds_write_b128 v{V_ZBASE[b]}, v[{V_ZERO}:{V_ZERO + 3}] offset:{1024 * i}")
        e(f"{skip}_w3:")
        e("s_waitcnt lgkmcnt(0)")
        e(f"{skip}:")

    # ------------------------------------------------------------------ stores of the tile that ended (ReLU(acc) in V_STAGE)
    def store_groups(self):
        groups = []
        for g in range(4):
            for mb in range(4):
                st = self.stage(mb, g)
                groups.append([f"s_mov_b64 exec, {s2(f'SM{g}')}",
                               f"global_store_dwordx4 v{V_SOFF + g}, v[{st}:{st + 3}], {s2('OUTF')} offset:{64 * mb}",
                               "s_mov_b64 exec, -1"])
        if self.pool:
            # MaxPool2d(2): rows (g, g+1) pair up in registers, columns px ^ 1 across lanes (DPP); even px lanes store
            for gp in range(2):
                for mb in range(4):
                    a0, a1 = self.stage(mb, 2 * gp), self.stage(mb, 2 * gp + 1)
                    grp = [f"v_max_f32 v{V_T + 8 + r}, v{a0 + r}, v{a1 + r}" for r in range(4)]
                    grp += ["s_nop 1"]        # (five wait states between a write of exec and a DPP instruction: the drain runs groups back to back)
                    grp += [f"v_mov_b32_dpp v{V_T + 12 + r}, v{V_T + 8 + r} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" for r in range(4)]
                    grp += [f"v_max_f32 v{V_T + 8 + r}, v{V_T + 8 + r}, v{V_T + 12 + r}" for r in range(4)]
                    grp += [f"s_mov_b64 exec, {s2(f'PM{gp}')}",
                            f"global_store_dwordx4 v{V_POFF + gp}, v[{V_T + 8}:{V_T + 11}], {s2('POOLF')} offset:{64 * mb}",
                            "s_mov_b64 exec, -1"]
                    groups.append(grp)
        return groups

    # ------------------------------------------------------------------ one chunk's MFMAs
    def tap_reads(self, b, t, oset, tag):
        """the eight operand reads of tap t of buffer b into operand set oset, as (instruction, tracker entry)"""
        dy, dx = t // 3, t % 3
        base = V_OPS[oset]
        out = []
        for mb in range(4):
            out.append((f"ds_read_b128 v[{base + 4 * mb}:{base + 4 * mb + 3}], v{V_ABASE[b]} offset:{(t * 4 + mb) * 1024}", (tag, t, "A", mb)))
        for g in range(4):
            out.append((f"ds_read_b128 v[{base + 16 + 4 * g}:{base + 19 + 4 * g}], v{V_BBASE[b]} offset:{((g + dy) * 18 + dx) * 64}",
                        (tag, t, "B", g)))
        return out

    def first_tap_reads(self, b):
        """entry state of every chunk copy: tap 0 of buffer b on its way into operand set b"""
        for text, entry in self.tap_reads(b, 0, b, "entry"):
            self.lds_op(text, entry)

    def chunk(self, b, groups, read_next=True, every=EVERY):
        """576 MFMAs on buffer b.  On entry tap 0's operand reads are in flight (first_tap_reads); tap t uses operand set
        (t + b) % 2.  `groups`: instruction groups, one behind every `every`-th MFMA of taps 0..7.  Before tap 8: wait for this
        wave's DMA, barrier (the other buffer is complete, nobody reads this one's halo and weights any more except through
        registers already loaded), then the next chunk's first tap is read behind tap 8's MFMAs."""
        e = self.e
        self.lds = [("entry", 0, "A", mb) for mb in range(4)] + [("entry", 0, "B", g) for g in range(4)]
        pending = list(groups)
        nmf = 0
        for t in range(9):
            late = []
            if t < 8:
                for text, entry in self.tap_reads(b, t + 1, (t + 1 + b) % 2, "c"):
                    self.lds_op(text, entry)
            self.wait_lds(("entry" if t == 0 else "c", t, "B", 3))
            if t == 8:
                assert not self.lds
                e("s_waitcnt vmcnt(0)")
                e("s_barrier")
                self.trace(2, plus_cc=True)
                if read_next:
                    late = self.tap_reads(1 - b, 0, 1 - b, "entry")
            base = V_OPS[(t + b) % 2]
            for ks in range(4):
                for mb in range(4):
                    for g in range(4):
                        c = self.acc(mb, g)
                        e(f"v_mfma_f32_16x16x4_f32 {c}, v{base + 4 * mb + ks}, v{base + 16 + 4 * g + ks}, {c}")
                        nmf += 1
                        if late:
                            self.lds_op(*late.pop(0))
                        elif t < 8 and pending and nmf % every == 0:
                            for x in pending.pop(0):
                                e(x)
        assert not pending, len(pending)
        assert len(self.lds) == (8 if read_next else 0)

    # ------------------------------------------------------------------ end of a tile
    def gate_groups(self):
        """lin: the gate quads of this tile -> staging registers, from the tile's last chunk (LASTF: all ones there if there is
        a gate, else zero); SOFF / SM of the tile are set up just before that chunk"""
        groups = []
        for g in range(4):
            for mb in range(4):
                st = self.stage(mb, g)
                groups.append([f"s_and_b64 exec, {s2(f'SM{g}')}, {s2('LASTF')}",
                               f"global_load_dwordx4 v[{st}:{st + 3}], v{V_SOFF + g}, {s2('GATEF')} offset:{64 * mb}",
                               "s_mov_b64 exec, -1"])
        return groups

    def tile_end(self):
        """ReLU(acc) -> staging registers; store offsets and masks of this tile (its stores are issued later)"""
        e = self.e
        e("s_nop 7")
        e("s_nop 7")
        if self.lin:
            # (addresses: store_setup ran before the tile's last chunk)  stage = acc, or acc where the gate is positive
            plain, done = self.label("nogate"), self.label("staged")
            e(f"s_cmp_eq_u64 {s2('GATE')}, 0")
            e(f"s_cbranch_scc1 {plain}")
            for i in range(64):
                e(f"v_cmp_lt_f32 vcc, 0, v{V_STAGE + i}")
                e(f"v_cndmask_b32 v{V_STAGE + i}, 0, v{V_ACC + i}, vcc")
            e(f"s_branch {done}")
            e(f"{plain}:")
            for i in range(64):
                e(f"v_mov_b32 v{V_STAGE + i}, v{V_ACC + i}")
            e(f"{done}:")
            return
        self.store_setup()
        for mb in range(4):
            for g in range(4):
                for r in range(4):
                    e(f"v_max_f32 v{self.stage(mb, g) + r}, 0, v{V_ACC + (mb * 4 + g) * 4 + r}")
        if self.fuse:
            self.output_conv()
        if self.pool:
            self.pool_setup()

    def store_setup(self):
        """OUTF, SOFF_g, SM_g (and T2 = image row of g = 0, T3 = 256 ct, V_GX) of tile (TX, TY, CT, FRAME)"""
        e = self.e
        # OUTF = out + frame * H*W*cout*4
        e(f"s_mul_i32 {s('T0')}, {s('H')}, {s('W')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('COUT')}")
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 2")
        e(f"s_mul_hi_u32 {s('T1')}, {s('T0')}, {s('FRAME')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('FRAME')}")
        e(f"s_add_u32 {s('OUTF')}, {s('OUT')}, {s('T0')}")
        e(f"s_addc_u32 {s('OUTF1')}, {s('OUT1')}, {s('T1')}")
        # lane's column and row 0 of this wave; store offset of row g: ((gy W + gx) cout + 64 ct) * 4 + 16 q  (+ 64 mb as immediate)
        e(f"s_lshl_b32 {s('X0')}, {s('TX')}, 4")
        e(f"s_lshl_b32 {s('Y0')}, {s('TY')}, 4")
        e(f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 2")
        e(f"s_add_u32 {s('T2')}, {s('T2')}, {s('Y0')}")           # gy of g = 0
        e(f"v_add_u32 v{V_GX}, {s('X0')}, v{V_PX}")               # gx
        e(f"v_cmp_gt_u32 vcc, {s('W')}, v{V_GX}")
        e(f"s_mov_b64 {s2('COLOK')}, vcc")
        e(f"s_lshl_b32 {s('T3')}, {s('CT')}, 8")                   # 64 ct * 4
        for g in range(4):
            e(f"s_add_u32 {s('T0')}, {s('T2')}, {g}")
            e(f"s_cmp_lt_u32 {s('T0')}, {s('H')}")
            e(f"s_cselect_b64 {s2(f'SM{g}')}, {s2('COLOK')}, 0")
            e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('W')}")
            e(f"v_add_u32 v{V_T + 1}, {s('T0')}, v{V_GX}")
            e(f"v_mul_lo_u32 v{V_T + 1}, v{V_T + 1}, {s('COUT')}")
            e(f"v_lshl_add_u32 v{V_T + 1}, v{V_T + 1}, 2, v{V_Q16}")
            e(f"v_add_u32 v{V_SOFF + g}, {s('T3')}, v{V_T + 1}")
        if self.lin:          # the gate has the output's shape: same offsets from its own frame base
            e(f"s_sub_u32 {s('T0')}, {s('OUTF')}, {s('OUT')}")
            e(f"s_subb_u32 {s('T1')}, {s('OUTF1')}, {s('OUT1')}")
            e(f"s_add_u32 {s('GATEF')}, {s('GATE')}, {s('T0')}")
            e(f"s_addc_u32 {s('GATEF1')}, {s('GATE1')}, {s('T1')}")

    def pool_setup(self):
        """POOLF, POFF_gp, PM_gp of the tile (after store_setup: T2, T3, V_GX)"""
        e = self.e
        e(f"s_lshr_b32 {s('Y0')}, {s('H')}, 1")                # H2
        e(f"s_lshr_b32 {s('X0')}, {s('W')}, 1")                # W2
        e(f"s_mul_i32 {s('T0')}, {s('Y0')}, {s('X0')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('COUT')}")
        e(f"s_lshl_b32 {s('T0')}, {s('T0')}, 2")
        e(f"s_mul_hi_u32 {s('T1')}, {s('T0')}, {s('FRAME')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('FRAME')}")
        e(f"s_add_u32 {s('POOLF')}, {s('POOL')}, {s('T0')}")
        e(f"s_addc_u32 {s('POOLF1')}, {s('POOL1')}, {s('T1')}")
        e(f"v_lshrrev_b32 v{V_T + 2}, 1, v{V_GX}")             # px2 = gx / 2
        e(f"v_cmp_gt_u32 vcc, {s('X0')}, v{V_T + 2}")
        e(f"s_mov_b64 {s2('P0')}, vcc")
        e(f"v_and_b32 v{V_T + 3}, 1, v{V_GX}")
        e(f"v_cmp_eq_u32 vcc, 0, v{V_T + 3}")
        e(f"s_and_b64 {s2('P0')}, {s2('P0')}, vcc")
        for gp in range(2):
            e(f"s_add_u32 {s('T0')}, {s('T2')}, {2 * gp}")
            e(f"s_lshr_b32 {s('T0')}, {s('T0')}, 1")             # py2
            e(f"s_cmp_lt_u32 {s('T0')}, {s('Y0')}")
            e(f"s_cselect_b64 {s2(f'PM{gp}')}, {s2('P0')}, 0")
            e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('X0')}")
            e(f"v_add_u32 v{V_T + 4}, {s('T0')}, v{V_T + 2}")
            e(f"v_mul_lo_u32 v{V_T + 4}, v{V_T + 4}, {s('COUT')}")
            e(f"v_lshl_add_u32 v{V_T + 4}, v{V_T + 4}, 2, v{V_Q16}")
            e(f"v_add_u32 v{V_POFF + gp}, {s('T3')}, v{V_T + 4}")

    def output_conv(self):
        """fwd_out: out3[pixel][o] = outb[o] + sum_c outw[o][c] h[c] -- per lane an fma chain over its 16 channels (mb, r in
        order, from 0), then + the lanes 16 and 32 further (q), as conv3x3_kernel<true>; lanes q == 0 store 12 bytes.
        Registers: the operand set the tile's last tap used (V_OPS[1]) and the temporaries; T2 = row of g = 0, SM_g = row masks."""
        e = self.e
        P, X, WB = V_OPS[1], V_OPS[1] + 16, (V_T, V_OPS[1] + 16)     # p[g][o] at P + 4 g + o; shuffle temporaries; weight quads of o
        def read_w(o):
            for mb in range(4):
                e(f"ds_read_b128 v[{WB[o % 2] + 4 * mb}:{WB[o % 2] + 4 * mb + 3}], v{V_OWB} offset:{256 * o + 64 * mb}")
        e("s_waitcnt lgkmcnt(0)")          # (the next tile's first operands: the counter is only exact without them)
        read_w(0)
        read_w(1)
        # OUT3F = out3 + frame * H*W*12 (POOL / POOLF hold out3 in this variant)
        e(f"s_mul_i32 {s('T0')}, {s('H')}, {s('W')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, 12")
        e(f"s_mul_hi_u32 {s('T1')}, {s('T0')}, {s('FRAME')}")
        e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('FRAME')}")
        e(f"s_add_u32 {s('POOLF')}, {s('POOL')}, {s('T0')}")
        e(f"s_addc_u32 {s('POOLF1')}, {s('POOL1')}, {s('T1')}")
        for o in range(3):
            e("s_waitcnt lgkmcnt(4)" if o < 2 else "s_waitcnt lgkmcnt(0)")
            for g in range(4):
                for mb in range(4):
                    for r in range(4):
                        c = "0" if (mb, r) == (0, 0) else f"v{P + 4 * g + o}"
                        e(f"v_fma_f32 v{P + 4 * g + o}, v{WB[o % 2] + 4 * mb + r}, v{self.stage(mb, g) + r}, {c}")
            if o == 0:
                read_w(2)
        for dist in (V_X16, V_X32):
            for g in range(4):
                for o in range(3):
                    e(f"ds_bpermute_b32 v{X + 4 * g + o}, v{dist}, v{P + 4 * g + o}")
            e("s_waitcnt lgkmcnt(0)")
            for g in range(4):
                for o in range(3):
                    e(f"v_add_f32 v{P + 4 * g + o}, v{P + 4 * g + o}, v{X + 4 * g + o}")
        for g in range(4):
            for o in range(3):
                e(f"v_add_f32 v{P + 4 * g + o}, {s(f'OB{o}')}, v{P + 4 * g + o}")
        for g in range(4):
            e(f"s_add_u32 {s('T0')}, {s('T2')}, {g}")
            e(f"s_mul_i32 {s('T0')}, {s('T0')}, {s('W')}")
            e(f"v_add_u32 v{V_T}, {s('T0')}, v{V_GX}")
            e(f"v_mul_lo_u32 v{V_T}, v{V_T}, 12")
            e(f"s_and_b64 exec, {s2(f'SM{g}')}, 0xffff")
            e(f"global_store_dwordx3 v{V_T}, v[{P + 4 * g}:{P + 4 * g + 2}], {s2('POOLF')}")
            e("s_mov_b64 exec, -1")
        # the activation itself is stored only if the caller keeps it
        keep = self.label("keep")
        e(f"s_cmp_lg_u64 {s2('OUT')}, 0")
        e(f"s_cbranch_scc1 {keep}")
        for g in range(4):
            e(f"s_mov_b64 {s2(f'SM{g}')}, 0")
        e(f"{keep}:")

    # ------------------------------------------------------------------ the whole body
    def emit(self):
        e = self.e
        # ---- operands and kernel arguments
        for dst, src in (("LDSB", "ldsbase"), ("TILE", "tile0"), ("TEND", "tile_end"), ("TX", "tx0"), ("TY", "ty0"), ("CT", "ct0"),
                         ("FRAME", "frame0"), ("WAVE", "wave")):
            e(f"s_mov_b32 {s(dst)}, %[{src}]")
        e(f"s_mov_b64 {s2('KARG')}, %[karg]")
        for dst, off in (("INA", "oinA"), ("INB", "oinB"), ("WB", "ow"), ("BIAS", "obias"), ("OUT", "oout")) + (() if self.fuse or self.lin else (("POOL", "opool"),)):
            e(f"s_load_dwordx2 {s2(dst)}, {s2('KARG')}, %[{off}]")
        for dst, off in (("CA", "oCA"), ("CB", "oCB"), ("COUT", "ocout"), ("H", "oH"), ("W", "oW"), ("TILESX", "otx"), ("TILESY", "oty"),
                         ("NCT", "onct")):
            e(f"s_load_dword {s(dst)}, {s2('KARG')}, %[{off}]")
        if TRACE:
            e(f"s_load_dwordx2 {s2('TRACE')}, {s2('KARG')}, %[otrace]")
        if self.lin:
            e(f"s_load_dwordx2 {s2('GATE')}, {s2('KARG')}, %[ogate]")
        if self.fuse:      # POOL <- out3; output weights and bias through PM0 / PM1 (zeroed below)
            e(f"s_load_dwordx2 {s2('POOL')}, {s2('KARG')}, %[oout3]")
            e(f"s_load_dwordx2 {s2('PM0')}, {s2('KARG')}, %[ooutw]")
            e(f"s_load_dwordx2 {s2('PM1')}, {s2('KARG')}, %[ooutb]")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_add_u32 {s('NCH')}, {s('CA')}, {s('CB')}")
        e(f"s_lshr_b32 {s('NCH')}, {s('NCH')}, 4")
        if self.fuse:
            e(f"s_load_dword {s('OB0')}, {s2('PM1')}, 0")
            e(f"s_load_dword {s('OB1')}, {s2('PM1')}, 4")
            e(f"s_load_dword {s('OB2')}, {s2('PM1')}, 8")
        e(f"s_lshr_b32 {s('CHB')}, {s('CA')}, 4")
        e(f"s_mul_i32 {s('FO')}, {s('H')}, {s('W')}")              # bytes of one input frame (the launcher checks < 2^31)
        e(f"s_mul_i32 {s('FO')}, {s('FO')}, {s('CA')}")
        e(f"s_lshl_b32 {s('FO')}, {s('FO')}, 2")
        # ---- lane constants
        e(f"v_mbcnt_lo_u32_b32 v{V_T}, -1, 0")                     # lane (every VGPR is spoken for: no vector operand)
        e(f"v_mbcnt_hi_u32_b32 v{V_T}, -1, v{V_T}")
        e(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_T}")
        e(f"v_and_b32 v{V_T + 1}, 3, v{V_T}")
        e(f"v_lshlrev_b32 v{V_PART16}, 4, v{V_T + 1}")
        e(f"v_and_b32 v{V_PX}, 15, v{V_T}")
        e(f"v_lshrrev_b32 v{V_T + 1}, 4, v{V_T}")
        e(f"v_lshlrev_b32 v{V_Q16}, 4, v{V_T + 1}")
        for b in range(2):
            e(f"s_add_u32 {s('T0')}, {s('LDSB')}, {b * BUF + HALO_BYTES}")
            e(f"v_add_u32 v{V_ABASE[b]}, {s('T0')}, v{V_LANE16}")
            # B base: ((4 wave) * 18 + px) * 64 + 16 q
            e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, {4 * 18 * 64}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {b * BUF}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDSB')}")
            e(f"v_lshl_add_u32 v{V_T + 2}, v{V_PX}, 6, v{V_Q16}")
            e(f"v_add_u32 v{V_BBASE[b]}, {s('T0')}, v{V_T + 2}")
            e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, {6 * 1024}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {b * BUF}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDSB')}")
            e(f"v_add_u32 v{V_ZBASE[b]}, {s('T0')}, v{V_LANE16}")
        for gi in range(3):
            e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, 9216")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {gi * 4096}")
            e(f"v_add_u32 v{V_WOFF + gi}, {s('T0')}, v{V_LANE16}")
        e(f"v_lshrrev_b32 v{V_T + 1}, 2, v{V_T}")                  # lane / 4
        for i in range(6):
            e(f"s_mul_i32 {s('T0')}, {s('WAVE')}, {6 * 16}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {16 * i}")
            e(f"v_add_u32 v{V_T + 2}, {s('T0')}, v{V_T + 1}")       # p
            e(f"v_mul_u32_u24 v{V_T + 3}, 3641, v{V_T + 2}")
            e(f"v_lshrrev_b32 v{V_PROW + i}, 16, v{V_T + 3}")       # p / 18 (exact for p < 400)
            e(f"v_mul_u32_u24 v{V_T + 3}, 18, v{V_PROW + i}")
            e(f"v_sub_u32 v{V_PCOL + i}, v{V_T + 2}, v{V_T + 3}")
            # halo offset relative to the tile's first halo pixel: ((prow W + pcol) CA) * 4 + 16 (lane & 3)
            e(f"v_mul_lo_u32 v{V_T + 3}, v{V_PROW + i}, {s('W')}")
            e(f"v_add_u32 v{V_T + 3}, v{V_T + 3}, v{V_PCOL + i}")
            e(f"v_mul_lo_u32 v{V_T + 3}, v{V_T + 3}, {s('CA')}")
            e(f"v_lshl_add_u32 v{V_HOFF + i}, v{V_T + 3}, 2, v{V_PART16}")
        for r in range(4):
            e(f"v_mov_b32 v{V_ZERO + r}, 0")
        if self.fuse:
            # the 768 bytes of output weights -> LDS behind the two buffers (wave 0, 48 lanes); the wait and barrier of the prime
            e(f"s_add_u32 {s('T0')}, {s('LDSB')}, {2 * BUF}")
            e(f"v_add_u32 v{V_OWB}, {s('T0')}, v{V_Q16}")
            e(f"v_xor_b32 v{V_X16}, 16, v{V_T}")
            e(f"v_lshlrev_b32 v{V_X16}, 2, v{V_X16}")
            e(f"v_xor_b32 v{V_X32}, 32, v{V_T}")
            e(f"v_lshlrev_b32 v{V_X32}, 2, v{V_X32}")
            now = self.label("noweights")
            e(f"s_cmp_lg_u32 {s('WAVE')}, 0")
            e(f"s_cbranch_scc1 {now}")
            e(f"s_mov_b32 m0, {s('T0')}")
            e("s_mov_b32 exec_hi, 0xffff")
            e(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('PM0')}")
            e("s_mov_b64 exec, -1")
            e(f"{now}:")
            e("s_waitcnt lgkmcnt(0)")          # the output bias
        # which DMA lanes carry an existing halo pixel: all but wave 3's i = 2 (pixels 320..335 -> lanes < 16) and i >= 3 (none)
        e(f"s_mov_b64 {s2('VALID2')}, -1")
        e(f"s_mov_b64 {s2('VALID3')}, -1")
        e(f"s_cmp_eq_u32 {s('WAVE')}, 3")
        e(f"s_cselect_b64 {s2('VALID2')}, 0xffff, {s2('VALID2')}")
        e(f"s_cselect_b64 {s2('VALID3')}, 0, {s2('VALID3')}")
        # no tile has ended yet: the first tile's chunk 0 issues its stores with every lane off
        for n in ("SM0", "SM1", "SM2", "SM3", "PM0", "PM1"):
            e(f"s_mov_b64 {s2(n)}, 0")
        # ---- first tile: set up, fetch chunk 0 into buffer 0
        self.fetch_setup("TX", "TY", "CT", "FRAME")
        self.zero_fill(0)
        for g in self.dma_groups(0):
            for x in g:
                e(x)
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        self.lds = []
        self.first_tap_reads(0)

        e("S2LC_TILE:")
        self.trace(0)
        # accumulators = bias of this channel tile (the C++ kernel's initial value; loaded when the tile's fetch was set up)
        for mb in range(4):
            for g in range(4):
                for r in range(4):
                    if self.lin:
                        e(f"v_mov_b32 v{V_ACC + (mb * 4 + g) * 4 + r}, 0")
                    else:
                        e(f"v_accvgpr_read_b32 v{V_ACC + (mb * 4 + g) * 4 + r}, a{A_BIAS + 4 * mb + r}")
        e(f"s_mov_b32 {s('CC')}, 0")
        self.trace(1)
        # ---- chunk 0 (buffer 0): fetches chunk 1 of the same tile, and issues the stores of the tile before
        self.advance_in_tile()
        self.zero_fill(1)
        dma, st = self.dma_groups(1), self.store_groups()
        mixed = []
        while dma or st:
            if dma:
                mixed.append(dma.pop(0))
            if st:
                mixed.append(st.pop(0))
        self.chunk(0, mixed, every=EVERY_FIRST)
        e(f"s_mov_b32 {s('CC')}, 1")
        e("s_branch S2LC_ODD")
        # ---- even chunks after the first (buffer 0): the next chunk is always in the same tile
        e("S2LC_EVEN:")
        self.advance_in_tile()
        self.zero_fill(1)
        self.chunk(0, self.dma_groups(1))
        e(f"s_add_u32 {s('CC')}, {s('CC')}, 1")
        # ---- odd chunks (buffer 1): the next chunk is chunk CC + 1 of this tile, or chunk 0 of the next tile, or nothing
        e("S2LC_ODD:")
        nexttile, fetch, nofetch, after = self.label("nexttile"), self.label("fetch"), self.label("nofetch"), self.label("after")
        e(f"s_add_u32 {s('T0')}, {s('CC')}, 1")
        e(f"s_cmp_lt_u32 {s('T0')}, {s('NCH')}")
        e(f"s_cbranch_scc0 {nexttile}")
        self.advance_in_tile()
        if self.lin:
            e(f"s_mov_b64 {s2('LASTF')}, 0")
        e(f"s_branch {fetch}")
        e(f"{nexttile}:")
        if self.lin:      # the tile's last chunk: its stores' addresses (the stores of the tile before are all issued), its gate
            self.store_setup()
            e(f"s_cmp_lg_u64 {s2('GATE')}, 0")
            e(f"s_cselect_b64 {s2('LASTF')}, -1, 0")
        e(f"s_add_u32 {s('T0')}, {s('TILE')}, 1")
        e(f"s_cmp_lt_u32 {s('T0')}, {s('TEND')}")
        e(f"s_cbranch_scc0 {nofetch}")
        # coordinates of the next tile (x fastest, then y, channel tile, frame) and its set-up
        ok = self.label("coords")
        e(f"s_add_u32 {s('NTX')}, {s('TX')}, 1")
        e(f"s_mov_b32 {s('NTY')}, {s('TY')}")
        e(f"s_mov_b32 {s('NCTN')}, {s('CT')}")
        e(f"s_mov_b32 {s('NFR')}, {s('FRAME')}")
        e(f"s_cmp_lt_u32 {s('NTX')}, {s('TILESX')}")
        e(f"s_cbranch_scc1 {ok}")
        e(f"s_mov_b32 {s('NTX')}, 0")
        e(f"s_add_u32 {s('NTY')}, {s('NTY')}, 1")
        e(f"s_cmp_lt_u32 {s('NTY')}, {s('TILESY')}")
        e(f"s_cbranch_scc1 {ok}")
        e(f"s_mov_b32 {s('NTY')}, 0")
        e(f"s_add_u32 {s('NCTN')}, {s('NCTN')}, 1")
        e(f"s_cmp_lt_u32 {s('NCTN')}, {s('NCT')}")
        e(f"s_cbranch_scc1 {ok}")
        e(f"s_mov_b32 {s('NCTN')}, 0")
        e(f"s_add_u32 {s('NFR')}, {s('NFR')}, 1")
        e(f"{ok}:")
        self.fetch_setup("NTX", "NTY", "NCTN", "NFR")
        e(f"{fetch}:")
        self.zero_fill(0)
        dma, gates = self.dma_groups(0), (self.gate_groups() if self.lin else [])
        mixed = []
        while dma or gates:
            if dma:
                mixed.append(dma.pop(0))
            if gates:
                mixed.append(gates.pop(0))
        self.chunk(1, mixed)
        e(f"s_branch {after}")
        e(f"{nofetch}:")
        self.chunk(1, self.gate_groups() if self.lin else [], read_next=False)
        e(f"{after}:")
        e(f"s_add_u32 {s('CC')}, {s('CC')}, 1")
        e(f"s_cmp_lt_u32 {s('CC')}, {s('NCH')}")
        e("s_cbranch_scc1 S2LC_EVEN")
        # ---- the tile is complete
        self.tile_end()
        self.trace(20)
        e(f"s_add_u32 {s('TILE')}, {s('TILE')}, 1")
        e(f"s_mov_b32 {s('TX')}, {s('NTX')}")
        e(f"s_mov_b32 {s('TY')}, {s('NTY')}")
        e(f"s_mov_b32 {s('CT')}, {s('NCTN')}")
        e(f"s_mov_b32 {s('FRAME')}, {s('NFR')}")
        e(f"s_cmp_lt_u32 {s('TILE')}, {s('TEND')}")
        e("s_cbranch_scc1 S2LC_TILE")
        # ---- the last tile's stores
        for grp in self.store_groups():
            for x in grp:
                e(x)
        e("s_waitcnt vmcnt(0)")
        if TRACE:
            e("s_dcache_wb")
        return self.L


OPERANDS = """      :
      : [karg] "s"(karg), [ldsbase] "s"(ldsbase), [tile0] "s"(tile0), [tile_end] "s"(tile_end), [tx0] "s"(tx0), [ty0] "s"(ty0),
        [ct0] "s"(ct0), [frame0] "s"(frame0), [wave] "s"(wave),
        [oinA] "n"(offsetof(ConvArgs, inA)), [oinB] "n"(offsetof(ConvArgs, inB)), [ow] "n"(offsetof(ConvArgs, w)),
        [obias] "n"(offsetof(ConvArgs, bias)), [oout] "n"(offsetof(ConvArgs, out)), [opool] "n"(offsetof(ConvArgs, pool)),
        [oCA] "n"(offsetof(ConvArgs, CA)), [oCB] "n"(offsetof(ConvArgs, CB)), [ocout] "n"(offsetof(ConvArgs, cout)),
        [oH] "n"(offsetof(ConvArgs, H)), [oW] "n"(offsetof(ConvArgs, W)), [otx] "n"(offsetof(ConvArgs, tiles_x)),
        [oty] "n"(offsetof(ConvArgs, tiles_y)), [onct] "n"(offsetof(ConvArgs, n_ct)), [oout3] "n"(offsetof(ConvArgs, out3)),
        [ooutw] "n"(offsetof(ConvArgs, outw)), [ooutb] "n"(offsetof(ConvArgs, outb)), [ogate] "n"(offsetof(ConvArgs, gate))
"""


def main(outdir):
    total = 0
    for variant in ("fwd", "fwd_pool", "fwd_out", "lin"):
        lines = Body(variant).emit()
        # labels are per variant
        lines = [x.replace("S2LC_TILE", f"S2LC_{variant}_TILE").replace("S2LC_EVEN", f"S2LC_{variant}_EVEN").replace("S2LC_ODD", f"S2LC_{variant}_ODD")
                 for x in lines]
        clob = ([f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(A_BIAS, A_BIAS + 16)]
                + [f"s{r}" for r in range(8, S_LAST + 1) if r not in (32, 33)] + ["vcc", "scc", "memory"])
        out = ["// GENERATED by csrc/gen_conv_body.py -- do not edit; the generator is the source.", "asm volatile("]
        out += [f'    "{x}\\n\\t"' for x in lines]
        out.append(OPERANDS.rstrip("\n") + (',\n        [otrace] "n"(offsetof(ConvArgs, trace))' if TRACE else ""))
        out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
        with open(os.path.join(outdir, f"conv_body_{variant}.inc"), "w") as f:
            f.write("\n".join(out) + "\n")
        total += len(lines)
    return total


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"conv_body_*.inc: {main(d)} instructions")
