"""Generates bwd16_body.inc: the body of s2l::b16::bwd_asm_bf16_kernel (csrc/train_bf16.hip) as ONE fixed-register gfx950 assembly
text -- the bf16 training backward (dz chain) with 64 rows per wave, one wave per SIMD: the twin of gen_fwd16_body.py.

    g_7 = (Wout^T drgb) . m_7;   g_{l-1} = (W_l^T g_l) . m_{l-1}, l = 7..1 (l = 5: the h_4 half of pts_linears[5]);
    d audio = G5[:, audio]^T g_5 + G0[:, audio]^T g_0,  summed over the 256 rows of the tile (one frame's rows).

Same arithmetic in the same order as bwd_bf16_kernel for the dz images (bit-identical: tests switch between the two).  The audio
gradient differs in form, not in mathematics: the C++ kernel writes it per ROW (dxa [N,64] fp32, the second audio stage a
read-modify-write: 1.8 GB of traffic, + 0.6 GB read by the per-frame column sums that follow); here the two audio products
accumulate in ONE fp32 accumulator set (AGPRs) and leave the kernel as the column sums of the tile, 256 B per 256 rows.  The
caller uses this kernel when a tile never straddles two frames (4 H W a multiple of 256) and the C++ kernel otherwise.

  * workgroup = 4 waves = 256 rows; wave w owns row groups g0 = 8 tile + 2 w, g0 + 1 (nb = 0, 1); an A quad feeds two MFMAs;
  * stage = two 32-row slabs of one W_l^T (or of the audio columns), 32 KiB, by LDS-DMA into one of two buffers -- 30 per tile
    in consumption order (s2l_bf16.h: OFF_BWD_H); which buffer is a run-time alternation (the two audio stages make the parity
    of a layer's stages differ between layers): v_swap_b32 of the two A-quad address registers, one scalar toggle;
  * activation (gradient) sets: P in VGPRs 0..127, Q in AGPRs 0..127 ([nb][k-step 16][4]): g_7, g_5, g_3, g_1 live in P, the
    others in Q; a layer reads one and writes the other;
  * epilogue per pair: v_cvt_pk_bf16_f32, v_lshrrev + v_and (the pair's two mask bits -> 0 / 1 halves), v_pk_mul_lo_u16; the
    stage's mask dwords (one per lane and row group) are fetched a stage ahead;
  * the dz images of a stage are stored during the NEXT stage, like the forward's;
  * g_7: four pseudo-stages on the 8 KiB of Wout^T kept in LDS, K = 16 (3 used), B operand built from drgb (fetched during the
    tile before);
  * tile end: audio accumulators -> LDS -> column sums over the wave's 64 rows (fixed order) -> the four waves' sums -> 64
    floats of dxa_tiles[tile].
"""
import os
import sys

TRACE = False
SLAB_H = 16384
STAGE = 2 * SLAB_H                      # 32768
NSTAGES = 30
LDS_U0 = 2 * STAGE                      # Wout^T slabs: 8 x 1 KiB
LDS_SUMS = LDS_U0 + 8192                # [4 waves][64] floats
LDS_BYTES = LDS_SUMS + 1024

# ---- vector registers
V_P = 0                        # 128: set P [nb][t][4]
V_ACC = 128                    # 64: acc[(which * 2 + nb) * 16 + r]
V_MW = 192                     # 4: mask dwords [slot][nb]
V_DRGB = 196                   # 8: drgb of the tile to come [nb][3 of 4] (register tuples must start on an even register)
V_B0 = 204                     # 8: U0's B operand [nb][4]
V_LANE16, V_LANE16B, V_LANE4, V_AHC, V_AHN, V_U0, V_ROWOFF, V_DIM4, V_SUMRD, V_DUMP = 212, 213, 214, 215, 216, 217, 218, 220, 221, 222
V_T = 224                      # temporaries 224..235
V_LAST = 235
# ---- accumulation registers
A_Q = 0                        # 128: set Q
A_AUD = 128                    # 64: audio accumulators [(which * 2 + nb) * 16 + r]
NSLOT = 5
A_A = 192                      # 8 NSLOT: A quads [slot][which][4]
A_LAST = A_A + 8 * NSLOT - 1


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
                singles="LDSB WAVE TILE NTILES GRID L U T0 T1 T2 T3 ONE PENDOK G0 NROWS BUFN BUFSUM MDST NXT".split(),
                pairs=("KARG", "WB", "DRGB", "MASKS", "DZT", "DXA", "LSTR", "MSTR", "WP", "WQ", "DZL", "MKL", "MKN", "PEND", "EX", "TA"))
S_LAST = max(S.values())
assert S_LAST <= 101, S_LAST


def s(n):
    return f"s{S[n]}"


def s2(n):
    return f"s[{S[n]}:{S[n] + 1}]"


def vq(b):
    return f"v[{b}:{b + 3}]"


def aq(b):
    return f"a[{b}:{b + 3}]"


class Body:
    def __init__(self):
        self.L, self.lds, self.nlabel = [], [], 0

    def e(self, t):
        self.L.append(t)

    def label(self, stem):
        self.nlabel += 1
        return f"S2LB_{stem}_{self.nlabel}"

    def lds_op(self, text, tag):
        self.e(text)
        self.lds.append(tag)

    def wait_lds(self, tag):
        if tag not in self.lds:
            return
        newer = min(15, len(self.lds) - 1 - self.lds.index(tag))
        self.e(f"s_waitcnt lgkmcnt({newer})")
        self.lds = self.lds[len(self.lds) - newer:] if newer else []

    # ------------------------------------------------------------------ registers
    @staticmethod
    def acc(which, nb):
        b = V_ACC + (which * 2 + nb) * 16
        return f"v[{b}:{b + 15}]"

    @staticmethod
    def aud(which, nb):
        b = A_AUD + (which * 2 + nb) * 16
        return f"a[{b}:{b + 15}]"

    @staticmethod
    def setreg(setname, nb, t):
        return vq(V_P + nb * 64 + t * 4) if setname == "P" else aq(A_Q + nb * 64 + t * 4)

    # ------------------------------------------------------------------ the next stage's DMA (every real stage consumes one)
    def stage_program(self):
        """set-up and LDS-DMA of stage U + 1 (mod 30) into the OTHER buffer (BUFN), as groups of <= 5 instructions, one group per
        MFMA slot: a wave copies bytes [8192 wave, + 8192) of the 32 KiB -- eight 1-KiB instructions"""
        groups = [[f"s_add_u32 {s('T0')}, {s('U')}, 1", f"s_cmp_eq_u32 {s('T0')}, {NSTAGES}", f"s_cselect_b32 {s('T0')}, 0, {s('T0')}",
                   f"s_lshl_b32 {s('T1')}, {s('T0')}, 15"],
                  [f"s_add_u32 {s('WP')}, {s('WB')}, {s('T1')}", f"s_addc_u32 {s('WP1')}, {s('WB1')}, 0", f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 13"],
                  [f"s_add_u32 {s('WQ')}, {s('WP')}, {s('T2')}", f"s_addc_u32 {s('WQ1')}, {s('WP1')}, 0", f"s_add_u32 {s('MDST')}, {s('BUFN')}, {s('T2')}",
                   f"s_mov_b32 m0, {s('MDST')}"]]
        loads = lambda offs: [f"global_load_lds_dwordx4 v{V_LANE16}, {s2('WQ')} offset:{o}" for o in offs]
        groups.append(loads((0, 1024)))
        groups.append(loads((2048, 3072)))
        groups.append([f"s_add_u32 {s('WQ')}, {s('WQ')}, 4096", f"s_addc_u32 {s('WQ1')}, {s('WQ1')}, 0", f"s_add_u32 {s('MDST')}, {s('MDST')}, 4096",
                       f"s_mov_b32 m0, {s('MDST')}"])
        groups.append(loads((0, 1024)))
        groups.append(loads((2048, 3072)))
        return groups

    def mask_prefetch(self, slot, base, q):
        """mask dwords of the stage whose epilogue uses `slot`: one per lane and row group, at base + nb * 1024 + q * 256"""
        return [f"global_load_dword v{V_MW + 2 * slot + nb}, v{V_LANE4}, {s2(base)} offset:{nb * 1024 + q * 256}" for nb in range(2)]

    def store_items(self, setname, qprev, guarded):
        """the eight image stores of the stage before (blocks 2 qprev, 2 qprev + 1 of `setname`) at PEND"""
        items = []
        for nb in range(2):
            for which in range(2):
                for half in range(2):
                    t = 2 * (2 * qprev + which) + half
                    src = self.setreg(setname, nb, t)
                    it = []
                    skip = self.label("nostore")
                    if guarded:
                        it += [f"s_cmp_eq_u32 {s('PENDOK')}, 0", f"s_cbranch_scc1 {skip}"]
                    it.append(f"global_store_dwordx4 v{V_LANE16B if nb else V_LANE16}, {src}, {s2('PEND')} offset:{which * 2048 + half * 1024}" + os.environ.get("S2L_BWD_STORE_MOD", " nt"))
                    if guarded:
                        it.append(f"{skip}:")
                    items.append(it)
        return items

    # ------------------------------------------------------------------ k-loop of a real stage (16 k-steps, 64 MFMAs)
    def kloop(self, inset, items, audio=None):
        """A quads through NSLOT AGPR slots, NSLOT - 1 k-steps ahead; items: (MFMA number, [instructions]) in ascending order.
        audio: None (hidden stage: accumulators in VGPRs, C = 0 for the first MFMA of each) | 'first' | 'second' (the audio
        accumulators in AGPRs; 'second' continues them)"""
        e = self.e

        def read(i):
            for which in range(2):
                dst = A_A + ((i % NSLOT) * 2 + which) * 4
                self.lds_op(f"ds_read_b128 {aq(dst)}, v{V_AHC} offset:{which * SLAB_H + i * 1024}", ("A", i, which))
        self.lds = [("A", i, which) for i in range(NSLOT - 1) for which in range(2)]       # read by first_reads of the stage before
        pending = list(items)
        nmf = 0
        for i in range(16):
            if i + NSLOT - 1 < 16:
                read(i + NSLOT - 1)
            self.wait_lds(("A", i, 1))
            for which in range(2):
                a = aq(A_A + ((i % NSLOT) * 2 + which) * 4)
                for nb in range(2):
                    c = self.aud(which, nb) if audio else self.acc(which, nb)
                    srcc = "0" if (i == 0 and audio != "second") else c
                    e(f"v_mfma_f32_32x32x16_bf16 {c}, {a}, {self.setreg(inset, nb, i)}, {srcc}")
                    nmf += 1
                    while pending and nmf >= pending[0][0]:
                        for x in pending.pop(0)[1]:
                            e(x)
        for _, grp in pending:
            for x in grp:
                e(x)

    def first_reads(self, areg=V_AHN):
        """A quads of the first NSLOT - 1 k-steps of the stage computed next: inside a real stage that is the OTHER buffer (V_AHN);
        behind the g_7 pseudo-stages it is the current one"""
        for i in range(NSLOT - 1):
            for which in range(2):
                self.e(f"ds_read_b128 {aq(A_A + ((i % NSLOT) * 2 + which) * 4)}, v{areg} offset:{which * SLAB_H + i * 1024}")

    def stage_end(self):
        """the other buffer becomes the current one; the stage counter moves on"""
        e = self.e
        e(f"v_swap_b32 v{V_AHC}, v{V_AHN}")
        e(f"s_sub_u32 {s('BUFN')}, {s('BUFSUM')}, {s('BUFN')}")
        e(f"s_add_u32 {s('U')}, {s('U')}, 1")
        e(f"s_cmp_eq_u32 {s('U')}, {NSTAGES}")
        e(f"s_cselect_b32 {s('U')}, 0, {s('U')}")

    # ------------------------------------------------------------------ epilogue: bf16, mask, out set; images pending
    def epilogue(self, q, outset, slot, items=()):
        e = self.e
        items = list(items)
        e("s_nop 7")
        e("s_nop 7")
        e("s_nop 7")
        for which in range(2):
            R = 2 * q + which
            for g4 in range(2):
                for nb in range(2):
                    ds = []
                    for k in range(4):
                        d8 = 4 * g4 + k
                        src = V_ACC + (which * 2 + nb) * 16 + 2 * d8
                        dst = nb * 64 + (2 * R + g4) * 4 + k
                        d = f"v{V_P + dst}" if outset == "P" else f"v{V_T + k}"
                        ds.append((d, dst, 8 * which + d8))
                        e(f"v_cvt_pk_bf16_f32 {d}, v{src}, v{src + 1}")
                    for k, (d, _, pair) in enumerate(ds):
                        e(f"v_lshrrev_b32 v{V_T + 4 + k}, {15 - pair}, v{V_MW + 2 * slot + nb}")
                    for k in range(4):
                        e(f"v_and_b32 v{V_T + 4 + k}, {s('ONE')}, v{V_T + 4 + k}")
                    for k, (d, _, _) in enumerate(ds):
                        e(f"v_pk_mul_lo_u16 {d}, {d}, v{V_T + 4 + k}")
                    if outset == "Q":
                        for d, dst, _ in ds:
                            e(f"v_accvgpr_write_b32 a{A_Q + dst}, {d}")
                    if items:
                        for x in items.pop(0):
                            e(x)
        for grp in items:
            for x in grp:
                e(x)
        # the images of this stage are stored during the next stage
        e(f"s_add_u32 {s('PEND')}, {s('DZL')}, {q * 4096}")
        e(f"s_addc_u32 {s('PEND1')}, {s('DZL1')}, 0")
        e(f"s_mov_b32 {s('PENDOK')}, 1")

    # ------------------------------------------------------------------ addresses of a layer
    def layer_setup(self, layer_reg):
        """DZL = dzT + layer * LSTR + 16384 g0, MKL = masks + layer * MSTR + 1024 g0, MKN = MKL - MSTR (the layer below)"""
        e = self.e
        for dst, base, stride, shift in (("DZL", "DZT", "LSTR", 14), ("MKL", "MASKS", "MSTR", 10)):
            e(f"s_mul_hi_u32 {s('T1')}, {s(stride)}, {s(layer_reg)}")
            e(f"s_mul_i32 {s('T0')}, {s(stride)}, {s(layer_reg)}")
            e(f"s_mul_i32 {s('T2')}, {s(stride + '1')}, {s(layer_reg)}")
            e(f"s_add_u32 {s('T1')}, {s('T1')}, {s('T2')}")
            e(f"s_add_u32 {s(dst)}, {s(base)}, {s('T0')}")
            e(f"s_addc_u32 {s(dst + '1')}, {s(base + '1')}, {s('T1')}")
            e(f"s_lshr_b32 {s('T1')}, {s('G0')}, {32 - shift}")
            e(f"s_lshl_b32 {s('T0')}, {s('G0')}, {shift}")
            e(f"s_add_u32 {s(dst)}, {s(dst)}, {s('T0')}")
            e(f"s_addc_u32 {s(dst + '1')}, {s(dst + '1')}, {s('T1')}")
        e(f"s_sub_u32 {s('MKN')}, {s('MKL')}, {s('MSTR')}")
        e(f"s_subb_u32 {s('MKN1')}, {s('MKL1')}, {s('MSTR1')}")

    # ------------------------------------------------------------------ g_7: four pseudo-stages on Wout^T (LDS), K = 16
    def u_body(self):
        """target layer 7 (T3 = 7 for layer_setup); out set P; masks (7, 0) are in slot 0 (fetched during the tile before)"""
        e = self.e
        e(f"s_mov_b32 {s('T3')}, 7")
        self.layer_setup("T3")
        # B operand from the raw drgb: lanes hh == 0 of live rows hold (d0, d1, d2), everything else 0
        for nb in range(2):
            e(f"v_cvt_pk_bf16_f32 v{V_B0 + 4 * nb}, v{V_DRGB + 4 * nb}, v{V_DRGB + 4 * nb + 1}")
            e(f"v_cvt_pk_bf16_f32 v{V_B0 + 4 * nb + 1}, v{V_DRGB + 4 * nb + 2}, 0")
            e(f"v_mov_b32 v{V_B0 + 4 * nb + 2}, 0")
            e(f"v_mov_b32 v{V_B0 + 4 * nb + 3}, 0")
        for q in range(4):
            for which in range(2):
                e(f"ds_read_b128 {aq(A_A + which * 4)}, v{V_U0} offset:{(2 * q + which) * 1024}")
            # masks of the next epilogue: (7, q + 1), or layer 6's q = 0 for the first real stage
            for x in (self.mask_prefetch((q + 1) & 1, "MKL", q + 1) if q < 3 else self.mask_prefetch(0, "MKN", 0)):
                e(x)
            e("s_waitcnt lgkmcnt(0)")
            for which in range(2):
                for nb in range(2):
                    e(f"v_mfma_f32_32x32x16_bf16 {self.acc(which, nb)}, {aq(A_A + which * 4)}, {vq(V_B0 + 4 * nb)}, 0")
            st = self.store_items("P", q - 1, guarded=False) if q else []       # (q = 0: the tile before has no images pending)
            e("s_waitcnt vmcnt(0)")       # this stage's mask dwords (loads and stores are not ordered among each other: no partial count)
            self.epilogue(q, "P", q & 1, st)
        e("s_waitcnt vmcnt(0)")
        self.first_reads(V_AHC)       # the first real stage of the tile sits in the CURRENT buffer

    # ------------------------------------------------------------------ a hidden layer: four stages
    def layer_body(self, kind):
        """kind B: reads P, writes Q; C: reads Q, writes P.  s[L] = the layer l whose W_l^T multiplies; target layer l - 1."""
        e = self.e
        inset, outset = ("P", "Q") if kind == "B" else ("Q", "P")
        e(f"s_sub_u32 {s('T3')}, {s('L')}, 1")
        self.layer_setup("T3")
        for q in range(4):
            st = self.store_items(inset, 3, guarded=True) if q == 0 else self.store_items(outset, q - 1, guarded=False)
            prog = self.stage_program()
            mixed = [(1 + k, g) for k, g in enumerate(prog)]
            n0 = len(mixed) + 1
            if q < 3:
                mixed.append((n0, self.mask_prefetch((q + 1) & 1, "MKL", q + 1)))
            else:      # the layer below's q = 0 (nothing below layer 0: the loads would still be valid addresses? no -- skip)
                skip = self.label("nomask")
                mixed.append((n0, [f"s_cmp_eq_u32 {s('L')}, 1", f"s_cbranch_scc1 {skip}", *self.mask_prefetch(0, "MKN", 0), f"{skip}:"]))
            self.kloop(inset, mixed)
            e("s_waitcnt vmcnt(0) lgkmcnt(0)")
            self.lds = []
            e("s_barrier")
            self.first_reads()
            self.epilogue(q, outset, q & 1, st)
            self.stage_end()

    # ------------------------------------------------------------------ the audio stages
    def audio_stage(self, inset, second):
        e = self.e
        st = self.store_items(inset, 3, guarded=False)      # the images of the layer that just ended
        prog = self.stage_program()
        mixed = [(1 + k, g) for k, g in enumerate(prog)]
        mixed += [(len(mixed) + 2 + 3 * k, g) for k, g in enumerate(st)]
        if second:
            mixed += [(len(prog) + 30, self.next_tile_prefetch())]
        self.kloop(inset, mixed, audio="second" if second else "first")
        e(f"s_mov_b32 {s('PENDOK')}, 0")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lds = []
        e("s_barrier")
        if not second:
            self.first_reads()
            self.stage_end()

    def next_tile_prefetch(self):
        """drgb and the (7, 0) masks of this wave's rows in the NEXT tile of this workgroup (NXT = 1 if there is one)"""
        skip = self.label("nonext")
        out = [f"s_add_u32 {s('T0')}, {s('TILE')}, {s('GRID')}", f"s_cmp_lt_u32 {s('T0')}, {s('NTILES')}", f"s_cselect_b32 {s('NXT')}, 1, 0",
               f"s_cbranch_scc0 {skip}"]
        out += self.tile_inputs("T0")
        out.append(f"{skip}:")
        return out

    def tile_inputs(self, tile_reg):
        """loads of drgb -> V_DRGB and masks (7, 0) -> slot 0 for tile s[tile_reg]; clobbers T0..T3, MKN, EX, vcc"""
        out = []
        for k in range(8):
            out.append(f"v_mov_b32 v{V_DRGB + k}, 0")
        # first row of this wave in that tile: 256 tile + 64 wave; row of lane = + 32 nb + n; byte offset 12 row (rows < 2^24)
        out += [f"s_lshl_b32 {s('T1')}, {s(tile_reg)}, 8", f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 6", f"s_add_u32 {s('T1')}, {s('T1')}, {s('T2')}"]
        for nb in range(2):
            out += [f"v_add_u32 v{V_T}, {s('T1')}, v{V_ROWOFF}" if nb == 0 else f"v_add_u32 v{V_T}, 32, v{V_T}",
                    f"v_cmp_gt_u32 vcc, {s('NROWS')}, v{V_T}",
                    f"v_mul_lo_u32 v{V_T + 1}, v{V_T}, 12",
                    f"s_and_b64 {s2('EX')}, vcc, 0xffffffff",      # hh == 0: lanes 0..31
                    f"s_mov_b64 exec, {s2('EX')}",
                    f"global_load_dwordx3 v[{V_DRGB + 4 * nb}:{V_DRGB + 4 * nb + 2}], v{V_T + 1}, {s2('DRGB')}",
                    "s_mov_b64 exec, -1"]
        # masks + 7 MSTR + 1024 (8 tile + 2 wave)
        out += [f"s_mul_hi_u32 {s('T3')}, {s('MSTR')}, 7", f"s_mul_i32 {s('T2')}, {s('MSTR')}, 7", f"s_mul_i32 {s('T1')}, {s('MSTR1')}, 7",
                f"s_add_u32 {s('T3')}, {s('T3')}, {s('T1')}", f"s_add_u32 {s('MKN')}, {s('MASKS')}, {s('T2')}", f"s_addc_u32 {s('MKN1')}, {s('MASKS1')}, {s('T3')}",
                f"s_lshl_b32 {s('T1')}, {s(tile_reg)}, 3", f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 1", f"s_add_u32 {s('T1')}, {s('T1')}, {s('T2')}",
                f"s_lshr_b32 {s('T2')}, {s('T1')}, 22", f"s_lshl_b32 {s('T1')}, {s('T1')}, 10",
                f"s_add_u32 {s('MKN')}, {s('MKN')}, {s('T1')}", f"s_addc_u32 {s('MKN1')}, {s('MKN1')}, {s('T2')}"]
        out += self.mask_prefetch(0, "MKN", 0)
        return out

    # ------------------------------------------------------------------ tile end: column sums of the audio gradient
    def tile_sums(self):
        """audio accumulators (fp32, [which][nb][16] per lane: dims 32 which + (r & 3) + 8 (r >> 2) + 4 hh of rows n, n + 32) ->
        sums over the tile's 256 rows -> dxa_tiles[tile][64].  Buffer BUFN... no: the stage buffers are busy (the next stage has
        landed in one, the other was read a moment ago); the dump goes to the buffer the audio stage just finished with, which
        nobody touches until the next real stage's DMA -- issued after the barrier below."""
        e = self.e
        e("s_nop 7")
        e("s_nop 7")
        e("s_nop 7")
        for which in range(2):
            for r in range(16):
                i = which * 16 + r
                e(f"v_accvgpr_read_b32 v{V_T}, a{A_AUD + (which * 2) * 16 + r}")
                e(f"v_accvgpr_read_b32 v{V_T + 1}, a{A_AUD + (which * 2 + 1) * 16 + r}")
                e("s_nop 1")
                e(f"v_add_f32 v{V_ACC + i}, v{V_T}, v{V_T + 1}")
        # dump: lane's 32 values at V_DUMP (= current buffer + 8192 wave + 128 lane)
        e(f"v_add_u32 v{V_T + 2}, v{V_AHC}, v{V_DUMP}")
        for k in range(8):
            e(f"ds_write_b128 v{V_T + 2}, {vq(V_ACC + 4 * k)} offset:{16 * k}")
        e("s_waitcnt lgkmcnt(0)")
        # lane (hh, idx): sum over n of dump[n + 32 hh][idx], in row order
        e(f"v_add_u32 v{V_T + 3}, v{V_AHC}, v{V_SUMRD}")
        for n in range(32):
            e(f"ds_read_b32 v{V_ACC + 32 + n}, v{V_T + 3} offset:{128 * n}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_mov_b32 v{V_T}, v{V_ACC + 32}")
        for n in range(1, 32):
            e(f"v_add_f32 v{V_T}, v{V_T}, v{V_ACC + 32 + n}")
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 8")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('LDSB')}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, {LDS_SUMS}")          # (beyond the 16-bit offset field of a DS instruction)
        e(f"v_add_u32 v{V_T + 1}, {s('T0')}, v{V_DIM4}")
        e(f"ds_write_b32 v{V_T + 1}, v{V_T}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        skip = self.label("notwave0")
        e(f"s_cmp_lg_u32 {s('WAVE')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_add_u32 {s('T0')}, {s('LDSB')}, {LDS_SUMS}")
        e(f"v_add_u32 v{V_T + 1}, {s('T0')}, v{V_LANE4}")
        for w in range(4):
            e(f"ds_read_b32 v{V_T + 4 + w}, v{V_T + 1} offset:{256 * w}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_add_f32 v{V_T + 4}, v{V_T + 4}, v{V_T + 5}")
        e(f"v_add_f32 v{V_T + 4}, v{V_T + 4}, v{V_T + 6}")
        e(f"v_add_f32 v{V_T + 4}, v{V_T + 4}, v{V_T + 7}")
        e(f"s_lshr_b32 {s('T1')}, {s('TILE')}, 24")
        e(f"s_lshl_b32 {s('T0')}, {s('TILE')}, 8")
        e(f"s_add_u32 {s('TA')}, {s('DXA')}, {s('T0')}")
        e(f"s_addc_u32 {s('TA1')}, {s('DXA1')}, {s('T1')}")
        e(f"global_store_dword v{V_LANE4}, v{V_T + 4}, {s2('TA')}")
        e(f"{skip}:")
        e("s_barrier")      # the sums block and the dump buffer are free again

    # ------------------------------------------------------------------ the whole body
    def emit(self):
        e = self.e
        for dst, src in (("LDSB", "ldsbase"), ("WAVE", "wave"), ("TILE", "tile0"), ("GRID", "grid")):
            e(f"s_mov_b32 {s(dst)}, %[{src}]")
        e(f"s_mov_b64 {s2('KARG')}, %[karg]")
        for dst, off in (("WB", "owb"), ("DRGB", "odrgb"), ("MASKS", "omasks"), ("DZT", "odzt"), ("DXA", "odxa"), ("LSTR", "olstr"), ("MSTR", "omstr")):
            e(f"s_load_dwordx2 {s2(dst)}, {s2('KARG')}, %[{off}]")
        e(f"s_load_dword {s('NTILES')}, {s2('KARG')}, %[ontiles]")
        e(f"s_load_dword {s('NROWS')}, {s2('KARG')}, %[onrows]")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_lshl_b64 {s2('LSTR')}, {s2('LSTR')}, 1")            # halves -> bytes
        e(f"s_lshl_b64 {s2('MSTR')}, {s2('MSTR')}, 3")            # uint64 -> bytes
        e(f"s_mov_b32 {s('ONE')}, 0x00010001")
        e(f"s_mov_b32 {s('PENDOK')}, 0")
        e(f"s_mov_b32 {s('U')}, 0")
        # WB: the stage stream starts behind the forward stages and the U0 slabs (byte offsets passed as immediates)
        e(f"s_add_u32 {s('WQ')}, {s('WB')}, %[ou0]")              # Wout^T slabs
        e(f"s_addc_u32 {s('WQ1')}, {s('WB1')}, 0")
        e(f"s_add_u32 {s('WB')}, {s('WB')}, %[ostages]")
        e(f"s_addc_u32 {s('WB1')}, {s('WB1')}, 0")
        # ---- lane constants
        e(f"v_mbcnt_lo_u32_b32 v{V_T}, -1, 0")
        e(f"v_mbcnt_hi_u32_b32 v{V_T}, -1, v{V_T}")               # lane
        e(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_T}")
        e(f"v_add_u32 v{V_LANE16B}, 16384, v{V_LANE16}")
        e(f"v_lshlrev_b32 v{V_LANE4}, 2, v{V_T}")
        e(f"v_add_u32 v{V_AHC}, {s('LDSB')}, v{V_LANE16}")
        e(f"s_add_u32 {s('T0')}, {s('LDSB')}, {STAGE}")
        e(f"v_add_u32 v{V_AHN}, {s('T0')}, v{V_LANE16}")
        e(f"s_add_u32 {s('BUFN')}, {s('LDSB')}, {STAGE}")
        e(f"s_lshl_b32 {s('BUFSUM')}, {s('LDSB')}, 1")
        e(f"s_add_u32 {s('BUFSUM')}, {s('BUFSUM')}, {STAGE}")
        e(f"s_add_u32 {s('T0')}, {s('LDSB')}, {LDS_U0}")
        e(f"v_add_u32 v{V_U0}, {s('T0')}, v{V_LANE16}")
        e(f"v_and_b32 v{V_ROWOFF}, 31, v{V_T}")                    # n
        # V_DIM4 = 4 * (32 (idx >> 4) + (idx & 3) + 8 ((idx & 15) >> 2) + 4 hh), idx = lane & 31, hh = lane >> 5
        e(f"v_and_b32 v{V_T + 1}, 31, v{V_T}")                     # idx
        e(f"v_lshrrev_b32 v{V_T + 2}, 4, v{V_T + 1}")
        e(f"v_lshlrev_b32 v{V_DIM4}, 5, v{V_T + 2}")               # 32 which
        e(f"v_and_b32 v{V_T + 2}, 3, v{V_T + 1}")
        e(f"v_add_u32 v{V_DIM4}, v{V_DIM4}, v{V_T + 2}")
        e(f"v_bfe_u32 v{V_T + 2}, v{V_T + 1}, 2, 2")
        e(f"v_lshlrev_b32 v{V_T + 2}, 3, v{V_T + 2}")
        e(f"v_add_u32 v{V_DIM4}, v{V_DIM4}, v{V_T + 2}")
        e(f"v_lshrrev_b32 v{V_T + 3}, 5, v{V_T}")                  # hh
        e(f"v_lshlrev_b32 v{V_T + 2}, 2, v{V_T + 3}")
        e(f"v_add_u32 v{V_DIM4}, v{V_DIM4}, v{V_T + 2}")
        e(f"v_lshlrev_b32 v{V_DIM4}, 2, v{V_DIM4}")
        # V_DUMP = 8192 wave + 128 lane - 16 lane (V_AHC already holds 16 lane); V_SUMRD = 8192 wave + 4096 hh + 4 idx - 16 lane
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 13")
        e(f"v_mul_u32_u24 v{V_T + 2}, 112, v{V_T}")
        e(f"v_add_u32 v{V_DUMP}, {s('T0')}, v{V_T + 2}")
        e(f"v_lshlrev_b32 v{V_T + 2}, 12, v{V_T + 3}")
        e(f"v_lshlrev_b32 v{V_T + 4}, 2, v{V_T + 1}")
        e(f"v_add_u32 v{V_T + 2}, v{V_T + 2}, v{V_T + 4}")
        e(f"v_sub_u32 v{V_T + 2}, v{V_T + 2}, v{V_LANE16}")
        e(f"v_add_u32 v{V_SUMRD}, {s('T0')}, v{V_T + 2}")
        # ---- first tile: stage 0 -> buffer 0, Wout^T slabs -> LDS, drgb + masks (7, 0)
        done = "S2LB_DONE"
        e(f"s_cmp_lt_u32 {s('TILE')}, {s('NTILES')}")
        e(f"s_cbranch_scc0 {done}")
        e(f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 13")
        e(f"s_add_u32 {s('WP')}, {s('WB')}, {s('T2')}")
        e(f"s_addc_u32 {s('WP1')}, {s('WB1')}, 0")
        e(f"s_add_u32 {s('MDST')}, {s('LDSB')}, {s('T2')}")
        e(f"s_mov_b32 m0, {s('MDST')}")
        e("s_nop 0")
        for o in (0, 1024, 2048, 3072):
            e(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('WP')} offset:{o}")
        e(f"s_add_u32 {s('WP')}, {s('WP')}, 4096")
        e(f"s_addc_u32 {s('WP1')}, {s('WP1')}, 0")
        e(f"s_add_u32 {s('MDST')}, {s('MDST')}, 4096")
        e(f"s_mov_b32 m0, {s('MDST')}")
        e("s_nop 0")
        for o in (0, 1024, 2048, 3072):
            e(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('WP')} offset:{o}")
        # Wout^T: 8 KiB, wave w copies bytes [2048 w, + 2048)
        e(f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 11")
        e(f"s_add_u32 {s('WQ')}, {s('WQ')}, {s('T2')}")
        e(f"s_addc_u32 {s('WQ1')}, {s('WQ1')}, 0")
        e(f"s_add_u32 {s('MDST')}, {s('LDSB')}, {LDS_U0}")
        e(f"s_add_u32 {s('MDST')}, {s('MDST')}, {s('T2')}")
        e(f"s_mov_b32 m0, {s('MDST')}")
        e("s_nop 0")
        for o in (0, 1024):
            e(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('WQ')} offset:{o}")
        for x in self.tile_inputs("TILE"):
            e(x)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")

        e("S2LB_TILE:")
        e(f"s_lshl_b32 {s('G0')}, {s('TILE')}, 3")
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 1")
        e(f"s_add_u32 {s('G0')}, {s('G0')}, {s('T0')}")
        self.u_body()
        e(f"s_mov_b32 {s('L')}, 7")
        e("S2LB_B:")
        self.layer_body("B")
        e(f"s_cmp_eq_u32 {s('L')}, 1")
        e("s_cbranch_scc1 S2LB_A0")
        e(f"s_sub_u32 {s('L')}, {s('L')}, 1")
        self.layer_body("C")
        e(f"s_sub_u32 {s('L')}, {s('L')}, 1")
        e(f"s_cmp_eq_u32 {s('L')}, 5")
        e("s_cbranch_scc0 S2LB_B")
        self.audio_stage("P", second=False)        # g_5 is in P
        e("s_branch S2LB_B")
        e("S2LB_A0:")
        self.audio_stage("Q", second=True)         # g_0 is in Q
        self.tile_sums()
        # the next real stage (stage 0 of the next tile) landed during the audio stage; its first A quads are read by u_body's tail
        self.stage_end()
        e(f"s_add_u32 {s('TILE')}, {s('TILE')}, {s('GRID')}")
        e(f"s_cmp_eq_u32 {s('NXT')}, 1")
        e("s_cbranch_scc1 S2LB_TILE")
        e("s_waitcnt vmcnt(0)")
        e(f"{done}:")
        return [x for x in self.L if x is not None]


OPERANDS = """      :
      : [karg] "s"(karg), [ldsbase] "s"(ldsbase), [wave] "s"(wave), [tile0] "s"(tile0), [grid] "s"(grid),
        [owb] "n"(offsetof(BwdArgs, wb)), [odrgb] "n"(offsetof(BwdArgs, drgb)), [omasks] "n"(offsetof(BwdArgs, masks)),
        [odzt] "n"(offsetof(BwdArgs, dzT)), [odxa] "n"(offsetof(BwdArgs, dxa)), [olstr] "n"(offsetof(BwdArgs, layer_stride)),
        [omstr] "n"(offsetof(BwdArgs, mask_layer_stride)), [ontiles] "n"(offsetof(BwdArgs, n_tiles)),
        [onrows] "n"(offsetof(BwdArgs, n_rows)), [ou0] "n"(2 * OFF_BWD_U0), [ostages] "n"(2 * OFF_BWD_H)
"""


def main(outdir):
    lines = Body().emit()
    clob = ([f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)]
            + [f"s{r}" for r in range(8, S_LAST + 1) if r not in (32, 33)] + ["vcc", "scc", "memory"])
    out = ["// GENERATED by csrc/gen_bwd16_body.py -- do not edit; the generator is the source.", "asm volatile("]
    out += [f'    "{x}\\n\\t"' for x in lines]
    out.append(OPERANDS.rstrip("\n"))
    out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
    with open(os.path.join(outdir, "bwd16_body.inc"), "w") as f:
        f.write("\n".join(out) + "\n")
    return len(lines)


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"bwd16_body.inc: {main(d)} instructions")
