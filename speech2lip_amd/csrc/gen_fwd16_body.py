"""Generates fwd16_body.inc: the body of s2l::b16::fwd_asm_bf16_kernel (csrc/train_bf16.hip) as ONE fixed-register gfx950 assembly
text -- the bf16 training forward with 64 rows per wave, one wave per SIMD.

Why assembly: inside one wave nothing overlaps (MFMAs, the VALU epilogue, a burst of stores, the stage DMA simply
add up), so the schedule has to be owned: every memory instruction rides behind an MFMA, the k-steps carry nothing but two LDS
reads and four MFMAs, and the two activation sets are pinned to the two register halves.  Same arithmetic in the same order as
fwd_bf16_kernel: activation images, mask dwords and rgb are bit-identical.

  * workgroup = 4 waves = 256 rows; wave w owns row groups g0 = 8 tile + 2 w and g0 + 1 (32 rows each, nb = 0, 1): an A quad
    (weights) read from LDS feeds TWO MFMAs -- half the LDS traffic per MFMA of the 32-row kernel;
  * stage = a quarter layer (output blocks R = 2q, 2q + 1): [X slab R0][X slab R1][H slab R0][H slab R1] (s2l_bf16.h), 48 KiB, by
    LDS-DMA into one of two buffers: the next stage's twelve 1-KiB instructions ride behind MFMAs of this stage's k-loop, the wait
    for them and the stage's one barrier sit right behind the k-loop;
  * accumulators acc[which][nb] in VGPRs, their first MFMA takes the bias quads (read from LDS during the stage before) as C;
  * activation sets: P in VGPRs 0..127, Q in AGPRs 0..127 ([nb][k-step 16][4]); a layer reads one and writes the other
    (layer 0 writes P; odd layers write Q through v_accvgpr_write); the embedded rows bx live in AGPRs 128..191 and are
    prefetched for the next tile during layer 6;
  * epilogue per pair: v_cvt_pk_bf16_f32, v_pk_max_i16 (ReLU), v_pk_min_u16 + v_lshl_or_b32 (mask bit): 4 VALU per two values;
  * the images of a stage are stored during the NEXT stage: four stores behind MFMAs of its k-loop, four between the pairs of its
    epilogue (16 B per lane and clock is all the store path takes).  The stores are non-temporal (`nt`) since round 3: every store
    instruction writes 1 KiB of contiguous memory that nothing reads before the weight-gradient kernels stream it, 9.7 GB later
    (forward 3.06 -> 2.98 ms, backward 2.59 -> 2.48 ms; with round 2's first image layout -- a lane's two 16-byte halves in
    different instructions, meeting only in L2 -- the same modifier had cost a factor of two).

State: bit-identical; 3.04 ms against the C++ kernel's 3.3-3.4 ms.  S2L_FWD_EXP builds ablate stores / epilogue /
MFMAs / DMA / barrier, S2L_FWD_TRACE=1 builds time the phases of every stage (tools/trace_fwd16.py).

Layer bodies: A (layer 0: x only, -> P), B (h from P -> Q: layers 1, 3, 7), C (h from Q -> P: 2, 4, 6), D (layer 5: x + h from P ->
Q); four stages each, sequenced by a scalar layer counter."""
import os
import sys

TRACE = os.environ.get("S2L_FWD_TRACE") == "1"      # experiment builds (tools/trace_fwd16.py): per-stage phase timestamps of wave 0
BX_MODE = os.environ.get("S2L_FWD_BX", "kloop")      # embedded rows of the next tile: "start" = loaded at its start (exposed), "epilogue" / "kloop" = prefetched in layer 6
BX_AT_TILE_START = BX_MODE == "start"   # embedded rows of a tile: loaded at its start (exposed) or prefetched in layer 6
EXP = int(os.environ.get("S2L_FWD_EXP", "0"))      # timing experiments (results wrong): 1 no image stores, 2 no epilogue VALU, 4 no MFMAs, 8 no stage DMA, 16 no barrier
SLAB_H, SLAB_X = 16384, 8192            # bytes
STAGE = 2 * SLAB_X + 2 * SLAB_H         # 49152
LDS_W = 2 * STAGE                       # bias block behind the two stage buffers: [8][256] floats + bout[4]
EVERY = 3

# ---- vector registers
V_P = 0                        # 128: activation set P [nb][t][4]
V_ACC = 128                    # 64: acc[(which * 2 + nb) * 16 + r]
V_BIAS = 192                   # 32: bias regs [which][16] of the stage being computed (srcC of its first MFMAs)
V_LANE16, V_LANE32, V_LANE16B, V_LANE4, V_DMAOFF = 224, 225, 226, 227, 228      # (V_LANE32: unused since the image planes)
V_AX = (229, 230)              # LDS address of this lane's A quads, X part of buffer 0 / 1
V_AH = (231, 232)              # ... H part
V_BIAS0 = 233                  # LDS address of the bias block + 16 hh
V_BN = 234                     # ... + 1024 * layer of the stage whose bias is read next
V_RGBOFF = (235, 236)          # byte offset of this lane's row (nb) in rgb, relative to the tile's first row
V_M = (237, 238)               # mask dwords being built
V_T = 240                      # temporaries 240..251 (quads at 240, 244, 248)
V_LAST = 251
# ---- accumulation registers
A_Q = 0                        # 128: activation set Q
A_BX = 128                     # 64: embedded rows [nb][t 8][4]
NSLOT = int(os.environ.get("S2L_FWD_NSLOT", "5"))   # A-quad slots: reads run NSLOT - 1 k-steps ahead of their MFMAs
A_A = 192                      # 8 NSLOT: A quads [slot][which 2][4]
A_LAST = A_A + 8 * NSLOT - 1
assert A_LAST <= 255


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
                singles="LDSB WAVE TILE NTILES GRID L ST T0 T1 T2 T3 NX NH ONE PENDOK BXF MDST G0 NROWS".split(),
                pairs=("KARG", "WB", "XT", "HT", "MASKS", "RGB", "LSTR", "MSTR", "WP", "HTL", "MKL", "PEND", "XTN", "XTA", "EX", "WQ") + (("TRACE", "TS", "TA") if TRACE else ()))
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
        return f"S2LF_{stem}_{self.nlabel}"

    def lds_op(self, text, tag):
        self.e(text)
        self.lds.append(tag)

    def wait_lds(self, tag):
        if tag not in self.lds:
            return
        newer = min(15, len(self.lds) - 1 - self.lds.index(tag))      # (the counter has four bits: waiting for fewer is stricter)
        self.e(f"s_waitcnt lgkmcnt({newer})")
        self.lds = self.lds[len(self.lds) - newer:] if newer else []

    def wait_all_lds(self):
        if self.lds:
            self.e("s_waitcnt lgkmcnt(0)")
            self.lds = []

    def trace(self, slot):
        """experiment builds: wave 0 stores s_memtime to trace[(TILE * 32 + ST) * 8 + slot]"""
        if not TRACE:
            return
        e = self.e
        skip = self.label("notrace")
        e(f"s_memtime {s2('TS')}")
        e(f"s_cmp_eq_u64 {s2('TRACE')}, 0")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_cmp_eq_u32 {s('WAVE')}, 0")
        e(f"s_cbranch_scc0 {skip}")
        e(f"s_lshl_b32 {s('TA')}, {s('TILE')}, 5")
        e(f"s_add_u32 {s('TA')}, {s('TA')}, {s('ST')}")
        e(f"s_lshl_b32 {s('TA')}, {s('TA')}, 6")
        e(f"s_add_u32 {s('TA')}, {s('TA')}, {8 * slot}")
        e(f"s_add_u32 {s('TA')}, {s('TRACE')}, {s('TA')}")
        e(f"s_addc_u32 {s('TA1')}, {s('TRACE1')}, 0")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_store_dwordx2 {s2('TS')}, {s2('TA')}, 0")
        e("s_waitcnt lgkmcnt(0)")
        e(f"{skip}:")

    # ------------------------------------------------------------------ registers
    @staticmethod
    def acc(which, nb):
        b = V_ACC + (which * 2 + nb) * 16
        return f"v[{b}:{b + 15}]"

    @staticmethod
    def setreg(setname, nb, t):
        """B operand quad of k-step t of row group nb in set P (VGPR) / Q (AGPR) / X (bx, AGPR)"""
        if setname == "P":
            return vq(V_P + nb * 64 + t * 4)
        if setname == "Q":
            return aq(A_Q + nb * 64 + t * 4)
        return aq(A_BX + nb * 32 + t * 4)

    # ------------------------------------------------------------------ per-stage scalar set-up
    def stage_program(self, kind, q):
        """the next stage's set-up (its global address, the other LDS buffer, V_BN = bias base of its layer) and its LDS-DMA, as
        groups of at most five instructions, one group per MFMA slot.  A wave copies CONTIGUOUS pieces -- 4 KiB of the X part,
        8 KiB of the H part -- so that four DMA instructions share one base and one M0 (the instruction offset moves both the
        global and the LDS address): 12 loads and ~22 scalar instructions per stage.  Which parts the next stage needs is
        static except at three layer ends, where the loads are skipped inside their own group."""
        # (x part, h part) of the next stage: True / False / ("eq" | "ne", value of the scalar L for which it is present)
        if kind == "A":
            need = (True, False) if q < 3 else (False, True)
        elif kind == "D":
            need = (True, True) if q < 3 else (False, True)
        elif kind == "C":
            need = (False, True) if q < 3 else (("eq", 4), True)
        else:
            need = (False, True) if q < 2 else ((("eq", 7), True) if q == 2 else (("eq", 7), ("ne", 7)))
        groups = [[f"s_add_u32 {s('T0')}, {s('ST')}, 1", f"s_and_b32 {s('T0')}, {s('T0')}, 31", f"s_mul_i32 {s('T1')}, {s('T0')}, {STAGE}"],
                  [f"s_add_u32 {s('WP')}, {s('WB')}, {s('T1')}", f"s_addc_u32 {s('WP1')}, {s('WB1')}, 0",
                   f"s_add_u32 {s('T3')}, {s('LDSB')}, {((q + 1) & 1) * STAGE}"],
                  [f"s_lshr_b32 {s('T1')}, {s('T0')}, 2", f"s_lshl_b32 {s('T1')}, {s('T1')}, 10", f"v_add_u32 v{V_BN}, {s('T1')}, v{V_BIAS0}"]]

        def loads(present, offs):
            out = []
            skip = self.label("nodma")
            if present is not True:
                out += [f"s_cmp_{'lg' if present[0] == 'eq' else 'eq'}_u32 {s('L')}, {present[1]}", f"s_cbranch_scc1 {skip}"]
            for o in offs:
                if not EXP & 8:
                    out.append(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('WQ')} offset:{o}")
            if present is not True:
                out.append(f"{skip}:")
            return out
        if need[0] is not False:      # X part: bytes [4096 wave, + 4096) of the stage
            groups.append([f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 12", f"s_add_u32 {s('WQ')}, {s('WP')}, {s('T2')}", f"s_addc_u32 {s('WQ1')}, {s('WP1')}, 0",
                           f"s_add_u32 {s('MDST')}, {s('T3')}, {s('T2')}", f"s_mov_b32 m0, {s('MDST')}"])
            groups.append(loads(need[0], (0, 1024)))
            groups.append(loads(need[0], (2048, 3072)))
        if need[1] is not False:      # H part: bytes [16384 + 8192 wave, + 8192)
            groups.append([f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 13", f"s_add_u32 {s('T2')}, {s('T2')}, 16384", f"s_add_u32 {s('WQ')}, {s('WP')}, {s('T2')}",
                           f"s_addc_u32 {s('WQ1')}, {s('WP1')}, 0"])
            groups.append([f"s_add_u32 {s('MDST')}, {s('T3')}, {s('T2')}", f"s_mov_b32 m0, {s('MDST')}"])
            groups.append(loads(need[1], (0, 1024)))
            groups.append(loads(need[1], (2048, 3072)))
            groups.append([f"s_add_u32 {s('WQ')}, {s('WQ')}, 4096", f"s_addc_u32 {s('WQ1')}, {s('WQ1')}, 0", f"s_add_u32 {s('MDST')}, {s('MDST')}, 4096",
                           f"s_mov_b32 m0, {s('MDST')}"])
            groups.append(loads(need[1], (0, 1024)))
            groups.append(loads(need[1], (2048, 3072)))
        return groups

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
                    if not EXP & 1:
                        it.append(f"global_store_dwordx4 v{V_LANE16B if nb else V_LANE16}, {src}, {s2('PEND')} offset:{which * 2048 + half * 1024}" + os.environ.get("S2L_FWD_STORE_MOD", " nt"))
                    if guarded:
                        it.append(f"{skip}:")
                    items.append(it)
        return items

    def bx_items(self, half):
        """embedded rows of the NEXT tile -> bx (AGPRs), eight loads per call; only in layer 6 and if there is a next tile (BXF)"""
        items = []
        for k in range(8 * half, 8 * half + 8):
            nb, t = k >> 3, k & 7
            skip = self.label("nobx")
            items.append([f"s_cmp_eq_u32 {s('BXF')}, 0", f"s_cbranch_scc1 {skip}",
                          f"s_add_u32 {s('XTA')}, {s('XTN')}, {nb * 8192 + (t >> 1) * 2048}",
                          f"s_addc_u32 {s('XTA1')}, {s('XTN1')}, 0",
                          f"global_load_dwordx4 {self.setreg('X', nb, t)}, v{V_LANE16}, {s2('XTA')} offset:{(t & 1) * 1024}",
                          f"{skip}:"])
        return items

    # ------------------------------------------------------------------ k-loop of a stage
    def kloop(self, buf, parts, items, every=EVERY, lds_after=None):
        """parts: list of (part 'x' | 'h', set name of the B operands); A quads through three AGPR slots two k-steps ahead;
        the first MFMA of each accumulator takes the bias registers as C.  `items`: instruction groups, one behind every
        `every`-th MFMA; lds_after = (n, [(text, tag)]): LDS operations issued behind MFMA n (through the tracker)."""
        e = self.e
        steps = []
        for part, setname in parts:
            for t in range(8 if part == "x" else 16):
                steps.append((part, setname, t))

        def read(i):
            part, _, t = steps[i]
            base = (V_AX if part == "x" else V_AH)[buf]
            slab = SLAB_X if part == "x" else SLAB_H
            for which in range(2):
                dst = A_A + ((i % NSLOT) * 2 + which) * 4
                self.lds_op(f"ds_read_b128 {aq(dst)}, v{base} offset:{which * slab + t * 1024}", ("A", i, which))
        # the first NSLOT - 1 k-steps' A quads were read by the stage before (first_reads: behind its barrier, during its epilogue)
        self.lds = [("A", i, which) for i in range(min(NSLOT - 1, len(steps))) for which in range(2)]
        pending = list(items)
        nmf = 0
        for i, (part, setname, t) in enumerate(steps):
            if i + NSLOT - 1 < len(steps):
                read(i + NSLOT - 1)
            self.wait_lds(("A", i, 1))
            for which in range(2):
                a = aq(A_A + ((i % NSLOT) * 2 + which) * 4)
                for nb in range(2):
                    c = self.acc(which, nb)
                    srcc = f"v[{V_BIAS + which * 16}:{V_BIAS + which * 16 + 15}]" if i == 0 else c
                    bsrc = self.setreg("X" if part == "x" else setname, nb, t)
                    if not EXP & 4:
                        e(f"v_mfma_f32_32x32x16_bf16 {c}, {a}, {bsrc}, {srcc}")
                    nmf += 1
                    if lds_after and nmf == lds_after[0]:
                        for text, tag in lds_after[1]:
                            self.lds_op(text, tag)
                    if pending and (nmf >= pending[0][0] if isinstance(pending[0], tuple) else nmf % every == 0):
                        grp = pending.pop(0)
                        for x in (grp[1] if isinstance(grp, tuple) else grp):
                            if x is not None:
                                e(x)
        while pending:
            grp = pending.pop(0)
            for x in (grp[1] if isinstance(grp, tuple) else grp):
                if x is not None:
                    e(x)

    def first_reads(self, buf, part):
        """A quads of the first NSLOT - 1 k-steps of the stage that computes next from buffer `buf`, whose first part is `part`"""
        base = (V_AX if part == "x" else V_AH)[buf]
        slab = SLAB_X if part == "x" else SLAB_H
        for i in range(min(NSLOT - 1, 8 if part == "x" else 16)):
            for which in range(2):
                self.e(f"ds_read_b128 {aq(A_A + ((i % NSLOT) * 2 + which) * 4)}, v{base} offset:{which * slab + i * 1024}")

    def bias_reads(self, qnext):
        """bias quads of the NEXT stage (blocks 2 qnext, 2 qnext + 1 of the layer V_BN points at) -> V_BIAS, as (text, tag)"""
        out = []
        for which in range(2):
            for a4 in range(4):
                out.append((f"ds_read_b128 {vq(V_BIAS + which * 16 + a4 * 4)}, v{V_BN} offset:{(2 * qnext + which) * 128 + a4 * 32}",
                            ("bias", which, a4)))
        return out

    # ------------------------------------------------------------------ epilogue of a stage
    def epilogue(self, q, outset, items=()):
        """bf16, ReLU, mask bits: batches of four pairs (independent instructions back to back; one update of the mask dword per
        batch, in the order of fwd_bf16_kernel: which, pair); items: instruction groups, one behind each batch"""
        e = self.e
        items = list(items)
        e("s_nop 7")
        e("s_nop 7")
        e("s_nop 7")
        for nb in range(2):
            e(f"v_mov_b32 v{V_M[nb]}, 0")
        for which in range(2):
            R = 2 * q + which
            for g4 in range(2):
                for nb in range(2):
                    if EXP & 2:
                        continue
                    ds, ts = [], []
                    for k in range(4):
                        d8 = 4 * g4 + k
                        src = V_ACC + (which * 2 + nb) * 16 + 2 * d8
                        dst = nb * 64 + (2 * R + g4) * 4 + k
                        d = f"v{V_P + dst}" if outset == "P" else f"v{V_T + k}"
                        ds.append((d, dst))
                        ts.append(f"v{V_T + 4 + k}")
                        e(f"v_cvt_pk_bf16_f32 {d}, v{src}, v{src + 1}")
                    for d, _ in ds:
                        e(f"v_pk_max_i16 {d}, {d}, 0")
                    for (d, _), tt in zip(ds, ts):
                        e(f"v_pk_min_u16 {tt}, {d}, {s('ONE')}")
                    if outset == "Q":
                        for d, dst in ds:
                            e(f"v_accvgpr_write_b32 a{A_Q + dst}, {d}")
                    e(f"v_lshl_or_b32 {ts[0]}, {ts[0]}, 1, {ts[1]}")          # t0 t1
                    e(f"v_lshl_or_b32 {ts[2]}, {ts[2]}, 1, {ts[3]}")          # t2 t3
                    e(f"v_lshl_or_b32 {ts[0]}, {ts[0]}, 2, {ts[2]}")          # t0 t1 t2 t3
                    e(f"v_lshl_or_b32 v{V_M[nb]}, v{V_M[nb]}, 4, {ts[0]}")
                    if items:
                        for x in items.pop(0):
                            if x is not None:
                                e(x)
        for grp in items:
            for x in grp:
                if x is not None:
                    e(x)
        for nb in range(2):
            e(f"global_store_dword v{V_LANE4}, v{V_M[nb]}, {s2('MKL')} offset:{nb * 1024 + q * 256}")
        # the images of this stage are stored during the next stage
        e(f"s_add_u32 {s('PEND')}, {s('HTL')}, {q * 4096}")
        e(f"s_addc_u32 {s('PEND1')}, {s('HTL1')}, 0")
        e(f"s_mov_b32 {s('PENDOK')}, 1")

    # ------------------------------------------------------------------ a layer body: four stages
    def layer_setup(self):
        """HTL = hT + L * LSTR + 16384 g0, MKL = masks + L * MSTR + 1024 g0"""
        e = self.e
        for dst, base, stride, shift in (("HTL", "HT", "LSTR", 14), ("MKL", "MASKS", "MSTR", 10)):
            e(f"s_mul_hi_u32 {s('T1')}, {s(stride)}, {s('L')}")
            e(f"s_mul_i32 {s('T0')}, {s(stride)}, {s('L')}")
            e(f"s_mul_i32 {s('T2')}, {s(stride + '1')}, {s('L')}")
            e(f"s_add_u32 {s('T1')}, {s('T1')}, {s('T2')}")
            e(f"s_add_u32 {s(dst)}, {s(base)}, {s('T0')}")
            e(f"s_addc_u32 {s(dst + '1')}, {s(base + '1')}, {s('T1')}")
            e(f"s_lshr_b32 {s('T1')}, {s('G0')}, {32 - shift}")
            e(f"s_lshl_b32 {s('T0')}, {s('G0')}, {shift}")
            e(f"s_add_u32 {s(dst)}, {s(dst)}, {s('T0')}")
            e(f"s_addc_u32 {s(dst + '1')}, {s(dst + '1')}, {s('T1')}")

    def layer_body(self, kind):
        e = self.e
        inset = {"A": "Q", "B": "P", "C": "Q", "D": "P"}[kind]        # (A: only the stores of the tile before read it)
        outset = {"A": "P", "B": "Q", "C": "P", "D": "Q"}[kind]
        parts = {"A": [("x", None)], "B": [("h", "P")], "C": [("h", "Q")], "D": [("x", None), ("h", "P")]}[kind]
        self.layer_setup()
        if kind == "C":       # layer 6: the next tile's embedded rows may be fetched (bx is dead after layer 5)
            e(f"s_add_u32 {s('T0')}, {s('TILE')}, {s('GRID')}")
            e(f"s_cmp_lt_u32 {s('T0')}, {s('NTILES')}")
            e(f"s_cselect_b32 {s('BXF')}, 1, 0")
            e(f"s_cmp_eq_u32 {s('L')}, 6")
            e(f"s_cselect_b32 {s('BXF')}, {s('BXF')}, 0")
            e(f"s_lshl_b32 {s('T2')}, {s('T0')}, 3")                # first row group of this wave in the next tile
            e(f"s_lshl_b32 {s('T3')}, {s('WAVE')}, 1")
            e(f"s_add_u32 {s('T2')}, {s('T2')}, {s('T3')}")
            self.xt_pointer("T2")
        for q in range(4):
            st = self.store_items(inset, 3, guarded=(kind == "A")) if q == 0 else self.store_items(outset, q - 1, guarded=False)
            # tools/trace_fwd16.py: the next stage's set-up and DMA ride behind the first MFMAs, at most four scalar / memory
            # instructions per MFMA (a longer run lets the matrix pipe drain), so that the DMA is old when the wait right behind the
            # k-loop asks for it; the eight stores ride between the batches of the epilogue (behind the wait: they have a whole
            # stage until the next one; the texture path takes 40 cycles per contiguous 1-KiB store)
            prog = self.stage_program(kind, q)
            mixed = [(1 + k, g) for k, g in enumerate(prog)]
            nmf_total = sum(8 if p == "x" else 16 for p, _ in parts) * 4
            nk = int(os.environ.get("S2L_FWD_KSTORES", "0")) if nmf_total >= 64 else 0      # stores that ride in the k-loop
            if os.environ.get("S2L_FWD_KWHERE", "head") == "head":      # ahead of the set-up / DMA groups
                mixed = [(1 + k, st[k]) for k in range(nk)] + [(1 + nk + k, g) for k, g in enumerate(prog)]
            else:
                first = max(len(prog) + 2, nmf_total - 30)
                mixed += [(first + 10 * k, st[k]) for k in range(nk)]
            late = st[nk:]
            if kind == "C" and q < 2 and BX_MODE == "epilogue":
                late = late + self.bx_items(q)
            if kind == "C" and q < 2 and BX_MODE == "kloop":      # behind the DMA groups: nothing else asks for the texture path for a while
                mixed = mixed + [(len(mixed) + 2 + 2 * k, g) for k, g in enumerate(self.bx_items(q))]
            # the next stage's bias quads are read once this stage's first MFMAs (which take V_BIAS as C) have been issued
            self.trace(0)
            self.kloop(q & 1, parts, mixed, 1, lds_after=(8, self.bias_reads((q + 1) & 3)))
            self.trace(1)
            e("s_waitcnt vmcnt(0) lgkmcnt(0)")      # the next stage has landed (this wave's part); nobody reads this buffer any more
            self.trace(2)
            if not EXP & 16:
                e("s_barrier")
            self.trace(3)
            if q < 3:      # the next stage is the next quarter of this layer: its first A quads fly during the epilogue
                self.first_reads((q + 1) & 1, parts[0][0])
            self.epilogue(q, outset, late)
            self.trace(4)
            e(f"s_add_u32 {s('ST')}, {s('ST')}, 1")
            e(f"s_and_b32 {s('ST')}, {s('ST')}, 31")
            if kind == "B" and q == 3:
                self.output_layer()
            if q == 3:     # the next stage opens another layer (buffer 0): which one is a run-time question for two kinds
                if kind in ("A", "D"):
                    self.first_reads(0, "h")
                else:
                    alt, join = self.label("nextx"), self.label("nextjoin")
                    e(f"s_cmp_eq_u32 {s('L')}, {4 if kind == 'C' else 7}")        # C: layer 5 follows layer 4; B: layer 0 (of the next tile) follows layer 7
                    e(f"s_cbranch_scc1 {alt}")
                    self.first_reads(0, "h")
                    e(f"s_branch {join}")
                    e(f"{alt}:")
                    self.first_reads(0, "x")
                    e(f"{join}:")

    # ------------------------------------------------------------------ output layer (after stage 31's epilogue)
    def output_layer(self):
        """rgb[row][c] = bout[c] + sum_k Wout[c][k] h7[k][row]: weights = the X part of stage 31 (buffer 1) read as 16 k-steps,
        h7 = set Q; once per tile, not scheduled tightly"""
        e = self.e
        skip = self.label("notlast")
        e(f"s_cmp_lg_u32 {s('L')}, 7")
        e(f"s_cbranch_scc1 {skip}")
        e("s_waitcnt lgkmcnt(0)")
        for nb in range(2):
            for r in range(16):
                e(f"v_mov_b32 v{V_ACC + nb * 16 + r}, 0")
        for t0 in range(0, 16, 6):
            ts = list(range(t0, min(16, t0 + 6)))
            for t in ts:
                e(f"ds_read_b128 {aq(A_A + (t - t0) * 4)}, v{V_AX[1]} offset:{t * 1024}")
            e("s_waitcnt lgkmcnt(0)")
            for t in ts:
                for nb in range(2):
                    c = f"v[{V_ACC + nb * 16}:{V_ACC + nb * 16 + 15}]"
                    e(f"v_mfma_f32_32x32x16_bf16 {c}, {aq(A_A + (t - t0) * 4)}, {self.setreg('Q', nb, t)}, {c}")
        e(f"ds_read_b128 {vq(V_T)}, v{V_T + 8}")          # bout (V_T + 8 = its LDS address)
        e("s_nop 7")
        e("s_nop 7")
        e("s_nop 7")
        e("s_waitcnt lgkmcnt(0)")
        for nb in range(2):
            for c in range(3):
                e(f"v_add_f32 v{V_ACC + nb * 16 + c}, v{V_ACC + nb * 16 + c}, v{V_T + c}")
        for nb in range(2):      # lanes hh == 0 whose row exists store 12 bytes
            e(f"s_lshl_b32 {s('T0')}, {s('TILE')}, 8")
            e(f"s_lshl_b32 {s('T1')}, {s('WAVE')}, 6")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {s('T1')}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {32 * nb}")                   # first row of this group
            e(f"v_add_u32 v{V_T + 4}, {s('T0')}, v{V_T + 9}")                 # row = first + n
            e(f"v_cmp_gt_u32 vcc, {s('NROWS')}, v{V_T + 4}")
            e(f"s_and_b64 {s2('EX')}, vcc, 0xffffffff")                         # hh == 0: lanes 0..31
            e(f"v_mul_lo_u32 v{V_T + 5}, v{V_T + 4}, 12")
            e(f"s_mov_b64 exec, {s2('EX')}")
            e(f"global_store_dwordx3 v{V_T + 5}, v[{V_ACC + nb * 16}:{V_ACC + nb * 16 + 2}], {s2('RGB')}")
            e("s_mov_b64 exec, -1")
        e("s_barrier")      # the weights just read sit in buffer 1, which the next stage's DMA (for stage 1) overwrites
        e(f"{skip}:")

    # ------------------------------------------------------------------ the whole body
    def emit(self):
        e = self.e
        for dst, src in (("LDSB", "ldsbase"), ("WAVE", "wave"), ("TILE", "tile0"), ("GRID", "grid")):
            e(f"s_mov_b32 {s(dst)}, %[{src}]")
        if TRACE:
            e(f"s_mov_b64 {s2('TRACE')}, %[trace]")
        e(f"s_mov_b64 {s2('KARG')}, %[karg]")
        for dst, off in (("WB", "owb"), ("XT", "oxt"), ("HT", "oht"), ("MASKS", "omasks"), ("RGB", "orgb"), ("LSTR", "olstr"),
                         ("MSTR", "omstr")):
            e(f"s_load_dwordx2 {s2(dst)}, {s2('KARG')}, %[{off}]")
        e(f"s_load_dword {s('NTILES')}, {s2('KARG')}, %[ontiles]")
        e(f"s_load_dword {s('NROWS')}, {s2('KARG')}, %[onrows]")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_lshl_b64 {s2('LSTR')}, {s2('LSTR')}, 1")            # halves -> bytes
        e(f"s_lshl_b64 {s2('MSTR')}, {s2('MSTR')}, 3")            # uint64 -> bytes
        e(f"s_mov_b32 {s('ONE')}, 0x00010001")
        e(f"s_mov_b32 {s('PENDOK')}, 0")
        e(f"s_mov_b32 {s('ST')}, 0")
        # ---- lane constants
        e(f"v_mbcnt_lo_u32_b32 v{V_T}, -1, 0")
        e(f"v_mbcnt_hi_u32_b32 v{V_T}, -1, v{V_T}")
        e(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_T}")
        e(f"v_lshlrev_b32 v{V_LANE32}, 5, v{V_T}")
        e(f"v_add_u32 v{V_LANE16B}, 16384, v{V_LANE16}")
        e(f"v_lshlrev_b32 v{V_LANE4}, 2, v{V_T}")
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 10")
        e(f"v_add_u32 v{V_DMAOFF}, {s('T0')}, v{V_LANE16}")
        for b in range(2):
            e(f"s_add_u32 {s('T0')}, {s('LDSB')}, {b * STAGE}")
            e(f"v_add_u32 v{V_AX[b]}, {s('T0')}, v{V_LANE16}")
            e(f"s_add_u32 {s('T0')}, {s('T0')}, {2 * SLAB_X}")
            e(f"v_add_u32 v{V_AH[b]}, {s('T0')}, v{V_LANE16}")
        e(f"v_lshrrev_b32 v{V_T + 1}, 5, v{V_T}")                  # hh
        e(f"v_lshlrev_b32 v{V_T + 1}, 4, v{V_T + 1}")
        e(f"s_add_u32 {s('T0')}, {s('LDSB')}, {LDS_W}")
        e(f"v_add_u32 v{V_BIAS0}, {s('T0')}, v{V_T + 1}")
        e(f"s_add_u32 {s('T0')}, {s('T0')}, 8192")
        e(f"v_mov_b32 v{V_T + 8}, {s('T0')}")                      # LDS address of bout
        e(f"v_and_b32 v{V_T + 9}, 31, v{V_T}")                     # n
        # ---- first tile: stage 0 (x part) -> buffer 0, embedded rows -> bx, bias of stage 0
        done = "S2LF_DONE"
        e(f"s_cmp_lt_u32 {s('TILE')}, {s('NTILES')}")
        e(f"s_cbranch_scc0 {done}")
        e(f"s_lshl_b32 {s('T2')}, {s('WAVE')}, 12")
        e(f"s_add_u32 {s('WQ')}, {s('WB')}, {s('T2')}")
        e(f"s_addc_u32 {s('WQ1')}, {s('WB1')}, 0")
        e(f"s_add_u32 {s('MDST')}, {s('LDSB')}, {s('T2')}")
        e(f"s_mov_b32 m0, {s('MDST')}")
        e("s_nop 0")
        for o in (0, 1024, 2048, 3072):
            e(f"global_load_lds_dwordx4 v{V_LANE16}, {s2('WQ')} offset:{o}")
        e(f"s_lshl_b32 {s('G0')}, {s('TILE')}, 3")
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 1")
        e(f"s_add_u32 {s('G0')}, {s('G0')}, {s('T0')}")
        self.xt_pointer("G0")
        for k in range(16):
            nb, t = k >> 3, k & 7
            e(f"s_add_u32 {s('XTA')}, {s('XTN')}, {nb * 8192 + (t >> 1) * 2048}")
            e(f"s_addc_u32 {s('XTA1')}, {s('XTN1')}, 0")
            e(f"global_load_dwordx4 {self.setreg('X', nb, t)}, v{V_LANE16}, {s2('XTA')} offset:{(t & 1) * 1024}")
        e(f"v_mov_b32 v{V_BN}, v{V_BIAS0}")
        for text, _ in self.bias_reads(0):
            e(text)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lds = []
        e("s_barrier")
        self.first_reads(0, "x")

        e("S2LF_TILE:")
        e(f"s_mov_b32 {s('L')}, 0")
        self.layer_body("A")
        e(f"s_mov_b32 {s('L')}, 1")
        e("S2LF_B:")
        self.layer_body("B")
        e(f"s_cmp_eq_u32 {s('L')}, 7")
        e("s_cbranch_scc1 S2LF_TILE_END")
        e(f"s_add_u32 {s('L')}, {s('L')}, 1")
        e("S2LF_C:")
        self.layer_body("C")
        e(f"s_add_u32 {s('L')}, {s('L')}, 1")
        e(f"s_cmp_eq_u32 {s('L')}, 5")
        e("s_cbranch_scc0 S2LF_B")
        self.layer_body("D")
        e(f"s_mov_b32 {s('L')}, 6")
        e("s_branch S2LF_C")
        e("S2LF_TILE_END:")
        e(f"s_add_u32 {s('TILE')}, {s('TILE')}, {s('GRID')}")
        e(f"s_lshl_b32 {s('G0')}, {s('TILE')}, 3")
        e(f"s_lshl_b32 {s('T0')}, {s('WAVE')}, 1")
        e(f"s_add_u32 {s('G0')}, {s('G0')}, {s('T0')}")
        e(f"s_cmp_lt_u32 {s('TILE')}, {s('NTILES')}")
        if BX_AT_TILE_START:
            e("s_cbranch_scc0 S2LF_LAST")
            self.xt_pointer("G0")
            for k in range(16):
                nb, t = k >> 3, k & 7
                e(f"s_add_u32 {s('XTA')}, {s('XTN')}, {nb * 8192 + (t >> 1) * 2048}")
                e(f"s_addc_u32 {s('XTA1')}, {s('XTN1')}, 0")
                e(f"global_load_dwordx4 {self.setreg('X', nb, t)}, v{V_LANE16}, {s2('XTA')} offset:{(t & 1) * 1024}")
            e("s_waitcnt vmcnt(0)")
            e("s_branch S2LF_TILE")
            e("S2LF_LAST:")
        else:
            e("s_cbranch_scc1 S2LF_TILE")
        # ---- the last stage's images (blocks 6, 7 of layer 7: set Q)
        for it in self.store_items("Q", 3, guarded=False):
            for x in it:
                e(x)
        e("s_waitcnt vmcnt(0)")
        if TRACE:
            e("s_dcache_wb")
        e(f"{done}:")
        return [x for x in self.L if x is not None]

    def xt_pointer(self, g):
        """XTN = xT + 8192 * group g (bytes; 4 blocks of 2 KiB per 32-row group)"""
        e = self.e
        e(f"s_lshr_b32 {s('T1')}, {s(g)}, 19")
        e(f"s_lshl_b32 {s('T0')}, {s(g)}, 13")
        e(f"s_add_u32 {s('XTN')}, {s('XT')}, {s('T0')}")
        e(f"s_addc_u32 {s('XTN1')}, {s('XT1')}, {s('T1')}")


OPERANDS = """      :
      : [karg] "s"(karg), [ldsbase] "s"(ldsbase), [wave] "s"(wave), [tile0] "s"(tile0), [grid] "s"(grid),
        [owb] "n"(offsetof(FwdArgs, wb)), [oxt] "n"(offsetof(FwdArgs, xT)), [oht] "n"(offsetof(FwdArgs, hT)),
        [omasks] "n"(offsetof(FwdArgs, masks)), [orgb] "n"(offsetof(FwdArgs, rgb)), [olstr] "n"(offsetof(FwdArgs, layer_stride)),
        [omstr] "n"(offsetof(FwdArgs, mask_layer_stride)), [ontiles] "n"(offsetof(FwdArgs, n_tiles)),
        [onrows] "n"(offsetof(FwdArgs, n_rows))
"""


def main(outdir):
    lines = Body().emit()
    clob = ([f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)]
            + [f"s{r}" for r in range(8, S_LAST + 1) if r not in (32, 33)] + ["vcc", "scc", "memory"])
    out = ["// GENERATED by csrc/gen_fwd16_body.py -- do not edit; the generator is the source.", "asm volatile("]
    out += [f'    "{x}\\n\\t"' for x in lines]
    out.append(OPERANDS.rstrip("\n") + (', [trace] "s"(g_ftrace)' if TRACE else ""))
    out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
    with open(os.path.join(outdir, "fwd16_body.inc"), "w") as f:
        f.write("\n".join(out) + "\n")
    return len(lines)


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"fwd16_body.inc: {main(d)} instructions")
