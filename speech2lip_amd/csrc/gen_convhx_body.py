"""Generates convhx_body.inc / convhxr_body.inc: the half-width 3x3 convolution (gen_convh_body.py has the arithmetic, gen_convh8_body.py the
eight-wave tile: both unchanged, the outputs are the same bits) with the two waves of every SIMD in ALTERNATING ROLES
(MI355X_MICROARCH.md, "Two waves per SIMD"): while one wave runs an MFMA-only segment -- the 24 MFMAs of three taps, on operands it already
holds in registers -- its partner runs the segment that prepares ITS next 24: 24 operand reads from LDS, its share of the next chunk's
LDS-DMA requests, and at a tile boundary the whole epilogue of the tile it just finished (accumulators -> bf16 -> LDS transpose -> stores)
plus the next tile's bias reads.  s_barrier between segments; waves 0-3 (one per SIMD) lead, waves 4-7 follow one segment behind.

Why (tools/ubench/gen_mfma_segments.py, profiles/r05_mfma_segments.txt): with the convolution's LDS traffic and nothing else, both forms keep
the matrix pipe > 0.9 busy (interleaved 0.92, alternating 0.995) -- the eight-wave kernel's 0.53 is not lost between reads and MFMAs but in
what NO wave covers: all eight waves reach the epilogue, the barrier behind it, the bias reads and the first tap's exposed reads together.
Here every such phase of one wave sits beside 768 cycles of its partner's MFMAs.

Per chunk (one 32-channel plane, buffer p) and per SIMD, six barrier-separated intervals:
    leader    L0 + halo DMA | C0 | L1 + weight DMA | C1 | L2 | C2, vmcnt(0)
    follower  C2'| L0 + halo DMA | C0 | L1 + weight DMA | C1 | L2, vmcnt(0)            (C2' = the previous chunk's last segment)
Lk reads the operands of taps 3k .. 3k+2 (96 AGPRs, no double buffering: a wave never loads while it computes); the next chunk's halo
requests (HBM / L2 latency) leave in a group's FIRST load interval -- buffer p ^ 1's halo area is free from the chunk's first barrier on --
its weight requests (L2-hot) in the second: at a tile's first chunk the first two intervals hold the two groups' epilogues, whose store
staging is buffer 1's WEIGHT area (csrc/convh.hip: kCHStageX); every wave drains vmcnt before the barrier that ends the chunk's last interval.  Barriers: leader 6 per chunk + 1 after
its last epilogue, follower 1 before its first chunk + 6 per chunk -- the same count, pairwise aligned.

Register map: gen_convh8_body.py's without the gate pieces (the launcher sends gated launches -- the eval-mode chain's input gradients -- to the
interleaved kernel): a0-63 accumulators, a64-159 three operand sets, v0-87."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_convh8_body as G

NB = G.NB
OPS = (64, 96, 128)
A_LAST, V_LAST = 159, G.V_G - 1
s, s2, S = G.s, G.s2, G.S
WITH_RELU = False
DMA_IN_COMPUTE = os.environ.get("S2L_CHX_DMA", "compute") == "compute"      # where a wave issues its LDS-DMA requests: among its MFMAs | among its reads


class BodyX(G.Body):
    pfx = "S2LX"

    @staticmethod
    def opa(os_, ks, mb):
        return OPS[os_] + (ks * 2 + mb) * 4

    @staticmethod
    def opb(os_, ks, nb):
        return OPS[os_] + 16 + (ks * NB + nb) * 4

    # ---- segments
    def throttle(self):
        while len(self.lds) >= 14:                                    # (lgkmcnt counts to 15)
            self.wait_lds(self.lds[0])

    def emit_group_throttled(self, g):
        for it in g:
            if isinstance(it, tuple) and it[0] != "wait":
                self.throttle()
        self.emit_group(g)

    def load_seg(self, k, p, extra):
        """operand reads of taps 3k .. 3k+2 from buffer p into the three operand sets, `extra` (instruction groups: DMA requests and the
        zeroes of border slots) spread between them; ends with every LDS operation of this wave complete"""
        reads = []
        for t in range(3 * k, 3 * k + 3):
            reads += self.tap_reads(t, p, t % 3)
        every = max(1, len(reads) // (len(extra) + 1)) if extra else 0
        x = 0
        for i, (text, tag) in enumerate(reads):
            self.throttle()
            self.lds_op(text, tag)
            if extra and (i + 1) % every == 0 and x < len(extra):
                self.emit_group_throttled(extra[x])
                x += 1
        while x < len(extra):
            self.emit_group_throttled(extra[x])
            x += 1
        self.wait_all_lds()

    def compute_seg(self, extra=()):
        """the 24 MFMAs of three taps; `extra` (instruction groups: this wave's DMA requests) behind every fourth one"""
        extra = list(extra)
        if G.EXP & 2048:                                             # (ablation: no MFMAs)
            for g in extra:
                self.emit_group(g)
            self.wait_all_lds()
            return
        for k in range(3):
            sprinkle = [[] for _ in range(8)]
            for m in (2, 6):
                if extra:
                    sprinkle[m] = extra.pop(0)
            self.tap_mfmas(k, sprinkle)
        assert not extra
        self.wait_all_lds()                                           # (the zeroes of border slots)

    def store_tile_nogate(self):
        """gen_convh8_body.py's store_tile without the gate: accumulators -> bf16 -> this wave's 4 KiB of LDS staging -> whole 1-KiB rows"""
        e = self.e
        ta, tc, tb = G.V_T, G.V_T + 16, G.V_T
        for nb in range(NB):
            for mb in range(2):
                a0 = G.A_ACC + (mb * NB + nb) * 16
                for rq in range(4):
                    for j in range(4):
                        e(f"v_accvgpr_read_b32 v{ta + 4 * rq + j}, a{a0 + 4 * rq + j}")
                    if WITH_RELU:
                        for j in range(4):
                            e(f"v_max_f32 v{ta + 4 * rq + j}, {s('RELU')}, v{ta + 4 * rq + j}")
                    e(f"v_cvt_pk_bf16_f32 v{tc + 2 * rq}, v{ta + 4 * rq}, v{ta + 4 * rq + 1}")
                    e(f"v_cvt_pk_bf16_f32 v{tc + 2 * rq + 1}, v{ta + 4 * rq + 2}, v{ta + 4 * rq + 3}")
                    self.lds_op(f"ds_write_b64 v{G.V_SWA + mb * 4 + rq}, v[{tc + 2 * rq}:{tc + 2 * rq + 1}]", ("SW", nb, mb, rq))
            for j in range(4):
                self.lds_op(f"ds_read_b128 v[{tb + 4 * j}:{tb + 4 * j + 3}], v{G.V_SRA + j}", ("SR", nb, j))
            self.wait_all_lds()
            for j in range(4):
                self.row_exec(2 * nb + (j >> 1), j & 1, "OUTF", nb == 0 and j == 0)
                if not G.EXP & 1:
                    e(f"global_store_dwordx4 v{G.V_VS}, v[{tb + 4 * j}:{tb + 4 * j + 3}], {s2('ROWB')}")
            e("s_mov_b64 exec, -1")

    def epilogue(self):
        e = self.e
        e("s_nop 7")
        e("s_nop 7")                                                  # (MFMA results -> v_accvgpr_read)
        if not G.EXP & 512:
            self.store_tile_nogate()
        self.next_coords("", "CT")

    # ---- the two streams' chunk slots
    def leader_chunk(self, p):
        e = self.e
        if DMA_IN_COMPUTE:
            self.load_seg(0, p, [])
            e("s_barrier")
            self.staging_source()
            self.compute_seg(self.halo_items(p ^ 1))                  # (a request costs ~60 cycles of issue among MFMAs, 100-185 among reads)
            e("s_barrier")
            self.load_seg(1, p, [])
            e("s_barrier")
            self.compute_seg(self.dma_items(p ^ 1))                   # (weights: after both groups' epilogues, whose staging is buffer 1's weight area)
            self.advance_staging()
            e("s_barrier")
            self.load_seg(2, p, [])
            e("s_barrier")
            self.compute_seg()
            e("s_waitcnt vmcnt(0)")
            e("s_barrier")
            return
        self.staging_source()
        self.load_seg(0, p, self.halo_items(p ^ 1))                   # (the long-latency requests first: five intervals to land)
        e("s_barrier")
        self.compute_seg()
        e("s_barrier")
        self.load_seg(1, p, self.dma_items(p ^ 1))                    # (weights: after both groups' epilogues, whose staging is buffer 1's weight area)
        self.advance_staging()
        e("s_barrier")
        self.compute_seg()
        e("s_barrier")
        self.load_seg(2, p, [])
        e("s_barrier")
        self.compute_seg()
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")

    def follower_chunk(self, p):
        e = self.e
        if DMA_IN_COMPUTE:
            self.load_seg(0, p, [])
            e("s_barrier")
            self.staging_source()
            self.compute_seg(self.halo_items(p ^ 1))
            e("s_barrier")
            self.load_seg(1, p, [])
            e("s_barrier")
            self.compute_seg(self.dma_items(p ^ 1))
            self.advance_staging()
            e("s_barrier")
            self.load_seg(2, p, [])
            e("s_waitcnt vmcnt(0)")
            e("s_barrier")
            self.compute_seg()
            e("s_barrier")
            return
        self.staging_source()
        self.load_seg(0, p, self.halo_items(p ^ 1))
        e("s_barrier")
        self.compute_seg()
        e("s_barrier")
        self.load_seg(1, p, self.dma_items(p ^ 1))
        self.advance_staging()
        e("s_barrier")
        self.compute_seg()
        e("s_barrier")
        self.load_seg(2, p, [])
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        self.compute_seg()                                            # (beside the leader's L0 of the next chunk)
        e("s_barrier")


def generate():
    b = BodyX()
    e = b.e
    pfx = b.pfx
    G.emit_prologue(b, with_gate=False)
    # this form's store staging is buffer 1's WEIGHT area: the weights never stay resident across tiles, every chunk reloads them
    # (a 3000-shape soak caught the eight-wave form's "two chunks, one channel tile" shortcut reaching this body through dma_items)
    e(f"s_mov_b32 {s('WLD')}, 0x7fffffff")
    e(f"s_cmp_gt_u32 {s('WAVE')}, 3")
    e(f"s_cbranch_scc1 {pfx}_FOLLOW")
    # ================= leaders (waves 0-3)
    e(f"{pfx}_LTILE:")
    b.tile_begin()
    b.bias_init()
    e(f"s_mov_b32 {s('CC')}, 0")
    e(f"{pfx}_LPAIR:")
    b.leader_chunk(0)
    b.leader_chunk(1)
    e(f"s_add_u32 {s('CC')}, {s('CC')}, 2")
    e(f"s_cmp_lt_u32 {s('CC')}, {s('NCH')}")
    e(f"s_cbranch_scc1 {pfx}_LPAIR")
    b.epilogue()
    e(f"s_sub_u32 {s('NTL')}, {s('NTL')}, 1")
    e(f"s_cmp_gt_u32 {s('NTL')}, 0")
    e(f"s_cbranch_scc1 {pfx}_LTILE")
    e("s_barrier")                                                    # (pairs with the followers' last one)
    e(f"s_branch {pfx}_END")
    # ================= followers (waves 4-7), one interval behind
    e(f"{pfx}_FOLLOW:")
    e("s_barrier")                                                    # (the leaders' first L0)
    e(f"{pfx}_FTILE:")
    b.tile_begin()
    b.bias_init()
    e(f"s_mov_b32 {s('CC')}, 0")
    e(f"{pfx}_FPAIR:")
    b.follower_chunk(0)
    b.follower_chunk(1)
    e(f"s_add_u32 {s('CC')}, {s('CC')}, 2")
    e(f"s_cmp_lt_u32 {s('CC')}, {s('NCH')}")
    e(f"s_cbranch_scc1 {pfx}_FPAIR")
    b.epilogue()
    e(f"s_sub_u32 {s('NTL')}, {s('NTL')}, 1")
    e(f"s_cmp_gt_u32 {s('NTL')}, 0")
    e(f"s_cbranch_scc1 {pfx}_FTILE")
    e(f"{pfx}_END:")
    e("s_waitcnt vmcnt(0)")
    return b.L


def main(objdir):
    global WITH_RELU
    n = 0
    for relu, pfx, name in ((False, "S2LX", "convhx_body.inc"), (True, "S2LY", "convhxr_body.inc")):
        WITH_RELU = relu
        BodyX.pfx = pfx
        lines = generate()
        clob = [f"v{r}" for r in range(0, V_LAST + 1)] + [f"a{r}" for r in range(0, A_LAST + 1)] + [f"s{r}" for r in range(16, G.S_LAST + 1) if r not in (32, 33)]
        clob += ["vcc", "scc", "memory"]
        out = ["// GENERATED by csrc/gen_convhx_body.py -- do not edit; the generator is the source.", "asm volatile("]
        out += [f'    "{x}\\n\\t"' for x in lines]
        out.append(G.OPERANDS.rstrip("\n"))
        out.append("      : " + ", ".join(f'"{c}"' for c in clob) + ");")
        with open(os.path.join(objdir, name), "w") as f:
            f.write("\n".join(out) + "\n")
        n = len(lines)
    WITH_RELU, BodyX.pfx = False, "S2LX"
    return n


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build")
    print(f"convhx body: {main(d)} instructions")
