#!/usr/bin/env python
"""Generator (and checker) of the hand-scheduled main loop of `linear4w_kernel` (csrc/linear4w.hip): the dense projections of the
ZigMa block (reference call sites mamba_simple.py:290-294, selective_scan_interface.py:365, model_zigma.py:104-135 — all F.linear)
as ONE inline-asm statement per kernel variant, for gfx950.

    python zigma_amd/csrc/gen/linear4w_gen.py --emit      # writes csrc/linear4w_body.inc (committed; the build does not run this)
    python zigma_amd/csrc/gen/linear4w_gen.py --check     # runs the generated text through the simulator below

Why generated assembly.  The arrangement: ONE wave per SIMD — a 4-wave workgroup per CU, 512 registers per
lane: 256 accumulators in AGPRs + 224 VGPRs named literally here — tile 256 tokens x 256 features, BK = 64, two 64 KB LDS stages,
direct-to-LDS loads one k-step ahead.  With a single wave per SIMD nothing covers a stall, so every non-MFMA instruction has to sit
in the ~5 issue slots between two v_mfma_f32_32x32x16_bf16 (32 cycles apart): the fragment reads of the next sub-step, the 16
global_load_lds of the k-step after next, and — in the LAST k-step of a tile, which runs block-pair-major so that accumulators
become final early — the whole epilogue of the pair before (AGPR -> LDS in fp32 -> row reads -> bf16 -> 16-byte stores).  hipcc
cannot be told to do that (it clusters reads -> wait -> MFMAs, drains vmcnt in front of visible LDS stores, and would have to keep
256 accumulators in AGPRs across a pipelined loop), so the loop is emitted as text and the counted waits (vmcnt / lgkmcnt) are
computed by the generator from its own issue log.

Because a GPU is not available while building, the SAME text is executed by the simulator in this file (4 waves, LDS, the two
memory counters, barrier intervals): it checks the result against numpy AND the synchronisation discipline — a register that a load
has not been waited for, an LDS granule read before its DMA was covered by wait + barrier, a restage before every reader passed a
barrier, an AGPR read too soon after the MFMA that wrote it.  tests/test_linear4w_gen.py runs it on the CPU suite and checks that
the committed .inc is what the generator emits.

Register plan (per lane):  a0..a255 accumulators, block (nb, mb) at (nb*4+mb)*16 — nb = feature block of the wave (4 x 32), mb =
token block (4 x 32); D[i][j]: i = feature = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), j = token = lane & 31.
v0..v31 belong to the compiler (the asm operands live there); v32.. are named here, see class RegMap.
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

# ------------------------------------------------------------------------------------------------------------------
# geometry
# ------------------------------------------------------------------------------------------------------------------
BM, BN, BK = 256, 256, 64
STAGE = 65536                     # bytes per LDS stage: rows [0, 256) W rows, [256, 512) token rows, 128 B each
SCRATCH = 2 * STAGE               # + wave * 8192: fp32 transposition scratch of a wave (32 tokens x 64 features)
LDS_BYTES = SCRATCH + 4 * 8192    # 163 840 = the CU's whole LDS


class RegMap:
    """literal registers of the asm body"""
    R0 = 32                       # fragment pool R[0:128) = v32..v159
    ROW = 160                     # 8 x 4: rows read back from the scratch (fp32)
    PK = 192                      # 4 x 4: packed bf16 rows, store data
    VOFFW = 208                   # 8: per-lane source offsets of the W-row direct-to-LDS loads
    VOFFX = 216                   # 8: ... of the token rows
    AADDR = 224                   # [stage][ks]: fragment read addresses of the W rows
    BADDR = 232                   # [stage][ks]: ... token rows
    SCRW = 240                    # 8: scratch write addresses [(blk, q)]
    SCRR = 248                    # scratch read address
    STOFF = 249                   # store offset of this lane inside a group of 8 output rows
    TMP = 250                     # 250..255 scratch for the prologue
    # SGPRs
    S_W, S_X, S_OUT = 36, 38, 40                       # 64-bit bases
    S_WP, S_XP, S_OP = 42, 43, 44                      # row pitches in bytes
    S_NK, S_TN, S_TLEFT, S_STM, S_STN = 45, 46, 47, 48, 49
    S_LMT, S_LNT, S_LKT, S_LLEFT = 50, 51, 52, 53      # load cursor
    S_LW, S_LX = 54, 56                                # 64-bit: current W / X panel pointers of the load cursor (incl. k offset)
    S_CMT, S_CNT, S_KCNT = 58, 59, 60                  # compute cursor, normal k-steps left in this tile
    S_WM, S_WN = 61, 62                                # wave's token half / feature half
    S_RS = 64                                          # 64..67 output buffer descriptor of the wave tile
    S_SO = 68                                          # 68..83: store row-group offsets [(mb, g)] = (mb*32 + g*8) * out pitch
    S_LDSW = 84                                        # wave * 1024 (+ LDS base): direct-to-LDS destination of this wave inside a row group
    S_T = 86                                           # 86..95 temporaries


RM = RegMap


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def s(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


def R(i, n=4):
    return v(RM.R0 + i, n)


def acc(nb, mb):
    return (nb * 4 + mb) * 16


# ------------------------------------------------------------------------------------------------------------------
# emitter with issue log: counted waits are derived from it
# ------------------------------------------------------------------------------------------------------------------
OPTS = set()       # probe variants (timing only, results wrong): "nomfma", "noreads", "noglds", "nostore"


class Emit:
    def __init__(self):
        self.lines = []
        self.lgkm = []          # outstanding LDS ops of this block, oldest first: sets of destination registers (or None)
        self.vm_after_glds = None   # number of vector-memory ops issued after the last direct-to-LDS load (None: no such load yet in this block)
        self.vm_own = 0         # vector-memory ops issued in this block so far
        self.n_inst = 0
        self.mfma_at = {}       # accumulator base -> instruction index of the last MFMA that wrote it
        self.in_prologue = False

    def raw(self, text):
        self.lines.append(text)

    def ins(self, text):
        if "laxvm" in OPTS and text.startswith("s_waitcnt vmcnt("):
            text = "s_waitcnt vmcnt(63)" + text[text.index(")") + 1:]
        if "noepi" in OPTS and (text.startswith("v_cvt_pk") or (text.startswith("ds_read_b128 v[1") and int(text.split("[")[1].split(":")[0]) >= RM.ROW)):
            return
        self.lines.append(text)
        self.n_inst += 1
        if text.startswith("s_nop"):
            self.n_inst += int(text.split()[1])          # wait states

    def label(self, name):
        self.lines.append(f"{name}:")

    # --- LDS ---------------------------------------------------------------------------------------------
    def ds_read(self, dst, addr_v, off):
        assert 0 <= off < 65536
        if "noreads" in OPTS and dst < RM.ROW:
            return
        self.ins(f"ds_read_b128 {v(dst, 4)}, {v(addr_v)} offset:{off}")
        self.lgkm.append(set(range(dst, dst + 4)))

    def ds_write_acc(self, addr_v, areg, off):
        assert 0 <= off < 65536
        if "noepi" in OPTS:
            return
        blk = areg // 16 * 16
        if blk in self.mfma_at:                       # wait states between the MFMA that wrote the block and this read of it
            gap = self.n_inst - self.mfma_at[blk] - 1
            if gap < 14:
                self.ins(f"s_nop {min(14 - gap, 8) - 1}")
                if 14 - gap > 8:
                    self.ins(f"s_nop {14 - gap - 8 - 1}")
        self.ins(f"ds_write_b128 {v(addr_v)}, {a(areg, 4)} offset:{off}")
        self.lgkm.append(None)

    def need(self, regs):
        """s_waitcnt lgkmcnt(k) so that every outstanding LDS read into `regs` has landed (LDS operations return in order)"""
        regs = set(regs)
        last = -1
        for i, d in enumerate(self.lgkm):
            if d is not None and d & regs:
                last = i
        if last < 0:
            return
        k = len(self.lgkm) - 1 - last
        assert k <= 15
        self.ins(f"s_waitcnt lgkmcnt({k})")
        self.lgkm = self.lgkm[last + 1:]

    def wait_lgkm0(self):
        self.ins("s_waitcnt lgkmcnt(0)")
        self.lgkm = []

    # --- vector memory ------------------------------------------------------------------------------------
    def glds(self, voff_v, sbase, m0_expr):
        self.ins(f"s_add_u32 m0, {s(RM.S_LDSW)}, {m0_expr}")
        self.pending_glds = (voff_v, sbase)

    def glds_issue(self):
        voff_v, sbase = self.pending_glds
        if "noglds" in OPTS and not self.in_prologue:
            self.vm_after_glds = 0
            return
        if self.lines[-1].startswith("s_add_u32 m0"):          # one wait state between the write of M0 and its consumer
            self.ins("s_nop 0")
        self.ins(f"global_load_lds_dwordx4 {v(voff_v)}, {s(sbase, 2)}")
        self.vm_own += 1
        self.vm_after_glds = 0

    def store(self, data, soff, imm):
        if "nostore" in OPTS:
            return
        self.ins(f"buffer_store_dwordx4 {v(data, 4)}, {v(RM.STOFF)}, {s(RM.S_RS, 4)}, {s(soff)} offen offset:{imm}")
        self.vm_own += 1
        if self.vm_after_glds is not None:
            self.vm_after_glds += 1

    def mfma(self, c, fa, fb, zero=False):
        if "nomfma" in OPTS:
            return
        src = "0" if zero else a(c, 16)
        self.ins(f"v_mfma_f32_32x32x16_bf16 {a(c, 16)}, {v(fa, 4)}, {v(fb, 4)}, {src}")
        self.mfma_at[c] = self.n_inst - 1


def interleave(e, mfmas, fillers, first_gap=0, cap=None):
    """MFMAs with the filler thunks in the gaps behind them (nothing in the first `first_gap` gaps).  cap = None: spread evenly;
    cap = n: FRONT-LOADED, n per gap until they run out (loads go first so that their latency runs under the remaining MFMAs;
    whatever is left after the last MFMA is emitted behind it)."""
    n = len(mfmas)
    gaps = max(n - first_gap, 1)
    done = 0
    for i, m in enumerate(mfmas):
        m()
        if i < first_gap:
            continue
        if cap is None:
            want = (len(fillers) * (i - first_gap + 1) + gaps - 1) // gaps
        else:
            want = min(len(fillers), done + cap)
            left_gaps = n - 1 - i
            want = max(want, len(fillers) - left_gaps * cap)      # never leave more than cap per remaining gap
        while done < want and done < len(fillers):
            fillers[done]()
            done += 1
    while done < len(fillers):
        fillers[done]()
        done += 1


# ------------------------------------------------------------------------------------------------------------------
# pieces
# ------------------------------------------------------------------------------------------------------------------
def frag_read_fillers(e, stage, ks, dst_a, dst_b, order=None):
    """the 8 fragment reads of sub-step ks from `stage`: A (W rows) blocks nb -> dst_a[nb], B (token rows) blocks mb -> dst_b[mb]"""
    ops = []
    seq = order or [("a", 0), ("b", 0), ("b", 1), ("b", 2), ("b", 3), ("a", 1), ("a", 2), ("a", 3)]
    for kind, i in seq:
        if kind == "a":
            ops.append(lambda i=i: e.ds_read(dst_a[i], RM.AADDR + stage * 4 + ks, i * 4096))
        else:
            ops.append(lambda i=i: e.ds_read(dst_b[i], RM.BADDR + stage * 4 + ks, i * 4096))
    return ops


def glds_fillers(e, stage, with_reads=()):
    """the 16 direct-to-LDS loads of one k-step into `stage` (row group q = i * 4 + wave) as a chain of 17 thunks: thunk i issues
    load i - 1 and then writes M0 for load i, so that something else (an MFMA, or the load itself) sits between a write of M0 and
    the load that consumes it; `with_reads`: LDS reads to put at the head of the first thunks (one each)"""
    reads = list(with_reads)
    ops = []
    for i in range(17):
        def th(i=i):
            if i < len(reads):
                reads[i]()
            if i >= 1:
                e.glds_issue()
            if i < 16:
                voff = RM.VOFFW + i if i < 8 else RM.VOFFX + (i - 8)
                e.glds(voff, RM.S_LW if i < 8 else RM.S_LX, hex(stage * STAGE + i * 4096))
        ops.append(th)
    assert len(reads) <= 17
    return ops


def cursor_advance(e, uid):
    """load cursor: next k-step; at the end of a tile the next tile of this workgroup (or the same one again when none is left:
    redundant loads that nobody consumes are simpler than a conditional pipeline).  The per-k-step part is six scalar instructions
    that may be spread between MFMAs; the tile change is ONE contiguous run behind a branch (a taken branch must not skip MFMAs)."""
    T = RM.S_T
    head = [
        f"s_add_u32 {s(RM.S_LW)}, {s(RM.S_LW)}, 128", f"s_addc_u32 {s(RM.S_LW + 1)}, {s(RM.S_LW + 1)}, 0",
        f"s_add_u32 {s(RM.S_LX)}, {s(RM.S_LX)}, 128", f"s_addc_u32 {s(RM.S_LX + 1)}, {s(RM.S_LX + 1)}, 0",
        f"s_sub_u32 {s(RM.S_LKT)}, {s(RM.S_LKT)}, 1",
    ]
    change = [
        f"s_cmp_eq_u32 {s(RM.S_LKT)}, 0",
        f"s_cbranch_scc0 L_noadv_{uid}_%=",
        f"s_cmp_gt_u32 {s(RM.S_LLEFT)}, 1",
        f"s_cselect_b32 {s(T + 1)}, 1, 0",                              # T1 = another tile follows
        f"s_sub_u32 {s(RM.S_LLEFT)}, {s(RM.S_LLEFT)}, {s(T + 1)}",
        f"s_mul_i32 {s(T + 2)}, {s(RM.S_STM)}, {s(T + 1)}", f"s_add_u32 {s(RM.S_LMT)}, {s(RM.S_LMT)}, {s(T + 2)}",
        f"s_mul_i32 {s(T + 2)}, {s(RM.S_STN)}, {s(T + 1)}", f"s_add_u32 {s(RM.S_LNT)}, {s(RM.S_LNT)}, {s(T + 2)}",
        f"s_cmp_ge_u32 {s(RM.S_LNT)}, {s(RM.S_TN)}",
        f"s_cselect_b32 {s(T + 2)}, {s(RM.S_TN)}, 0", f"s_cselect_b32 {s(T + 3)}, 1, 0",
        f"s_sub_u32 {s(RM.S_LNT)}, {s(RM.S_LNT)}, {s(T + 2)}", f"s_add_u32 {s(RM.S_LMT)}, {s(RM.S_LMT)}, {s(T + 3)}",
        f"s_lshl_b32 {s(T + 2)}, {s(RM.S_LNT)}, 8", f"s_mul_i32 {s(T + 2)}, {s(T + 2)}, {s(RM.S_WP)}",
        f"s_add_u32 {s(RM.S_LW)}, {s(RM.S_W)}, {s(T + 2)}", f"s_addc_u32 {s(RM.S_LW + 1)}, {s(RM.S_W + 1)}, 0",
        f"s_lshl_b32 {s(T + 2)}, {s(RM.S_LMT)}, 8", f"s_mul_i32 {s(T + 2)}, {s(T + 2)}, {s(RM.S_XP)}",
        f"s_add_u32 {s(RM.S_LX)}, {s(RM.S_X)}, {s(T + 2)}", f"s_addc_u32 {s(RM.S_LX + 1)}, {s(RM.S_X + 1)}, 0",
        f"s_mov_b32 {s(RM.S_LKT)}, {s(RM.S_NK)}",
        f"L_noadv_{uid}_%=:",
    ]

    def run():
        for t in change:
            (e.raw if t.endswith(":") else e.ins)(t)
    return [lambda t=t: e.ins(t) for t in head] + [run]


def last_LA(nbp, nbl, ks):
    return RM.R0 + nbp * 32 + nbl * 16 + ks * 4


def last_LB(mb, ks):
    return RM.R0 + 64 + mb * 16 + ks * 4


def last_entry_reads(e, stage):
    """what the LAST step needs first (feature pair 0, token block 0, all four sub-steps), into ITS register layout"""
    ops = [lambda nbl=nbl, ks=ks: e.ds_read(last_LA(0, nbl, ks), RM.AADDR + stage * 4 + ks, nbl * 4096) for ks in range(4) for nbl in range(2)]
    ops += [lambda ks=ks: e.ds_read(last_LB(0, ks), RM.BADDR + stage * 4 + ks, 0) for ks in range(4)]
    return ops


def step_normal(e, p, first, uid, vm_entry=0, next_last=False):
    """One k-step, sub-step major, consuming stage p.  Entry: the sub-step-0 fragments are in flight into buffer 0.
    first: the accumulators start here (C = 0 in sub-step 0).  vm_entry: vector-memory operations issued by the predecessor after
    ITS direct-to-LDS batch (they are younger than the batch this step waits for)."""
    buf = lambda b: ([b * 32 + nb * 4 for nb in range(4)], [b * 32 + 16 + mb * 4 for mb in range(4)])   # offsets in R
    to_v = lambda offs: [RM.R0 + o for o in offs]
    for ks in range(4):
        fa, fb = (to_v(x) for x in buf(ks & 1))
        na, nb_ = (to_v(x) for x in buf((ks + 1) & 1))
        if ks == 3:
            # boundary: my loads of the next k-step have landed, my reads of this stage are done -> barrier ->
            # the other stage may be read, this stage may be refilled
            e.ins(f"s_waitcnt vmcnt({vm_entry + e.vm_own}) lgkmcnt(0)")
            e.lgkm = []
            e.ins("s_barrier")
            # (the last-step layout lives in R[0:32) + R[64:80): clear of buffer 1 = R[32:64), which sub-step 3 is using)
            nxt = last_entry_reads(e, 1 - p) if next_last else frag_read_fillers(e, 1 - p, 0, na, nb_)
            fillers, cap = glds_fillers(e, p, with_reads=nxt), 1
        else:
            e.wait_lgkm0()
            fillers, cap = frag_read_fillers(e, p, ks + 1, na, nb_), 1
            if ks == 0:          # the load cursor moves on behind the batch the step before issued (entry contract of every step)
                fillers, cap = fillers + cursor_advance(e, uid), 3
        mf = [lambda nb=nb, mb=mb: e.mfma(acc(nb, mb), fa[nb], fb[mb], zero=(first and ks == 0)) for nb in range(4) for mb in range(4)]
        interleave(e, mf, fillers, cap=cap)


def epilogue_E1(e, nbp, mb):
    """accumulators of the block pair (2 nbp, mb), (2 nbp + 1, mb) -> scratch, fp32, straight from the AGPRs"""
    ops = []
    for b in range(2):
        for q in range(4):
            ops.append(lambda b=b, q=q: e.ds_write_acc(RM.SCRW + b * 4 + q, acc(2 * nbp + b, mb) + 4 * q, 0))
    return ops


def epilogue_E2(e):
    """scratch -> ROW: 8 tokens x 64 features per pair of reads (a lane: 8 consecutive features of one token)"""
    ops = []
    for g in range(4):
        for h in range(2):
            ops.append(lambda g=g, h=h: e.ds_read(RM.ROW + (2 * g + h) * 4, RM.SCRR, g * 2048 + h * 16))
    return ops


def epilogue_F1(e, nbp, mb):
    """ROW -> bf16 -> four 16-byte stores (8 output rows x 128 B each)"""
    ops = []
    for g in range(4):
        def cv(g=g):
            e.need(range(RM.ROW + 8 * g, RM.ROW + 8 * g + 8))
            for d in range(4):
                e.ins(f"v_cvt_pk_bf16_f32 {v(RM.PK + 4 * g + d)}, {v(RM.ROW + 8 * g + 2 * d)}, {v(RM.ROW + 8 * g + 2 * d + 1)}")
        ops.append(cv)
        ops.append(lambda g=g: e.store(RM.PK + 4 * g, RM.S_SO + mb * 4 + g, nbp * 128))
    return ops


def step_last(e, p, uid):
    """Last k-step of a tile, block-pair major (pair s: token block s >> 1, feature blocks 2 (s & 1), 2 (s & 1) + 1), with the
    epilogue of pair s - 1 / s - 2 in the gaps of pair s and the boundary (next stage ready / this stage free) before pair 2."""
    LA, LB = last_LA, last_LB
    rd_a = lambda nbp: [lambda nbl=nbl, ks=ks: e.ds_read(LA(nbp, nbl, ks), RM.AADDR + p * 4 + ks, (2 * nbp + nbl) * 4096)
                        for ks in range(4) for nbl in range(2)]
    rd_b = lambda mb: [lambda ks=ks: e.ds_read(LB(mb, ks), RM.BADDR + p * 4 + ks, mb * 4096) for ks in range(4)]
    # entry: the 12 reads of last_entry_reads() are in flight (issued by the step before)
    e.lgkm = [set(range(LA(0, nbl, ks), LA(0, nbl, ks) + 4)) for ks in range(4) for nbl in range(2)] + \
             [set(range(LB(0, ks), LB(0, ks) + 4)) for ks in range(4)]
    for sl in range(8):
        mb, nbp = sl >> 1, sl & 1
        fill = []
        if sl == 0:
            fill += rd_a(1) + rd_b(1)
        if sl == 1:
            fill += rd_b(2) + rd_b(3) + cursor_advance(e, uid)      # (pending from the batch the step before issued)
        if sl >= 2:
            fill += epilogue_F1(e, (sl - 2) & 1, (sl - 2) >> 1)
        if sl >= 1:
            fill += epilogue_E1(e, (sl - 1) & 1, (sl - 1) >> 1) + epilogue_E2(e)
        if sl == 2:
            e.ins(f"s_waitcnt vmcnt({e.vm_own}) lgkmcnt(0)")      # (nothing of mine is younger than the awaited batch yet)
            e.lgkm = []
            e.ins("s_barrier")
            fill = glds_fillers(e, p) + fill          # the whole batch FIRST: every store of this tile is younger than it
        if sl == 7:
            # sub-step-0 fragments of the next k-step (other stage) into buffer 0 = R[0:32): LA(0, ...) is free behind pair 6
            fill += frag_read_fillers(e, 1 - p, 0, [RM.R0 + nb * 4 for nb in range(4)], [RM.R0 + 16 + m * 4 for m in range(4)])
        # this pair's fragments
        need = set()
        for ks in range(4):
            need |= set(range(LB(mb, ks), LB(mb, ks) + 4))
            for nbl in range(2):
                need |= set(range(LA(nbp, nbl, ks), LA(nbp, nbl, ks) + 4))
        e.need(need)
        mf = []
        for ks in range(4):
            for nbl in range(2):
                mf.append(lambda ks=ks, nbl=nbl: e.mfma(acc(2 * nbp + nbl, mb), LA(nbp, nbl, ks), LB(mb, ks)))
        interleave(e, mf, fill, first_gap=1 if sl >= 1 else 0)
    for f in epilogue_F1(e, 0, 3) + epilogue_E1(e, 1, 3) + epilogue_E2(e) + epilogue_F1(e, 1, 3):
        f()
    return e.vm_after_glds


def tile_advance(e, uid):
    """compute cursor -> next tile of this workgroup, output descriptor of the wave tile"""
    T = RM.S_T
    L = [
        f"s_add_u32 {s(RM.S_CMT)}, {s(RM.S_CMT)}, {s(RM.S_STM)}",
        f"s_add_u32 {s(RM.S_CNT)}, {s(RM.S_CNT)}, {s(RM.S_STN)}",
        f"s_cmp_ge_u32 {s(RM.S_CNT)}, {s(RM.S_TN)}",
        f"s_cbranch_scc0 L_cnw_{uid}_%=",
        f"s_sub_u32 {s(RM.S_CNT)}, {s(RM.S_CNT)}, {s(RM.S_TN)}",
        f"s_add_u32 {s(RM.S_CMT)}, {s(RM.S_CMT)}, 1",
        f"L_cnw_{uid}_%=:",
    ]
    return L + out_descriptor()


def out_descriptor():
    T = RM.S_T
    return [
        # byte offset of the wave tile: ((c_mt * 2 + wm) * 128) * out pitch + ((c_nt * 2 + wn) * 128) * 2
        f"s_lshl_b32 {s(T)}, {s(RM.S_CMT)}, 1", f"s_add_u32 {s(T)}, {s(T)}, {s(RM.S_WM)}", f"s_lshl_b32 {s(T)}, {s(T)}, 7",
        f"s_mul_i32 {s(T)}, {s(T)}, {s(RM.S_OP)}",
        f"s_lshl_b32 {s(T + 1)}, {s(RM.S_CNT)}, 1", f"s_add_u32 {s(T + 1)}, {s(T + 1)}, {s(RM.S_WN)}", f"s_lshl_b32 {s(T + 1)}, {s(T + 1)}, 8",
        f"s_add_u32 {s(T)}, {s(T)}, {s(T + 1)}",
        f"s_add_u32 {s(RM.S_RS)}, {s(RM.S_OUT)}, {s(T)}", f"s_addc_u32 {s(RM.S_RS + 1)}, {s(RM.S_OUT + 1)}, 0",
        f"s_and_b32 {s(RM.S_RS + 1)}, {s(RM.S_RS + 1)}, 0xffff",
        f"s_mov_b32 {s(RM.S_RS + 2)}, 0x7ffffffe", f"s_mov_b32 {s(RM.S_RS + 3)}, 0x00020000",
    ]


# operands of the asm statement: (constraint, C expression) in order; the body refers to them as %0 ...
OPERANDS = [
    ("v", "voffw0"), ("v", "voffx0"), ("v", "a_base"), ("v", "b_base"), ("v", "t_xor"), ("v", "scrw_base"), ("v", "j7"),
    ("v", "scrr"), ("v", "stoff"),
    ("s", "w_ptr"), ("s", "x_ptr"), ("s", "out_ptr"), ("s", "w_pitch"), ("s", "x_pitch"), ("s", "o_pitch"), ("s", "nk"),
    ("s", "tiles_n"), ("s", "my_tiles"), ("s", "step_m"), ("s", "step_n"), ("s", "mt0"), ("s", "nt0"), ("s", "wave"), ("s", "lds_base"),
]
OP = {name: i for i, (_, name) in enumerate(OPERANDS)}


def prologue(e):
    e.in_prologue = True
    o = lambda name: f"%{OP[name]}"
    T, TV = RM.S_T, RM.TMP
    L = []
    # scalar state
    for dst, src in ((RM.S_W, "w_ptr"), (RM.S_X, "x_ptr"), (RM.S_OUT, "out_ptr")):
        L.append(f"s_mov_b64 {s(dst, 2)}, {o(src)}")
    for dst, src in ((RM.S_WP, "w_pitch"), (RM.S_XP, "x_pitch"), (RM.S_OP, "o_pitch"), (RM.S_NK, "nk"), (RM.S_TN, "tiles_n"),
                     (RM.S_TLEFT, "my_tiles"), (RM.S_STM, "step_m"), (RM.S_STN, "step_n"), (RM.S_LMT, "mt0"), (RM.S_LNT, "nt0"),
                     (RM.S_CMT, "mt0"), (RM.S_CNT, "nt0")):
        L.append(f"s_mov_b32 {s(dst)}, {o(src)}")
    L += [f"s_mov_b32 {s(RM.S_LLEFT)}, {s(RM.S_TLEFT)}", f"s_mov_b32 {s(RM.S_LKT)}, {s(RM.S_NK)}",
          f"s_and_b32 {s(RM.S_WN)}, {o('wave')}, 1", f"s_lshr_b32 {s(RM.S_WM)}, {o('wave')}, 1",
          f"s_lshl_b32 {s(RM.S_LDSW)}, {o('wave')}, 10", f"s_add_u32 {s(RM.S_LDSW)}, {s(RM.S_LDSW)}, {o('lds_base')}"]
    for mb in range(4):
        for g in range(4):
            L += [f"s_mul_i32 {s(RM.S_SO + mb * 4 + g)}, {s(RM.S_OP)}, {mb * 32 + g * 8}"]
    # load cursor pointers of the first tile
    L += [f"s_lshl_b32 {s(T)}, {s(RM.S_LNT)}, 8", f"s_mul_i32 {s(T)}, {s(T)}, {s(RM.S_WP)}",
          f"s_add_u32 {s(RM.S_LW)}, {s(RM.S_W)}, {s(T)}", f"s_addc_u32 {s(RM.S_LW + 1)}, {s(RM.S_W + 1)}, 0",
          f"s_lshl_b32 {s(T)}, {s(RM.S_LMT)}, 8", f"s_mul_i32 {s(T)}, {s(T)}, {s(RM.S_XP)}",
          f"s_add_u32 {s(RM.S_LX)}, {s(RM.S_X)}, {s(T)}", f"s_addc_u32 {s(RM.S_LX + 1)}, {s(RM.S_X + 1)}, 0"]
    L += out_descriptor()
    # per-lane tables
    for i in range(8):
        L += [f"s_mul_i32 {s(T)}, {s(RM.S_WP)}, {i * 32}", f"v_add_u32 {v(RM.VOFFW + i)}, {s(T)}, {o('voffw0')}",
              f"s_mul_i32 {s(T)}, {s(RM.S_XP)}, {i * 32}", f"v_add_u32 {v(RM.VOFFX + i)}, {s(T)}, {o('voffx0')}"]
    for ks in range(4):
        L += [f"v_xor_b32 {v(TV)}, {ks * 2}, {o('t_xor')}", f"v_lshlrev_b32 {v(TV)}, 4, {v(TV)}",
              f"v_add_u32 {v(RM.AADDR + ks)}, {v(TV)}, {o('a_base')}", f"v_add_u32 {v(RM.BADDR + ks)}, {v(TV)}, {o('b_base')}",
              f"v_add_u32 {v(RM.AADDR + 4 + ks)}, {hex(STAGE)}, {v(RM.AADDR + ks)}", f"v_add_u32 {v(RM.BADDR + 4 + ks)}, {hex(STAGE)}, {v(RM.BADDR + ks)}"]
    for bq in range(8):
        L += [f"v_xor_b32 {v(TV)}, {bq}, {o('j7')}", f"v_lshlrev_b32 {v(TV)}, 5, {v(TV)}", f"v_add_u32 {v(RM.SCRW + bq)}, {v(TV)}, {o('scrw_base')}"]
    L += [f"v_mov_b32 {v(RM.SCRR)}, {o('scrr')}", f"v_mov_b32 {v(RM.STOFF)}, {o('stoff')}"]
    for t in L:
        e.ins(t)
    # k-steps 0 and 1 of the first tile in flight, the first one waited for, its sub-step-0 fragments requested
    for st in range(2):
        for f in glds_fillers(e, st) + (cursor_advance(e, f"pro{st}") if st == 0 else []):
            f()
        # (back to back in the prologue: the chain puts the write of M0 right in front of its load)
    e.ins("s_waitcnt vmcnt(16)")
    e.ins("s_barrier")
    for f in frag_read_fillers(e, 0, 0, [RM.R0 + nb * 4 for nb in range(4)], [RM.R0 + 16 + m * 4 for m in range(4)]):
        f()


def clobbers():
    c = ["memory", "scc", "vcc"]
    c += [f"v{i}" for i in range(32, 256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(36, 96)]
    return c


def generate():
    """-> (asm text with %N operands and %= label ids, T = vector-memory ops a LAST step issues behind its direct-to-LDS batch)"""
    out = []

    def block(fn, *args, **kw):
        e = Emit()
        r = fn(e, *args, **kw)
        out.extend(e.lines)
        return r

    # T: fixed point of "what a LAST step leaves behind its batch" (does not depend on the wait counts)
    probe = Emit()
    T_after = step_last(probe, 0, "probe")
    block(prologue)
    # control flow (nk >= 3): FIRST -> NORMAL x (nk - 3) -> NORMAL-before-last -> LAST, stage parity alternating; LAST -> FIRST
    out.append("s_branch L_first0_%=")
    for p in range(2):
        q = 1 - p
        variants = [(f"L_first_{p}", True, T_after, False), (f"L_norm_{p}", False, 0, False), (f"L_norml_{p}", False, 0, True)]
        if p == 0:
            variants.append(("L_first0", True, 0, False))
        for name, first, vm_entry, next_last in variants:
            out.append(f"{name}_%=:")
            block(step_normal, p, first, name[2:], vm_entry=vm_entry, next_last=next_last)
            if next_last:
                out += [f"s_branch L_last_{q}_%="]
                continue
            if first:
                out += [f"s_sub_u32 {s(RM.S_KCNT)}, {s(RM.S_NK)}, 2"]
            else:
                out += [f"s_sub_u32 {s(RM.S_KCNT)}, {s(RM.S_KCNT)}, 1"]
            out += [f"s_cmp_eq_u32 {s(RM.S_KCNT)}, 1", f"s_cbranch_scc1 L_norml_{q}_%=", f"s_branch L_norm_{q}_%="]
        out.append(f"L_last_{p}_%=:")
        t_here = block(step_last, p, f"last{p}")
        assert t_here == T_after
        out += [f"s_sub_u32 {s(RM.S_TLEFT)}, {s(RM.S_TLEFT)}, 1", f"s_cmp_eq_u32 {s(RM.S_TLEFT)}, 0", f"s_cbranch_scc1 L_end_%="]
        out += tile_advance(None, f"ta{p}")
        out += [f"s_branch L_first_{q}_%="]
    out += ["L_end_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return out, T_after


PROBE_VARIANTS = {"NOMFMA": {"nomfma"}, "LOADS": {"nomfma", "noreads", "nostore"}, "NOGLDS": {"noglds"}, "NOSTORE": {"nostore"},
                  "MFMAONLY": {"noglds", "nostore"},
                  "NOGLDS_LAX": {"noglds", "laxvm"}, "MFMA_NOEPI": {"noglds", "nostore", "noepi"}}


def emit_inc(path):
    global OPTS
    with open(path, "w") as fh:
        fh.write("// GENERATED by zigma_amd/csrc/gen/linear4w_gen.py --emit — do not edit (tests/test_linear4w_gen.py checks it is current).\n")
        fh.write("// The main loop of linear4w_kernel as one asm statement; operands in the order of OPERANDS in the generator.\n")

        def body(name, opts):
            global OPTS
            OPTS = set(opts)
            lines, _ = generate()
            OPTS = set()
            fh.write(f"#define ZIGMA_LINEAR4W_BODY{name} \\\n")
            for ln in lines:
                fh.write(f'    "{ln}\\n" \\\n')
            fh.write("    \"\"\n")
        body("", ())
        fh.write("#ifdef ZIGMA_LINEAR4W_PROBES   // timing probes (tools/linear4w_probe.py builds its own library with them): results are wrong\n")
        for name, opts in PROBE_VARIANTS.items():
            body("_" + name, opts)
        fh.write("#endif\n")
        fh.write("#define ZIGMA_LINEAR4W_OPERANDS(" + ", ".join(n for _, n in OPERANDS) + ") \\\n    " +
                 ", ".join(f'"{c}"({n})' for c, n in OPERANDS) + "\n")
        cl = clobbers()
        fh.write("#define ZIGMA_LINEAR4W_CLOBBERS \\\n")
        for i in range(0, len(cl), 16):
            fh.write("    " + ", ".join(f'"{c}"' for c in cl[i:i + 16]) + (", \\\n" if i + 16 < len(cl) else "\n"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--emit", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stats", action="store_true")
    args = ap.parse_args()
    if args.emit:
        emit_inc(os.path.join(os.path.dirname(HERE), "linear4w_body.inc"))
    if args.stats:
        lines, T = generate()
        n = sum(1 for l in lines if not l.endswith(":"))
        print(f"{n} instructions, {sum('v_mfma' in l for l in lines)} MFMAs, T = {T}")
    if args.check:
        sys.path.insert(0, HERE)
        import linear4w_sim
        linear4w_sim.main()
