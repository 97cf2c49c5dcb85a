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
    R0 = 32                       # fragment pool R[0:128) = v32..v159 (normal steps use R[0:64); R[64:128) holds per-tile temporaries then)
    ROW = 160                     # 2 x 8: one group of 8 output rows read back from the scratch (fp32), double-buffered
    RES = 176                     # 2 x 4: the residual rows of that group (gated-residual epilogue)
    PK = 184                      # 2 x 4: packed bf16 rows, store data
    G = 192                       # 16: gate of this wave tile's sample as fp32, [feature pair][8 features of the lane]
    VOFFW = 208                   # 8: per-lane source offsets of the W-row direct-to-LDS loads (wide tile)
    VOFFX = 216                   # 8: ... of the token rows
    AADDR = 224                   # [stage][ks]: fragment read addresses of the W rows
    BADDR = 232                   # [stage][ks]: ... token rows
    SCRW = 240                    # 8: scratch write addresses [(blk, q)]
    SCRR = 248                    # scratch read address
    STOFF = 249                   # store offset of this lane inside a group of 8 output rows (residual rows: the same)
    VOFFWE = 250                  # 4: W-row offsets of loads 4..7 for the tile the load cursor is on (narrow tiles re-map them)
    TMP = 254                     # 254, 255 scratch
    # temporaries of a tile's first steps, inside R[96:128) (the step before the last one prefetches into R[0:32) and R[64:80))
    BIASA = R0 + 96               # 4 x 4: A fragments [bias, 0, 0, 0] of the rank-1 bias product
    ONES = R0 + 112               # 4: B fragment [1.0 in k = 0, 0 ...]
    GRAW = R0 + 116               # 2 x 4: the gate rows as loaded (bf16)
    GOFF = R0 + 124               # (lane & 7) * 16
    # SGPRs (mutable state; read-only inputs are used as operands where they arrive)
    S_W, S_X, S_OUT = 36, 38, 40                       # 64-bit bases
    S_WP, S_XP, S_OP = 42, 43, 44                      # row pitches in bytes
    S_NK, S_TN, S_TLEFT, S_STM, S_STN = 45, 46, 47, 48, 49
    S_LMT, S_LNT, S_LKT, S_LLEFT = 50, 51, 52, 53      # load cursor
    S_LW, S_LX = 54, 56                                # 64-bit: current W / X panel pointers of the load cursor (incl. k offset)
    S_CMT, S_CNT, S_KCNT = 58, 59, 60                  # compute cursor, normal k-steps left in this tile
    S_WM, S_WN = 61, 62                                # wave's token half / feature half
    S_CNARROW = 63                                     # the compute tile is a narrow one (128 features)
    S_RS = 64                                          # 64..67 output buffer descriptor of the wave tile
    S_SO = 68                                          # 68..83: store row-group offsets [(mb, g)] = (mb*32 + g*8) * out pitch
    S_LDSW = 84                                        # wave * 1024 (+ LDS base): direct-to-LDS destination of this wave inside a row group
    S_NWIDE = 85                                       # number of 256-wide n-tiles (a narrow tile, if any, has index S_NWIDE)
    S_T = 86                                           # 86..91 temporaries
    S_RRS = 92                                         # 92..95 residual buffer descriptor of the wave tile
    S_BRS = 96                                         # 96..99 bias descriptor of the wave tile
    S_GP = 100                                         # 100..101 gate row pointer of the wave tile


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
# cache policy of the output stores: nt (streaming).  Measured inside the forward (tools/fwd_l4w_nt_ab.sh, four interleaved rounds on one
# box): 18.04 / 18.09 / 18.13 / 18.15 ms with the default policy against 17.98 / 18.04 / 18.07 / 18.11 with nt (-0.3 %, every round).  The same
# switch on the attention core and the norm kernels: +-0 together with this one; on the scan's output +-0, on conv + x_proj's +0.8 %: not
# adopted there.  L4W_STORE_NT=0 regenerates the default-policy body for that A/B.
STORE_POLICY = " nt" if os.environ.get("L4W_STORE_NT", "1") == "1" else ""
RES_AHEAD = os.environ.get("L4W_RES_AHEAD", "1") == "1"      # residual rows of the next block pair requested in front of this pair's stores
OPTS = set()       # probe variants (timing only, results wrong): "nomfma", "noreads", "noglds", "nostore", "laxvm", "noepi"
CFG = dict(narrow=False, res=False, bias=False)     # what the body being generated supports (see VARIANTS)


class Emit:
    def __init__(self):
        self.lines = []
        self.lgkm = []          # outstanding LDS ops of this block, oldest first: sets of destination registers (or None)
        self.vmq = []           # outstanding vector-memory ops of this block, oldest first: destination register sets (or None)
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

    def need_vm(self, regs):
        """the same for vector loads of this block (the counter retires in issue order, loads and stores alike)"""
        regs = set(regs)
        last = -1
        for i, d in enumerate(self.vmq):
            if d is not None and d & regs:
                last = i
        if last < 0:
            return
        k = len(self.vmq) - 1 - last
        assert k <= 63
        self.ins(f"s_waitcnt vmcnt({k})")
        self.vmq = self.vmq[last + 1:]

    def wait_lgkm0(self):
        self.ins("s_waitcnt lgkmcnt(0)")
        self.lgkm = []

    # --- vector memory ------------------------------------------------------------------------------------
    def _vm(self, dst=None):
        self.vm_own += 1
        self.vmq.append(dst)
        if self.vm_after_glds is not None:
            self.vm_after_glds += 1

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
        self._vm()
        self.vm_after_glds = 0

    def store(self, data, soff, imm):
        if "nostore" in OPTS:
            return
        self.ins(f"buffer_store_dwordx4 {v(data, 4)}, {v(RM.STOFF)}, {s(RM.S_RS, 4)}, {s(soff)} offen offset:{imm}{STORE_POLICY}")
        self._vm()

    def res_load(self, dst, soff, imm):
        self.ins(f"buffer_load_dwordx4 {v(dst, 4)}, {v(RM.STOFF)}, {s(RM.S_RRS, 4)}, {s(soff)} offen offset:{imm}")
        self._vm(set(range(dst, dst + 4)))

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


def o(name):
    return f"%{OP[name]}"


# ------------------------------------------------------------------------------------------------------------------
# pieces
# ------------------------------------------------------------------------------------------------------------------
def frag_read_fillers(e, stage, ks, dst_a, dst_b, nbw=4):
    """the fragment reads of sub-step ks from `stage`: A (W rows) blocks nb < nbw -> dst_a[nb], B (token rows) blocks mb -> dst_b[mb]"""
    seq = [("a", 0), ("b", 0), ("b", 1), ("b", 2), ("b", 3)] + [("a", i) for i in range(1, nbw)]
    ops = []
    for kind, i in seq:
        if kind == "a":
            ops.append(lambda i=i: e.ds_read(dst_a[i], RM.AADDR + stage * 4 + ks, i * 4096))
        else:
            ops.append(lambda i=i: e.ds_read(dst_b[i], RM.BADDR + stage * 4 + ks, i * 4096))
    return ops


def glds_fillers(e, stage, with_reads=()):
    """the 16 direct-to-LDS loads of one k-step into `stage` (row group q = i * 4 + wave) as a chain of 17 thunks: thunk i issues
    load i - 1 and then writes M0 for load i, so that something else (an MFMA, or the load itself) sits between a write of M0 and
    the load that consumes it; `with_reads`: LDS reads to put at the head of the first thunks (one each).
    W-row loads 4..7 take their offsets from VOFFWE: on a narrow tile (128 features; wave half wn reads LDS rows wn * 128 + [0, 64))
    they fetch features 64..127 (and, harmlessly, once more) instead of rows that a last n-tile does not have."""
    reads = list(with_reads)
    ops = []
    for i in range(17):
        def th(i=i):
            if i < len(reads):
                reads[i]()
            if i >= 1:
                e.glds_issue()
            if i < 16:
                voff = (RM.VOFFW + i if i < 4 else RM.VOFFWE + (i - 4)) if i < 8 else RM.VOFFX + (i - 8)
                e.glds(voff, RM.S_LW if i < 8 else RM.S_LX, hex(stage * STAGE + i * 4096))
        ops.append(th)
    assert len(reads) <= 17
    return ops


def voffwe_update():
    """VOFFWE for the tile the load cursor is on (contiguous code, not to be spread between MFMAs)"""
    L = []
    if CFG["narrow"]:
        L += [f"s_cmp_eq_u32 {s(RM.S_LNT)}, {s(RM.S_NWIDE)}", f"s_cbranch_scc1 L_vwn_@_%="]
    L += [f"v_mov_b32 {v(RM.VOFFWE + k)}, {v(RM.VOFFW + 4 + k)}" for k in range(4)]
    if CFG["narrow"]:
        L += ["s_branch L_vwd_@_%=", "L_vwn_@_%=:"]
        L += [f"v_mov_b32 {v(RM.VOFFWE + k)}, {v(RM.VOFFW + 2 + (k & 1))}" for k in range(4)]
        L += ["L_vwd_@_%=:"]
    return L


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
    ] + [t.replace("@", uid) for t in voffwe_update()] + [f"L_noadv_{uid}_%=:"]

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


# ---- per-tile epilogue operands: gate (gated-residual epilogue) and bias, fetched in a tile's FIRST step ----------
def tile_operand_loads_emit(e, nbw):
    """FIRST step, sub-step 0 (contiguous): raw gate rows -> GRAW, bias values -> BIASA[nb][0] (lanes >= 32 read out of range: 0),
    zeros / ones of the rank-1 fragments.  The loads are older than this step's direct-to-LDS batch, so the NEXT step's boundary
    wait covers them (nk >= 3: there is one before the values are used)."""
    if CFG["res"]:
        e.ins(f"v_and_b32 {v(RM.GOFF)}, 0x70, {v(RM.STOFF)}")                       # (lane & 7) * 16: the out pitch is a multiple of 256
        for nbp in range(nbw // 2):
            e.ins(f"global_load_dwordx4 {v(RM.GRAW + 4 * nbp, 4)}, {v(RM.GOFF)}, {s(RM.S_GP, 2)} offset:{nbp * 128}")
            e._vm(None)
    if CFG["bias"]:
        for nb in range(nbw):
            e.ins(f"buffer_load_ushort {v(RM.BIASA + 4 * nb)}, {o('bias_voff')}, {s(RM.S_BRS, 4)}, 0 offen offset:{nb * 64}")
            e._vm(None)
            for k in range(1, 4):
                e.ins(f"v_mov_b32 {v(RM.BIASA + 4 * nb + k)}, 0")
        e.ins(f"v_mov_b32 {v(RM.ONES)}, {o('ones0')}")
        for k in range(1, 4):
            e.ins(f"v_mov_b32 {v(RM.ONES + k)}, 0")


def tile_operand_finish(e, nbw):
    """NORMAL-before-last step, behind its boundary: gate -> fp32 (G), and the rank-1 bias product acc += bias (x) ones"""
    if CFG["res"]:
        for nbp in range(nbw // 2):
            for d in range(4):
                e.ins(f"v_lshlrev_b32 {v(RM.G + 8 * nbp + 2 * d)}, 16, {v(RM.GRAW + 4 * nbp + d)}")
                e.ins(f"v_and_b32 {v(RM.G + 8 * nbp + 2 * d + 1)}, 0xffff0000, {v(RM.GRAW + 4 * nbp + d)}")


def bias_mfmas(e, nbw):
    return [lambda nb=nb, mb=mb: e.mfma(acc(nb, mb), RM.BIASA + 4 * nb, RM.ONES) for nb in range(nbw) for mb in range(4)]


def step_normal(e, p, first, uid, vm_entry=0, next_last=False, nbw=4):
    """One k-step, sub-step major, consuming stage p.  Entry: the sub-step-0 fragments are in flight into buffer 0.
    first: the accumulators start here (C = 0 in sub-step 0).  vm_entry: vector-memory operations issued by the predecessor after
    ITS direct-to-LDS batch (they are younger than the batch this step waits for)."""
    buf = lambda b: ([b * 32 + nb * 4 for nb in range(4)], [b * 32 + 16 + mb * 4 for mb in range(4)])   # offsets in R
    to_v = lambda offs: [RM.R0 + x for x in offs]
    for ks in range(4):
        fa, fb = (to_v(x) for x in buf(ks & 1))
        na, nb_ = (to_v(x) for x in buf((ks + 1) & 1))
        pre = []
        if ks == 3:
            # boundary: my loads of the next k-step have landed, my reads of this stage are done -> barrier ->
            # the other stage may be read, this stage may be refilled
            e.ins(f"s_waitcnt vmcnt({vm_entry + e.vm_own}) lgkmcnt(0)")
            e.lgkm, e.vmq = [], []
            e.ins("s_barrier")
            # (the last-step layout lives in R[0:32) + R[64:80): clear of buffer 1 = R[32:64), which sub-step 3 is using)
            nxt = last_entry_reads(e, 1 - p) if next_last else frag_read_fillers(e, 1 - p, 0, na, nb_, nbw)
            fillers, cap = glds_fillers(e, p, with_reads=nxt), 1
            if next_last:            # (vmcnt(0) above: the tile's gate / bias loads, issued in its first step, have landed)
                tile_operand_finish(e, nbw)
                if CFG["bias"]:
                    pre = bias_mfmas(e, nbw)
        else:
            e.wait_lgkm0()
            fillers, cap = frag_read_fillers(e, p, ks + 1, na, nb_, nbw), 1
            if ks == 0:          # the load cursor moves on behind the batch the step before issued (entry contract of every step)
                fillers, cap = fillers + cursor_advance(e, uid), 3
                if first and (CFG["res"] or CFG["bias"]):
                    fillers = fillers + [lambda: tile_operand_loads_emit(e, nbw)]
        mf = [lambda nb=nb, mb=mb: e.mfma(acc(nb, mb), fa[nb], fb[mb], zero=(first and ks == 0)) for nb in range(nbw) for mb in range(4)]
        interleave(e, pre + mf, fillers, cap=cap)


def epilogue_E1(e, nbp, mb):
    """accumulators of the block pair (2 nbp, mb), (2 nbp + 1, mb) -> scratch, fp32, straight from the AGPRs"""
    ops = []
    for b in range(2):
        for q in range(4):
            ops.append(lambda b=b, q=q: e.ds_write_acc(RM.SCRW + b * 4 + q, acc(2 * nbp + b, mb) + 4 * q, 0))
    return ops


def epilogue_rows(e, nbp, mb, res16=None):
    """the pair's 32 output rows in 4 groups of 8: scratch (fp32) [+ residual rows] -> registers (double-buffered) -> bf16
    [out = residual + gate * value] -> 16-byte stores.  Loads of group g + 1 are requested before group g is converted.
    res16: 16 free registers for the residual rows of ALL four groups, requested up front — the memory counter retires in issue
    order, so a residual load issued behind a store is only usable once that store has been acknowledged by memory (microseconds):
    requested up front the pair waits once for the PREVIOUS pair's stores instead of once per group (measured: + 39 us on out_proj
    with one wait per group).  None: the 2 x 4 registers of RM.RES, two groups at a time."""
    res_of = (lambda g: res16 + 4 * g) if res16 is not None else (lambda g: RM.RES + 4 * (g & 1))

    def load(g):
        b = g & 1
        ops = []
        if CFG["res"] and res16 is None:
            ops.append(lambda: e.res_load(res_of(g), RM.S_SO + mb * 4 + g, nbp * 128))
        for h in range(2):
            ops.append(lambda h=h: e.ds_read(RM.ROW + 8 * b + 4 * h, RM.SCRR, g * 2048 + h * 16))
        return ops

    def conv(g):
        b = g & 1
        row, res, pk, t0, t1 = RM.ROW + 8 * b, res_of(g), RM.PK + 4 * b, RM.TMP, RM.TMP + 1

        def run():
            e.need(range(row, row + 8))
            if CFG["res"]:
                e.need_vm(range(res, res + 4))
            for d in range(4):
                if CFG["res"]:
                    gq = RM.G + 8 * nbp + 2 * d
                    e.ins(f"v_lshlrev_b32 {v(t0)}, 16, {v(res + d)}")
                    e.ins(f"v_and_b32 {v(t1)}, 0xffff0000, {v(res + d)}")
                    e.ins(f"v_fma_f32 {v(t0)}, {v(gq)}, {v(row + 2 * d)}, {v(t0)}")
                    e.ins(f"v_fma_f32 {v(t1)}, {v(gq + 1)}, {v(row + 2 * d + 1)}, {v(t1)}")
                    e.ins(f"v_cvt_pk_bf16_f32 {v(pk + d)}, {v(t0)}, {v(t1)}")
                else:
                    e.ins(f"v_cvt_pk_bf16_f32 {v(pk + d)}, {v(row + 2 * d)}, {v(row + 2 * d + 1)}")
        return [run, lambda: e.store(pk, RM.S_SO + mb * 4 + g, nbp * 128)]
    ops = load(0) + load(1)
    for g in range(4):
        ops += conv(g)
        if g + 2 < 4:
            ops += load(g + 2)
    return ops


def residual_loads(e, nbp, mb, res16):
    return [lambda g=g: e.res_load(res16 + 4 * g, RM.S_SO + mb * 4 + g, nbp * 128) for g in range(4)]


def step_last(e, p, uid, nbw=4):
    """Last k-step of a tile, block-pair major (pair = token block mb, feature blocks 2 nbp, 2 nbp + 1), with the epilogue of the pair
    before in the gaps of every pair and the boundary (next stage ready / this stage free) a quarter of the way in."""
    LA, LB = last_LA, last_LB
    nbps = nbw // 2
    pairs = [(mb, nbp) for mb in range(4) for nbp in range(nbps)]
    bpos = len(pairs) // 4                              # pair in front of which the boundary sits: 2 (wide), 1 (narrow)
    rd_a = lambda nbp: [lambda nbl=nbl, ks=ks: e.ds_read(LA(nbp, nbl, ks), RM.AADDR + p * 4 + ks, (2 * nbp + nbl) * 4096)
                        for ks in range(4) for nbl in range(2)]
    rd_b = lambda mb: [lambda ks=ks: e.ds_read(LB(mb, ks), RM.BADDR + p * 4 + ks, mb * 4096) for ks in range(4)]
    # entry: the 12 reads of last_entry_reads() are in flight (issued by the step before)
    e.lgkm = [set(range(LA(0, nbl, ks), LA(0, nbl, ks) + 4)) for ks in range(4) for nbl in range(2)] + \
             [set(range(LB(0, ks), LB(0, ks) + 4)) for ks in range(4)]
    # every remaining fragment read of this stage goes out before the boundary
    rest = (rd_a(1) if nbps == 2 else []) + rd_b(1) + rd_b(2) + rd_b(3)
    per = (len(rest) + bpos - 1) // bpos
    res_sets = {}                                       # pair index -> registers holding its residual rows, requested a pair ahead
    for sl, (mb, nbp) in enumerate(pairs):
        fill = []
        if sl < bpos:
            fill += rest[sl * per:(sl + 1) * per]
        if sl == bpos - 1:
            fill += cursor_advance(e, uid)              # (pending from the batch the step before issued)
        if sl >= 1:
            pmb, pnbp = pairs[sl - 1]
            # B fragments of token block 0 are dead once its pairs have issued: their 16 registers take the pair's residual rows.  Once
            # token block 1 is done too there are TWO such sets: from then on the rows of the NEXT pair are requested here as well, in FRONT
            # of this pair's stores — a load behind a store is usable only once memory has acknowledged that store (the counter retires in
            # issue order), so without the second set every pair starts by waiting out the stores of the pair before
            res16, loads = None, []
            if CFG["res"] and mb >= 1:
                if (sl - 1) in res_sets:
                    res16 = res_sets[sl - 1]                                  # requested one pair ago
                else:
                    res16 = last_LB(0, 0)
                    loads = residual_loads(e, pnbp, pmb, res16)
                if RES_AHEAD and mb >= 2:
                    nxt = last_LB(1, 0) if res16 == last_LB(0, 0) else last_LB(0, 0)
                    res_sets[sl] = nxt
                    loads = loads + residual_loads(e, nbp, mb, nxt)          # (the pair this step computes: its epilogue runs in the next step)
            fill += loads + epilogue_E1(e, pnbp, pmb) + epilogue_rows(e, pnbp, pmb, res16)
        if sl == bpos:
            e.ins(f"s_waitcnt vmcnt({e.vm_own}) lgkmcnt(0)")      # (everything of mine so far is younger than the awaited batch)
            e.lgkm, e.vmq = [], []
            e.ins("s_barrier")
            fill = glds_fillers(e, p) + fill            # the batch first: the stores of the pairs from here on are younger than it
        if sl == len(pairs) - 1:
            # sub-step-0 fragments of the next k-step (other stage) into buffer 0 = R[0:32): LA(0, ...) is free behind the last pair
            # of feature pair 0 (the wide layout: a narrow next tile simply ignores two of them)
            fill += frag_read_fillers(e, 1 - p, 0, [RM.R0 + nb * 4 for nb in range(4)], [RM.R0 + 16 + m * 4 for m in range(4)])
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
    pmb, pnbp = pairs[-1]
    res16 = (res_sets.get(len(pairs) - 1) or last_LB(0, 0)) if CFG["res"] else None
    tail_loads = residual_loads(e, pnbp, pmb, res16) if (res16 is not None and (len(pairs) - 1) not in res_sets) else []
    for f in tail_loads + epilogue_E1(e, pnbp, pmb) + epilogue_rows(e, pnbp, pmb, res16):
        f()
    return e.vm_after_glds


def tile_descriptors():
    """output (and residual) buffer descriptors, gate row pointer and bias descriptor of the wave tile the compute cursor is on;
    sets S_CNARROW.  Contiguous scalar code (runs between tiles and in the prologue)."""
    T = RM.S_T
    L = [f"s_mov_b32 {s(RM.S_CNARROW)}, 0"]
    if CFG["narrow"]:
        L += [f"s_cmp_eq_u32 {s(RM.S_CNT)}, {s(RM.S_NWIDE)}", f"s_cselect_b32 {s(RM.S_CNARROW)}, 1, 0"]
    L += [
        # rows: ((c_mt * 2 + wm) * 128) * out pitch;  columns (bytes): c_nt * 512 + wn * (256 wide | 128 narrow)
        f"s_lshl_b32 {s(T)}, {s(RM.S_CMT)}, 1", f"s_add_u32 {s(T)}, {s(T)}, {s(RM.S_WM)}", f"s_lshl_b32 {s(T + 4)}, {s(T)}, 7",
        f"s_mul_i32 {s(T)}, {s(T + 4)}, {s(RM.S_OP)}",
        f"s_cmp_eq_u32 {s(RM.S_CNARROW)}, 1", f"s_cselect_b32 {s(T + 2)}, 7, 8", f"s_lshl_b32 {s(T + 2)}, {s(RM.S_WN)}, {s(T + 2)}",
        f"s_lshl_b32 {s(T + 1)}, {s(RM.S_CNT)}, 9", f"s_add_u32 {s(T + 1)}, {s(T + 1)}, {s(T + 2)}",      # T1 = column byte offset
        f"s_add_u32 {s(T)}, {s(T)}, {s(T + 1)}",
        f"s_add_u32 {s(RM.S_RS)}, {s(RM.S_OUT)}, {s(T)}", f"s_addc_u32 {s(RM.S_RS + 1)}, {s(RM.S_OUT + 1)}, 0",
        f"s_and_b32 {s(RM.S_RS + 1)}, {s(RM.S_RS + 1)}, 0xffff",
        f"s_mov_b32 {s(RM.S_RS + 2)}, 0x7ffffffe", f"s_mov_b32 {s(RM.S_RS + 3)}, 0x00020000",
    ]
    if CFG["res"]:
        L += [
            f"s_add_u32 {s(RM.S_RRS)}, {o('res_lo')}, {s(T)}", f"s_addc_u32 {s(RM.S_RRS + 1)}, {o('res_hi')}, 0",
            f"s_and_b32 {s(RM.S_RRS + 1)}, {s(RM.S_RRS + 1)}, 0xffff",
            f"s_mov_b32 {s(RM.S_RRS + 2)}, 0x7ffffffe", f"s_mov_b32 {s(RM.S_RRS + 3)}, 0x00020000",
            # gate row of the sample this wave tile lies in: sample = first row >> log2(rows per sample)
            f"s_lshr_b32 {s(T + 2)}, {s(T + 4)}, {o('rpb_shift')}", f"s_mul_i32 {s(T + 2)}, {s(T + 2)}, {o('gate_bstride')}",
            f"s_add_u32 {s(T + 2)}, {s(T + 2)}, {s(T + 1)}",
            f"s_add_u32 {s(RM.S_GP)}, {o('gate_lo')}, {s(T + 2)}", f"s_addc_u32 {s(RM.S_GP + 1)}, {o('gate_hi')}, 0",
        ]
    if CFG["bias"]:
        L += [
            f"s_add_u32 {s(RM.S_BRS)}, {o('bias_lo')}, {s(T + 1)}", f"s_addc_u32 {s(RM.S_BRS + 1)}, {o('bias_hi')}, 0",
            f"s_and_b32 {s(RM.S_BRS + 1)}, {s(RM.S_BRS + 1)}, 0xffff",
            f"s_mov_b32 {s(RM.S_BRS + 2)}, 0x4000", f"s_mov_b32 {s(RM.S_BRS + 3)}, 0x00020000",
        ]
    return L


def tile_advance(uid):
    """compute cursor -> next tile of this workgroup, then its descriptors"""
    L = [
        f"s_add_u32 {s(RM.S_CMT)}, {s(RM.S_CMT)}, {s(RM.S_STM)}",
        f"s_add_u32 {s(RM.S_CNT)}, {s(RM.S_CNT)}, {s(RM.S_STN)}",
        f"s_cmp_ge_u32 {s(RM.S_CNT)}, {s(RM.S_TN)}",
        f"s_cbranch_scc0 L_cnw_{uid}_%=",
        f"s_sub_u32 {s(RM.S_CNT)}, {s(RM.S_CNT)}, {s(RM.S_TN)}",
        f"s_add_u32 {s(RM.S_CMT)}, {s(RM.S_CMT)}, 1",
        f"L_cnw_{uid}_%=:",
    ]
    return L + tile_descriptors()


# operands of the asm statement: (constraint, C expression) in order; the body refers to them as %0 ...  Scalars that never change
# arrive packed (the statement may have at most 30 operands); pointers of the epilogue operands as two halves (plain scalar sources).
OPERANDS = [
    ("v", "voffw0"), ("v", "voffx0"), ("v", "a_base"), ("v", "b_base"), ("v", "t_xor"), ("v", "scrw_base"), ("v", "j7"),
    ("v", "scrr"), ("v", "stoff"), ("v", "bias_voff"), ("v", "ones0"),
    ("s", "w_ptr"), ("s", "x_ptr"), ("s", "out_ptr"), ("s", "w_pitch"), ("s", "x_pitch"), ("s", "o_pitch"),
    ("s", "dims"),          # nk | tiles_n << 12 | n_wide << 22
    ("s", "my_tiles"),
    ("s", "steps"),         # step_m | step_n << 20
    ("s", "tile0"),         # mt0 | nt0 << 20
    ("s", "wave_lds"),      # LDS base (a multiple of 1024) + wave
    ("s", "res_lo"), ("s", "res_hi"), ("s", "gate_lo"), ("s", "gate_hi"), ("s", "gate_bstride"), ("s", "rpb_shift"),
    ("s", "bias_lo"), ("s", "bias_hi"),
]
assert len(OPERANDS) <= 30
OP = {name: i for i, (_, name) in enumerate(OPERANDS)}


def prologue(e):
    e.in_prologue = True
    T, TV = RM.S_T, RM.TMP
    L = []
    for dst, src in ((RM.S_W, "w_ptr"), (RM.S_X, "x_ptr"), (RM.S_OUT, "out_ptr")):
        L.append(f"s_mov_b64 {s(dst, 2)}, {o(src)}")
    for dst, src in ((RM.S_WP, "w_pitch"), (RM.S_XP, "x_pitch"), (RM.S_OP, "o_pitch"), (RM.S_TLEFT, "my_tiles")):
        L.append(f"s_mov_b32 {s(dst)}, {o(src)}")
    L += [f"s_and_b32 {s(RM.S_NK)}, {o('dims')}, 0xfff", f"s_lshr_b32 {s(RM.S_TN)}, {o('dims')}, 12", f"s_and_b32 {s(RM.S_TN)}, {s(RM.S_TN)}, 0x3ff",
          f"s_lshr_b32 {s(RM.S_NWIDE)}, {o('dims')}, 22",
          f"s_and_b32 {s(RM.S_STM)}, {o('steps')}, 0xfffff", f"s_lshr_b32 {s(RM.S_STN)}, {o('steps')}, 20",
          f"s_and_b32 {s(RM.S_LMT)}, {o('tile0')}, 0xfffff", f"s_lshr_b32 {s(RM.S_LNT)}, {o('tile0')}, 20",
          f"s_mov_b32 {s(RM.S_CMT)}, {s(RM.S_LMT)}", f"s_mov_b32 {s(RM.S_CNT)}, {s(RM.S_LNT)}",
          f"s_mov_b32 {s(RM.S_LLEFT)}, {s(RM.S_TLEFT)}", f"s_mov_b32 {s(RM.S_LKT)}, {s(RM.S_NK)}",
          f"s_and_b32 {s(T)}, {o('wave_lds')}, 3", f"s_and_b32 {s(RM.S_WN)}, {s(T)}, 1", f"s_lshr_b32 {s(RM.S_WM)}, {s(T)}, 1",
          f"s_lshl_b32 {s(RM.S_LDSW)}, {s(T)}, 10", f"s_and_b32 {s(T)}, {o('wave_lds')}, 0xfffffc00", f"s_add_u32 {s(RM.S_LDSW)}, {s(RM.S_LDSW)}, {s(T)}"]
    for mb in range(4):
        for g in range(4):
            L += [f"s_mul_i32 {s(RM.S_SO + mb * 4 + g)}, {s(RM.S_OP)}, {mb * 32 + g * 8}"]
    # load cursor pointers of the first tile
    L += [f"s_lshl_b32 {s(T)}, {s(RM.S_LNT)}, 8", f"s_mul_i32 {s(T)}, {s(T)}, {s(RM.S_WP)}",
          f"s_add_u32 {s(RM.S_LW)}, {s(RM.S_W)}, {s(T)}", f"s_addc_u32 {s(RM.S_LW + 1)}, {s(RM.S_W + 1)}, 0",
          f"s_lshl_b32 {s(T)}, {s(RM.S_LMT)}, 8", f"s_mul_i32 {s(T)}, {s(T)}, {s(RM.S_XP)}",
          f"s_add_u32 {s(RM.S_LX)}, {s(RM.S_X)}, {s(T)}", f"s_addc_u32 {s(RM.S_LX + 1)}, {s(RM.S_X + 1)}, 0"]
    L += tile_descriptors()
    # per-lane tables
    for i in range(8):
        L += [f"s_mul_i32 {s(T)}, {s(RM.S_WP)}, {i * 32}", f"v_add_u32 {v(RM.VOFFW + i)}, {s(T)}, {o('voffw0')}",
              f"s_mul_i32 {s(T)}, {s(RM.S_XP)}, {i * 32}", f"v_add_u32 {v(RM.VOFFX + i)}, {s(T)}, {o('voffx0')}"]
    L += [t.replace("@", "pro") for t in voffwe_update()]
    for ks in range(4):
        L += [f"v_xor_b32 {v(TV)}, {ks * 2}, {o('t_xor')}", f"v_lshlrev_b32 {v(TV)}, 4, {v(TV)}",
              f"v_add_u32 {v(RM.AADDR + ks)}, {v(TV)}, {o('a_base')}", f"v_add_u32 {v(RM.BADDR + ks)}, {v(TV)}, {o('b_base')}",
              f"v_add_u32 {v(RM.AADDR + 4 + ks)}, {hex(STAGE)}, {v(RM.AADDR + ks)}", f"v_add_u32 {v(RM.BADDR + 4 + ks)}, {hex(STAGE)}, {v(RM.BADDR + ks)}"]
    for bq in range(8):
        L += [f"v_xor_b32 {v(TV)}, {bq}, {o('j7')}", f"v_lshlrev_b32 {v(TV)}, 5, {v(TV)}", f"v_add_u32 {v(RM.SCRW + bq)}, {v(TV)}, {o('scrw_base')}"]
    L += [f"v_mov_b32 {v(RM.SCRR)}, {o('scrr')}", f"v_mov_b32 {v(RM.STOFF)}, {o('stoff')}"]
    for t in L:
        (e.raw if t.endswith(":") else e.ins)(t)
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
    c += [f"v{i}" for i in range(32, 256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(36, 102)]
    return c


def generate(cfg=None):
    """-> (asm text with %N operands and %= label ids, T = vector-memory ops a LAST step issues behind its direct-to-LDS batch)"""
    global CFG
    CFG = dict(narrow=False, res=False, bias=False)
    CFG.update(cfg or {})
    out = []

    def block(fn, *args, **kw):
        e = Emit()
        r = fn(e, *args, **kw)
        out.extend(e.lines)
        return r

    widths = [4, 2] if CFG["narrow"] else [4]
    # what a LAST step leaves behind its batch (does not depend on the wait counts); the next tile's FIRST step may follow either width
    T_after = min(step_last(Emit(), 0, "probe", nbw=w) for w in widths)
    block(prologue)
    tag = lambda w: "w" if w == 4 else "n"

    def goto_by_width(prefix, q):
        """branch to `prefix`_{w|n}_q by the width of the compute tile"""
        if not CFG["narrow"]:
            return [f"s_branch {prefix}_w_{q}_%="]
        return [f"s_cmp_eq_u32 {s(RM.S_CNARROW)}, 1", f"s_cbranch_scc1 {prefix}_n_{q}_%=", f"s_branch {prefix}_w_{q}_%="]

    # control flow (nk >= 3): FIRST -> NORMAL x (nk - 3) -> NORMAL-before-last -> LAST, stage parity alternating; LAST -> FIRST
    out += goto_by_width("L_first0", 0)
    for w in widths:
        for p in range(2):
            q = 1 - p
            variants = [(f"L_first_{tag(w)}_{p}", True, T_after, False), (f"L_norm_{tag(w)}_{p}", False, 0, False),
                        (f"L_norml_{tag(w)}_{p}", False, 0, True)]
            if p == 0:
                variants.append((f"L_first0_{tag(w)}_0", True, 0, False))
            for name, first, vm_entry, next_last in variants:
                out.append(f"{name}_%=:")
                block(step_normal, p, first, name[2:], vm_entry=vm_entry, next_last=next_last, nbw=w)
                if next_last:
                    out += [f"s_branch L_last_{tag(w)}_{q}_%="]
                    continue
                if first:
                    out += [f"s_sub_u32 {s(RM.S_KCNT)}, {s(RM.S_NK)}, 2"]
                else:
                    out += [f"s_sub_u32 {s(RM.S_KCNT)}, {s(RM.S_KCNT)}, 1"]
                out += [f"s_cmp_eq_u32 {s(RM.S_KCNT)}, 1", f"s_cbranch_scc1 L_norml_{tag(w)}_{q}_%=", f"s_branch L_norm_{tag(w)}_{q}_%="]
            out.append(f"L_last_{tag(w)}_{p}_%=:")
            block(step_last, p, f"last{tag(w)}{p}", nbw=w)
            out += [f"s_sub_u32 {s(RM.S_TLEFT)}, {s(RM.S_TLEFT)}, 1", f"s_cmp_eq_u32 {s(RM.S_TLEFT)}, 0", f"s_cbranch_scc1 L_end_%="]
            out += tile_advance(f"ta{tag(w)}{p}")
            out += goto_by_width("L_first", q)
    out += ["L_end_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return out, T_after


PROBE_VARIANTS = {"NOMFMA": {"nomfma"}, "LOADS": {"nomfma", "noreads", "nostore"}, "NOGLDS": {"noglds"}, "NOSTORE": {"nostore"},
                  "MFMAONLY": {"noglds", "nostore"},
                  "NOGLDS_LAX": {"noglds", "laxvm"}, "MFMA_NOEPI": {"noglds", "nostore", "noepi"}}
# kernel variants: name -> what the body supports
VARIANTS = {"": dict(), "_N": dict(narrow=True), "_NR": dict(narrow=True, res=True), "_NRB": dict(narrow=True, res=True, bias=True)}


def emit_inc(path):
    global OPTS
    with open(path, "w") as fh:
        fh.write("// GENERATED by zigma_amd/csrc/gen/linear4w_gen.py --emit — do not edit (tests/test_linear4w_gen.py checks it is current).\n")
        fh.write("// The main loop of linear4w_kernel as one asm statement per variant; operands in the order of OPERANDS in the generator.\n")

        def body(name, cfg, opts=()):
            global OPTS
            OPTS = set(opts)
            lines, _ = generate(cfg)
            OPTS = set()
            fh.write(f"#define ZIGMA_LINEAR4W_BODY{name} \\\n")
            for ln in lines:
                fh.write(f'    "{ln}\\n" \\\n')
            fh.write("    \"\"\n")
        for name, cfg in VARIANTS.items():
            body(name, cfg)
        fh.write("#ifdef ZIGMA_LINEAR4W_PROBES   // timing probes (tools/linear4w_probe.py builds its own library with them): results are wrong\n")
        for name, opts in PROBE_VARIANTS.items():
            body("_" + name, {}, opts)
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
        for name, cfg in VARIANTS.items():
            lines, T = generate(cfg)
            n = sum(1 for l in lines if not l.endswith(":"))
            print(f"variant '{name}': {n} instructions, {sum('v_mfma' in l for l in lines)} MFMAs, T = {T}")
    if args.check:
        sys.path.insert(0, HERE)
        import linear4w_sim
        linear4w_sim.main()
