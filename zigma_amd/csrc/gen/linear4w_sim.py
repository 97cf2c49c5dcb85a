"""Simulator / checker for the generated body of linear4w_kernel (see linear4w_gen.py).  TEST INFRASTRUCTURE.

Executes the SAME text the assembler gets, for the 4 waves of one workgroup, with numpy as the 64 lanes:
  * functional: SGPR / VGPR / AGPR files, SCC, M0, LDS bytes, global memory, MFMA 32x32x16 bf16, the packed conversion;
  * synchronisation discipline (what a correct run on hardware relies on, not what happens to work):
      - a register written by an LDS read / vector load is unusable until an s_waitcnt covers it (both counters retire in order);
      - an LDS granule filled by a direct-to-LDS load is readable by the issuing wave after its covering vmcnt wait, by the other
        waves only after a barrier that follows that wait; it may be refilled only after every other wave's last read of it lies
        behind a barrier and the issuing wave's own reads of it have been waited for;
      - an accumulator block is read (ds_write from AGPRs) no sooner than 12 instructions after the MFMA that wrote it;
      - M0 is not consumed by the instruction right behind its write; counter immediates stay inside their fields.
Waves run one after the other between barriers, in both orders."""
import re

import numpy as np

LDS_BYTES = 163840
GRAN = 16


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return (r & 0xFFFF).astype(np.uint32)


class SimError(Exception):
    pass


class Memory:
    def __init__(self):
        self.regions = []           # (base, bytearray-like np.uint8 array, writable)

    def add(self, base, arr, writable=False):
        self.regions.append((base, arr, writable))

    def find(self, addr, n, write=False):
        for base, arr, wr in self.regions:
            if base <= addr and addr + n <= base + arr.size:
                if write and not wr:
                    raise SimError(f"store to read-only region at {addr:#x}")
                return arr, addr - base
        raise SimError(f"memory access outside every buffer: {addr:#x} (+{n})")


class Wave:
    def __init__(self, wid, prog, labels, mem, lds, shared):
        self.wid, self.prog, self.labels, self.mem, self.lds, self.sh = wid, prog, labels, mem, lds, shared
        self.S = np.zeros(128, np.uint32)
        self.V = np.zeros((256, 64), np.uint32)
        self.A = np.zeros((256, 64), np.uint32)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.interval = 0
        self.vm = []                # outstanding vector-memory ops, oldest first
        self.lgkm = []              # outstanding LDS ops
        self.vpend = {}             # vgpr -> description of the outstanding load into it
        self.n_exec = 0
        self.mfma_at = {}           # accumulator block base -> n_exec of the last MFMA writing it
        self.m0_written_at = -10
        self.done = False
        self.stats = dict(mfma=0, inst=0)

    # ---- operands ----------------------------------------------------------------------------------------
    @staticmethod
    def rng(tok):
        m = re.fullmatch(r"([vas])\[(\d+):(\d+)\]", tok)
        if m:
            return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        m = re.fullmatch(r"([vas])(\d+)", tok)
        if m:
            return m.group(1), int(m.group(2)), 1
        return None

    def rs(self, tok):
        """scalar source -> python int (uint32)"""
        if tok == "m0":
            return self.m0
        r = self.rng(tok)
        if r:
            assert r[0] == "s" and r[2] == 1, tok
            return int(self.S[r[1]])
        return int(tok, 0) & 0xFFFFFFFF

    def rs64(self, tok):
        r = self.rng(tok)
        assert r and r[0] == "s" and r[2] == 2, tok
        return int(self.S[r[1]]) | (int(self.S[r[1] + 1]) << 32)

    def ws(self, tok, val):
        val &= 0xFFFFFFFF
        if tok == "m0":
            self.m0 = val
            self.m0_written_at = self.n_exec
            return
        r = self.rng(tok)
        assert r and r[0] == "s" and r[2] == 1, tok
        self.S[r[1]] = val

    def rv(self, tok):
        """vector source -> (64,) uint32"""
        r = self.rng(tok)
        if r and r[0] == "v":
            assert r[2] == 1
            self.chk_v(r[1], 1, tok)
            return self.V[r[1]].copy()
        if r and r[0] == "s":
            return np.full(64, self.S[r[1]], np.uint32)
        return np.full(64, int(tok, 0) & 0xFFFFFFFF, np.uint32)

    def chk_v(self, base, n, what):
        for i in range(base, base + n):
            if i in self.vpend:
                raise SimError(f"wave {self.wid} pc {self.pc}: v{i} used by `{what}` while {self.vpend[i]} is outstanding")

    # ---- execution ---------------------------------------------------------------------------------------
    def run_to_barrier(self):
        """execute until an s_barrier has been executed (returns True) or the program ends (False)"""
        while self.pc < len(self.prog):
            op, args, text = self.prog[self.pc]
            self.pc += 1
            self.n_exec += 1
            self.stats["inst"] += 1
            try:
                if self.step(op, args, text):
                    return True
            except SimError:
                raise
            except Exception as ex:
                raise SimError(f"wave {self.wid} at `{text}`: {type(ex).__name__}: {ex}")
        self.done = True
        return False

    def branch(self, label):
        self.pc = self.labels[label]

    def step(self, op, args, text):
        S = self.S
        if op == "s_barrier":
            self.interval += 1
            return True
        if op == "s_nop":
            self.n_exec += int(args[0])
        elif op == "s_waitcnt":
            for m in re.finditer(r"(vmcnt|lgkmcnt)\((\d+)\)", text):
                n = int(m.group(2))
                if m.group(1) == "vmcnt":
                    assert n <= 63
                    while len(self.vm) > n:
                        self.retire_vm(self.vm.pop(0))
                else:
                    assert n <= 15
                    while len(self.lgkm) > n:
                        for r in self.lgkm.pop(0)[0]:
                            self.vpend.pop(r, None)
        elif op == "s_mov_b32":
            self.ws(args[0], self.rs(args[1]))
        elif op == "s_mov_b64":
            d = self.rng(args[0])
            val = self.rs64(args[1])
            S[d[1]], S[d[1] + 1] = val & 0xFFFFFFFF, val >> 32
        elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32"):
            x, y = self.rs(args[1]), self.rs(args[2])
            if op == "s_add_u32":
                r = x + y
                self.scc = int(r > 0xFFFFFFFF)
            elif op == "s_addc_u32":
                r = x + y + self.scc
                self.scc = int(r > 0xFFFFFFFF)
            else:
                r = x - y
                self.scc = int(y > x)
            self.ws(args[0], r)
        elif op == "s_mul_i32":
            self.ws(args[0], self.rs(args[1]) * self.rs(args[2]))
        elif op == "s_lshl_b32":
            r = (self.rs(args[1]) << (self.rs(args[2]) & 31)) & 0xFFFFFFFF
            self.ws(args[0], r)
            self.scc = int(r != 0)
        elif op == "s_lshr_b32":
            r = self.rs(args[1]) >> (self.rs(args[2]) & 31)
            self.ws(args[0], r)
            self.scc = int(r != 0)
        elif op == "s_and_b32":
            r = self.rs(args[1]) & self.rs(args[2])
            self.ws(args[0], r)
            self.scc = int(r != 0)
        elif op == "s_cselect_b32":
            self.ws(args[0], self.rs(args[1]) if self.scc else self.rs(args[2]))
        elif op.startswith("s_cmp_"):
            x, y = self.rs(args[0]), self.rs(args[1])
            self.scc = int({"eq": x == y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y, "lg": x != y}[op.split("_")[2]])
            assert op.endswith("_u32")
        elif op == "s_cbranch_scc0":
            if not self.scc:
                self.branch(args[0])
        elif op == "s_cbranch_scc1":
            if self.scc:
                self.branch(args[0])
        elif op == "s_branch":
            self.branch(args[0])
        # ---- VALU ---------------------------------------------------------------------------------------
        elif op == "v_mov_b32":
            self.wv(args[0], self.rv(args[1]))
        elif op == "v_add_u32":
            self.wv(args[0], self.rv(args[1]) + self.rv(args[2]))
        elif op == "v_xor_b32":
            self.wv(args[0], self.rv(args[1]) ^ self.rv(args[2]))
        elif op == "v_and_b32":
            self.wv(args[0], self.rv(args[1]) & self.rv(args[2]))
        elif op == "v_lshlrev_b32":
            self.wv(args[0], self.rv(args[2]) << (self.rv(args[1]) & 31))
        elif op == "v_fma_f32":
            f = lambda t: self.rv(t).view(np.float32)
            self.wv(args[0], (f(args[1]).astype(np.float64) * f(args[2]) + f(args[3])).astype(np.float32).view(np.uint32))
        elif op == "v_cvt_pk_bf16_f32":
            lo, hi = self.rv(args[1]).view(np.float32), self.rv(args[2]).view(np.float32)
            self.wv(args[0], f32_to_bf16(lo) | (f32_to_bf16(hi) << 16))
        elif op == "v_mfma_f32_32x32x16_bf16":
            self.mfma(args, text)
        # ---- LDS ----------------------------------------------------------------------------------------
        elif op == "ds_read_b128":
            d = self.rng(args[0])
            addr = self.rv(args[1]).astype(np.int64) + self.imm(text)
            self.chk_v(d[1], 4, text)            # (a second load into a register whose first load is outstanding)
            grans = set()
            for l in range(64):
                g = self.lds_read_check(int(addr[l]), text)
                if g is not None:
                    grans.add(g)
            idx = addr[:, None] + np.arange(16)[None, :]
            data = np.ascontiguousarray(self.lds[idx]).view(np.uint32)             # (64, 4)
            for k in range(4):
                self.V[d[1] + k] = data[:, k]
                self.vpend[d[1] + k] = f"`{text}`"
            self.lgkm.append((list(range(d[1], d[1] + 4)), grans))
        elif op == "ds_write_b128":
            src = self.rng(args[1])
            addr = self.rv(args[0]).astype(np.int64) + self.imm(text)
            if src[0] == "a":
                blk = src[1] // 16 * 16
                if blk in self.mfma_at and self.n_exec - self.mfma_at[blk] - 1 < 12:
                    raise SimError(f"wave {self.wid}: `{text}` reads an accumulator {self.n_exec - self.mfma_at[blk] - 1} instructions after its MFMA")
                data = np.stack([self.A[src[1] + k] for k in range(4)], 1)
            else:
                self.chk_v(src[1], 4, text)
                data = np.stack([self.V[src[1] + k] for k in range(4)], 1)
            for l in range(64):
                g = int(addr[l]) // GRAN
                if g in self.sh["dma"]:
                    raise SimError(f"wave {self.wid}: `{text}` writes LDS {int(addr[l]):#x} where a direct-to-LDS load is outstanding")
                if int(addr[l]) < 2 * 65536:
                    raise SimError(f"wave {self.wid}: `{text}` writes into a stage")
            idx = addr[:, None] + np.arange(16)[None, :]
            self.lds[idx] = np.ascontiguousarray(data).view(np.uint8).reshape(64, 16)
            self.lgkm.append(([], set()))
        # ---- vector memory ------------------------------------------------------------------------------
        elif op == "global_load_lds_dwordx4":
            if self.n_exec - self.m0_written_at < 2:
                raise SimError(f"wave {self.wid}: `{text}` right behind the write of M0")
            voff = self.rv(args[0]).astype(np.int64)
            base = self.rs64(args[1])
            grans = []
            for l in range(64):
                src_arr, off = self.mem.find(base + int(voff[l]), 16)
                dst = self.m0 + l * 16
                if not (0 <= dst and dst + 16 <= 2 * 65536):
                    raise SimError(f"wave {self.wid}: `{text}` lands outside the stages ({dst:#x})")
                g = dst // GRAN
                # refill: every OTHER wave's last read of this granule lies behind a barrier; my own reads have been waited for
                for w2, iv in self.sh["last_read"].get(g, {}).items():
                    if w2 != self.wid and iv >= self.interval:
                        raise SimError(f"wave {self.wid}: `{text}` refills LDS {dst:#x} that wave {w2} read in barrier interval {iv} (now {self.interval})")
                for wv2 in self.sh["waves"]:                 # a read that was issued but never waited for may still be in flight
                    for regs, gs in wv2.lgkm:
                        if g in gs:
                            raise SimError(f"wave {self.wid}: `{text}` refills LDS {dst:#x} while a read of it by wave {wv2.wid} is outstanding")
                self.lds[dst:dst + 16] = src_arr[off:off + 16]
                self.sh["dma"][g] = self.wid
                grans.append(g)
            self.vm.append(("glds", grans))
        elif op == "buffer_store_dwordx4":
            d = self.rng(args[0])
            self.chk_v(d[1], 4, text)
            voff = self.rv(args[1]).astype(np.int64)
            rs = self.rng(args[2])
            base = int(self.S[rs[1]]) | ((int(self.S[rs[1] + 1]) & 0xFFFF) << 32)
            nrec = int(self.S[rs[1] + 2])
            soff = self.rs(args[3])
            data = np.stack([self.V[d[1] + k] for k in range(4)], 1)
            for l in range(64):
                off = int(voff[l]) + soff + self.imm(text)
                if off + 16 > nrec:
                    continue
                arr, o = self.mem.find(base + off, 16, write=True)
                arr[o:o + 16] = np.ascontiguousarray(data[l]).view(np.uint8)
                self.sh["stored"][base + off] = self.sh["stored"].get(base + off, 0) + 1
            self.vm.append(("store", None))
        elif op in ("buffer_load_dwordx4", "buffer_load_ushort", "global_load_dwordx4"):
            d = self.rng(args[0])
            nreg, nbytes = (4, 16) if op.endswith("dwordx4") else (1, 2)
            self.chk_v(d[1], nreg, text)
            voff = self.rv(args[1]).astype(np.int64)
            if op.startswith("buffer"):
                rs = self.rng(args[2])
                base = int(self.S[rs[1]]) | ((int(self.S[rs[1] + 1]) & 0xFFFF) << 32)
                nrec = int(self.S[rs[1] + 2])
                soff = self.rs(args[3])
            else:
                base, nrec, soff = self.rs64(args[2]), 1 << 62, 0
            data = np.zeros((64, 4), np.uint32)
            for l in range(64):
                off = int(voff[l]) + soff + self.imm(text)
                if off + nbytes > nrec:
                    continue                                  # out of range: 0
                arr, o_ = self.mem.find(base + off, nbytes)
                if nbytes == 16:
                    data[l] = np.ascontiguousarray(arr[o_:o_ + 16]).view(np.uint32)
                else:
                    data[l, 0] = int(arr[o_]) | (int(arr[o_ + 1]) << 8)
            for k in range(nreg):
                self.V[d[1] + k] = data[:, k]
                self.vpend[d[1] + k] = f"`{text}`"
            self.vm.append(("load", list(range(d[1], d[1] + nreg))))
        else:
            raise SimError(f"unknown instruction `{text}`")
        return False

    def imm(self, text):
        m = re.search(r"offset:(\d+)", text)
        return int(m.group(1)) if m else 0

    def wv(self, tok, val):
        r = self.rng(tok)
        assert r and r[0] == "v" and r[2] == 1
        self.chk_v(r[1], 1, tok)
        self.V[r[1]] = val.astype(np.uint32)

    def retire_vm(self, ent):
        kind, grans = ent
        if kind == "load":
            for r in grans:
                self.vpend.pop(r, None)
        if kind == "glds":
            for g in grans:
                if self.sh["dma"].get(g) == self.wid:
                    del self.sh["dma"][g]
                self.sh["covered"][g] = (self.wid, self.interval)

    def lds_read_check(self, addr, text):
        if addr >= 2 * 65536:
            return None                                      # scratch: wave-private, in-order LDS queue
        g = addr // GRAN
        if g in self.sh["dma"]:
            raise SimError(f"wave {self.wid}: `{text}` reads LDS {addr:#x} while the load that fills it is not covered by a wait (issued by wave {self.sh['dma'][g]})")
        cov = self.sh["covered"].get(g)
        if cov is None:
            raise SimError(f"wave {self.wid}: `{text}` reads LDS {addr:#x} that was never filled")
        w, iv = cov
        if w != self.wid and not (self.interval > iv):
            raise SimError(f"wave {self.wid}: `{text}` reads LDS {addr:#x} filled by wave {w} whose wait is not behind a barrier (intervals {iv} / {self.interval})")
        self.sh["last_read"].setdefault(g, {})[self.wid] = self.interval
        return g

    def mfma(self, args, text):
        d, fa, fb = self.rng(args[0]), self.rng(args[1]), self.rng(args[2])
        assert d[0] == "a" and d[2] == 16 and fa[2] == 4 and fb[2] == 4
        self.chk_v(fa[1], 4, text)
        self.chk_v(fb[1], 4, text)

        def frag(base):          # (64 lanes, 8 bf16) -> matrix [row = l % 32][k = 8 (l // 32) + e]
            raw = np.stack([self.V[base + k] for k in range(4)], 1)      # (64, 4) u32
            lo, hi = bf16_to_f32((raw & 0xFFFF).astype(np.uint16)), bf16_to_f32((raw >> 16).astype(np.uint16))
            e8 = np.stack([lo, hi], 2).reshape(64, 8)
            M = np.zeros((32, 16), np.float32)
            M[:, 0:8], M[:, 8:16] = e8[:32], e8[32:]
            return M
        Am, Bm = frag(fa[1]), frag(fb[1])
        P = (Am.astype(np.float64) @ Bm.astype(np.float64).T).astype(np.float32)        # D[i][j], i from A rows, j from B rows
        if args[3] == "0":
            C = np.zeros((32, 32), np.float32)
        else:
            c = self.rng(args[3])
            assert c == d
            C = self.acc_matrix(d[1])
        self.set_acc_matrix(d[1], C + P)
        self.mfma_at[d[1]] = self.n_exec
        self.stats["mfma"] += 1

    def acc_matrix(self, base):
        D = np.zeros((32, 32), np.float32)
        regs = self.A[base:base + 16].view(np.float32)                  # (16, 64)
        for r in range(16):
            for half in range(2):
                D[(r & 3) + 8 * (r >> 2) + 4 * half, :] = regs[r, half * 32:(half + 1) * 32]
        return D

    def set_acc_matrix(self, base, D):
        regs = np.zeros((16, 64), np.float32)
        for r in range(16):
            for half in range(2):
                regs[r, half * 32:(half + 1) * 32] = D[(r & 3) + 8 * (r >> 2) + 4 * half, :]
        self.A[base:base + 16] = regs.view(np.uint32)


def parse(lines, operand_values):
    """text -> [(op, args, text)], labels.  operand_values: per wave substitution of %N is done by the caller (here N -> token)"""
    prog, labels = [], {}
    for ln in lines:
        t = ln.strip().replace("_%=", "")
        if not t:
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(prog)
            continue
        t = re.sub(r"%(\d+)", lambda m: operand_values[int(m.group(1))], t)
        op, _, rest = t.partition(" ")
        args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
        # strip trailing modifiers from the last args (offen, offset:..)
        args = [x.split(" ")[0] for x in args]
        prog.append((op, args, t))
    return prog, labels


def run_workgroup(lines, operands_for_wave, mem, order=(0, 1, 2, 3)):
    """operands_for_wave(w) -> (list of tokens for %0.., dict of preset registers {('s'|'v', idx): value or (64,) array})"""
    lds = np.zeros(LDS_BYTES, np.uint8)
    shared = dict(dma={}, covered={}, last_read={}, own_read_pending={}, stored={}, waves=[])
    waves = []
    for w in range(4):
        toks, preset = operands_for_wave(w)
        prog, labels = parse(lines, toks)
        wv = Wave(w, prog, labels, mem, lds, shared)
        for (kind, idx), val in preset.items():
            if kind == "s":
                wv.S[idx] = np.uint32(val & 0xFFFFFFFF)
            else:
                wv.V[idx] = np.asarray(val, dtype=np.int64).astype(np.uint32)
        waves.append(wv)
    shared["waves"] = waves
    while True:
        hit = [waves[w].run_to_barrier() for w in order]
        if all(hit):
            continue
        if any(hit):
            raise SimError(f"barrier mismatch: {hit}")
        break
    return waves, shared


# ------------------------------------------------------------------------------------------------------------------
# a problem instance, exactly as csrc/linear4w.hip sets the operands up
# ------------------------------------------------------------------------------------------------------------------
def lane_operands(wave, w_pitch, x_pitch, o_pitch, lds_base=0):
    lane = np.arange(64)
    j, kh = lane & 31, lane >> 5
    wn, wm = wave & 1, wave >> 1
    piece = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))
    srow = wave * 8 + (lane >> 3)
    sw = (j >> 1) & 7
    u, t8 = lane & 7, lane >> 3
    return dict(
        voffw0=srow * w_pitch + piece * 16,
        voffx0=srow * x_pitch + piece * 16,
        a_base=lds_base + (wn * 128 + j) * 128,
        b_base=lds_base + (256 + wm * 128 + j) * 128,
        t_xor=kh ^ sw,
        scrw_base=lds_base + 2 * 65536 + wave * 8192 + j * 256 + kh * 16,
        j7=j & 7,
        scrr=lds_base + 2 * 65536 + wave * 8192 + t8 * 256 + ((u ^ t8) << 5),
        stoff=t8 * o_pitch + u * 16,
        bias_voff=np.where(lane < 32, j * 2, 0x7FFF0000),
        ones0=np.where(lane < 32, 0x3F80, 0),
    )


def simulate(M, N, K, n_wg=8, wg=0, seed=0, order=(0, 1, 2, 3), gen=None, cfg=None, rows_per_batch=256):
    """one workgroup `wg` of `n_wg` (grid semantics of the launcher: a multiple of 8 workgroups, XCD-chunked tile lists); returns
    (out as float32 (M, N), mask of what this workgroup owns, float64 reference, waves)"""
    import linear4w_gen as G
    cfg = dict(cfg or {})
    lines, _ = gen or G.generate(cfg)
    res_on, bias_on = bool(cfg.get("res")), bool(cfg.get("bias"))
    rng = np.random.default_rng(seed)
    rb = lambda a: f32_to_bf16(a.astype(np.float32)).astype(np.uint16)
    x = rb(rng.standard_normal((M, K)))
    w = rb(rng.standard_normal((N, K)) * K ** -0.5)
    nb_ = M // rows_per_batch
    res = rb(rng.standard_normal((M, N)))
    gate = rb(rng.standard_normal((nb_, N)))
    bias = rb(rng.standard_normal(N) * 0.5)
    out = np.full((M, N), 0x7FC0, np.uint16)            # NaN pattern: untouched outputs show
    mem = Memory()
    WB, XB, OB, RB, GB, BB = 0x10000000, 0x20000000, 0x40000000, 0x60000000, 0x70000000, 0x78000000
    mem.add(WB, w.view(np.uint8).reshape(-1))
    mem.add(XB, x.view(np.uint8).reshape(-1))
    mem.add(OB, out.view(np.uint8).reshape(-1), writable=True)
    mem.add(RB, res.view(np.uint8).reshape(-1))
    mem.add(GB, gate.view(np.uint8).reshape(-1))
    mem.add(BB, bias.view(np.uint8).reshape(-1))
    tiles_m, n_wide, narrow = M // 256, N // 256, int(N % 256 != 0)
    assert not narrow or cfg.get("narrow")
    tiles_n = n_wide + narrow
    n_tiles = tiles_m * tiles_n
    xcd, slot, wg_per_xcd = wg & 7, wg >> 3, n_wg >> 3
    chunk = (n_tiles + 7) >> 3
    chunk_end = min((xcd + 1) * chunk, n_tiles)
    tile0 = xcd * chunk + slot
    if tile0 >= chunk_end:
        return None
    my_tiles = (chunk_end - tile0 + wg_per_xcd - 1) // wg_per_xcd
    nk = K // 64
    sc = dict(w_ptr=WB, x_ptr=XB, out_ptr=OB, w_pitch=K * 2, x_pitch=K * 2, o_pitch=N * 2,
              dims=nk | (tiles_n << 12) | (n_wide << 22), my_tiles=my_tiles,
              steps=(wg_per_xcd // tiles_n) | ((wg_per_xcd % tiles_n) << 20), tile0=(tile0 // tiles_n) | ((tile0 % tiles_n) << 20),
              res_lo=RB & 0xFFFFFFFF, res_hi=RB >> 32, gate_lo=GB & 0xFFFFFFFF, gate_hi=GB >> 32, gate_bstride=N * 2,
              rpb_shift=int(np.log2(rows_per_batch)), bias_lo=BB & 0xFFFFFFFF, bias_hi=BB >> 32)

    def operands_for_wave(wv):
        lo = lane_operands(wv, K * 2, K * 2, N * 2)
        toks, preset = [], {}
        nv, ns = 0, 0
        for c, name in G.OPERANDS:
            if c == "v":
                toks.append(f"v{nv}")
                preset[("v", nv)] = lo[name]
                nv += 1
            else:
                val = (0 + wv) if name == "wave_lds" else sc[name]
                if name.endswith("_ptr"):
                    toks.append(f"s[{ns}:{ns + 1}]")
                    preset[("s", ns)], preset[("s", ns + 1)] = val & 0xFFFFFFFF, val >> 32
                    ns += 2
                else:
                    toks.append(f"s{ns}")
                    preset[("s", ns)] = val
                    ns += 1
        assert nv <= 32 and ns <= 36
        return toks, preset

    waves, shared = run_workgroup(lines, operands_for_wave, mem, order)
    f = lambda a: bf16_to_f32(a).astype(np.float64)
    val = f(x) @ f(w).T
    if bias_on:
        val = val + f(bias)[None, :]
    ref = val
    if res_on:
        ref = f(res) + np.repeat(f(gate), rows_per_batch, axis=0) * val
    owned = np.zeros((M, N), bool)
    t = tile0
    n_mfma = 0
    for _ in range(my_tiles):
        mt, nt = divmod(t, tiles_n)
        wdt = 256 if nt < n_wide else 128
        owned[mt * 256:(mt + 1) * 256, nt * 256:nt * 256 + wdt] = True
        n_mfma += (wdt // 64) * 4 * (nk * 4 + (1 if bias_on else 0))           # per wave: blocks x (k-steps x 4 sub-steps [+ the bias product])
        t += wg_per_xcd
    dup = [a for a, c in shared["stored"].items() if c != 1]
    if dup:
        raise SimError(f"{len(dup)} output pieces stored more than once")
    return bf16_to_f32(out), owned, ref, waves, n_mfma


def check(M, N, K, n_wg=8, wg=0, order=(0, 1, 2, 3), seed=0, gen=None, cfg=None, rows_per_batch=256):
    r = simulate(M, N, K, n_wg, wg, seed, order, gen, cfg, rows_per_batch)
    if r is None:
        return None
    out, owned, ref, waves, n_mfma = r
    if np.isnan(out[owned]).any():
        raise SimError(f"{int(np.isnan(out[owned]).sum())} owned outputs were never written")
    if not np.isnan(out[~owned]).all():
        raise SimError("outputs outside this workgroup's tiles were written")
    got, want = out[owned].astype(np.float64), ref[owned]
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    if not err < 3e-3:
        raise SimError(f"rel err {err:.3e}")
    for wv in waves:
        if wv.stats["mfma"] != n_mfma:
            raise SimError(f"wave {wv.wid}: {wv.stats['mfma']} MFMAs, expected {n_mfma}")
    return err, waves[0].stats


CASES = [
    # (M, N, K, n_wg, wg, order, cfg, rows_per_batch)
    (256, 256, 192, 8, 0, (0, 1, 2, 3), {}, 256),                                  # one tile, three k-steps
    (1024, 768, 192, 8, 0, (3, 2, 1, 0), {}, 256),                                 # two tiles, odd k-step count: stage parity alternates
    (1024, 768, 320, 8, 1, (0, 1, 2, 3), {}, 256),                                 # tile list wraps to the next m-tile; plain NORMAL steps
    (4096, 512, 192, 16, 9, (0, 1, 2, 3), {}, 256),                                # two workgroups per XCD: tile stride 2
    (512, 640, 256, 8, 0, (0, 1, 2, 3), dict(narrow=True), 256),                   # N = 640: wide, wide, narrow per m-tile; this one: wide
    (512, 640, 256, 8, 2, (1, 3, 0, 2), dict(narrow=True), 256),                   # ... the narrow tile of m-tile 0 (first tile narrow)
    (1024, 640, 192, 8, 1, (0, 1, 2, 3), dict(narrow=True), 256),                  # wide -> narrow inside one workgroup (chunk of 2)
    (1024, 640, 192, 8, 2, (3, 2, 1, 0), dict(narrow=True), 256),                  # narrow -> wide ... 
    (1024, 384, 320, 8, 0, (0, 1, 2, 3), dict(narrow=True, res=True), 512),        # gated residual, samples of 512 rows
    (1024, 640, 192, 8, 1, (2, 0, 3, 1), dict(narrow=True, res=True), 256),        # ... across a wide -> narrow change
    (512, 640, 256, 8, 2, (0, 1, 2, 3), dict(narrow=True, res=True, bias=True), 256),   # + bias (to_out)
    (1024, 640, 192, 8, 1, (0, 1, 2, 3), dict(narrow=True, res=True, bias=True), 512),
]


def main():
    import time
    for (M, N, K, n_wg, wg, order, cfg, rpb) in CASES:
        t0 = time.time()
        r = check(M, N, K, n_wg, wg, order, cfg=cfg, rows_per_batch=rpb)
        print(f"M={M} N={N} K={K} wg {wg}/{n_wg} {cfg} rpb {rpb}: {r and (f'rel err {r[0]:.2e}', r[1])}  ({time.time() - t0:.1f} s)")


if __name__ == "__main__":
    main()
