#!/usr/bin/env python3
"""Generator of the main loop of linear_ws_asm_kernel (csrc/linear_ws.hip): the weight-stationary projection for k = 640 as ONE asm
statement, every instruction placed by hand.

Same algorithm, LDS layout and results as the C++ form (linear_ws_kernel<40, 2>; tests compare them bit for bit).  What the C++ form
cannot have: with a single wave per SIMD everything that is not an MFMA has to sit in the gaps BETWEEN the four MFMAs of a k-group (a gap
hides about five single-issue instructions) — hipcc's scheduler places its own address arithmetic, accumulator reads and waits where
it likes, and its temporaries cost the ~20 registers that a second pair of fragment buffers (reads TWO k-groups ahead) needs.

One k-group (qg = 8 ks + q of a 64-token tile, 40 per tile):
    [q == 4 of every odd slice: s_waitcnt vmcnt(12); s_barrier]            slices g + 1, g + 2 landed; slots of g - 2, g - 1 free
    s_waitcnt lgkmcnt(n)                                                   this group's B fragments (requested two groups ago)
    MFMA (tb 0, fb 0)
      gap A: address + 2 ds_read_b128: the fragments of group qg + 2 -> buffer (qg + 2) % 4
    MFMA (tb 0, fb 1)
      gap B: epilogue of the PREVIOUS tile: store of the piece read back one group ago + read-back of the next | 4 accumulator reads
    MFMA (tb 1, fb 0)
      gap C: 2 x v_cvt_pk_bf16_f32 + address + ds_write_b64 of those four | a direct-to-LDS load
    MFMA (tb 1, fb 1)
      gap D: a direct-to-LDS load
Iteration t of the tile loop computes tile t into accumulator set t & 1 and carries the epilogue of tile t - 1 (iteration 0 stores
garbage to the rows of tile 0, which iteration 1 overwrites: no special first tile); the last tile's epilogue is a tail block.

Registers (fixed; the statement clobbers v16 .. v255, all AGPRs, s36 .. s101):
    a[0:127]   two accumulator sets x 4 blocks (2 tb + fb) x 16        a[128:255]  W fragments f = 2 kg + fb < 32
    v[64:255]  W fragments f >= 32                                      v[32:63]    B fragments: buffer b, token block tb at 32 + 8 b + 4 tb
    v[16:19] read-back piece, v[20:23] accumulator reads, v[24:25] packed pair, v26 .. v28 address temporaries, v29 = scr_r ^ 64
usage: linear_ws_gen.py --emit   (writes ../linear_ws_body.inc; tests/test_linear4w_gen.py checks that the committed text is current)"""
import argparse
import os

HERE = os.path.dirname(os.path.abspath(__file__))
KG, NS, NBUF, PF = 40, 5, 4, 2
NWC, NRC, W0 = 16, 8, 1
R0 = W0 + NWC + 2
QS = 4
STORE_POLICY = " nt"
PROBE = os.environ.get("LWS_PROBE", "")       # timing probes (results wrong): "nodma" = no direct-to-LDS loads inside the loop, "noepi" = no epilogue pieces

# ---- registers -------------------------------------------------------------------------------------------------
V_O, V_D, V_PK, V_TA, V_TB, V_TD, V_SR64 = 16, 20, 24, 26, 27, 28, 29
V_BF, V_W = 32, 64
S_XB, S_OB, S_XP, S_OP, S_NT, S_T, S_G, S_DMA = 36, 38, 40, 41, 42, 43, 44, 45
S_SB, S_SB1, S_SRC, S_OT, S_DST, S_TMP = 46, 47, 48, 50, 52, 54          # S_TMP .. S_TMP + 3
S_LT, S_LK, S_LG, S_XP4, S_XP8, S_XP12, S_TXP, S_TOP = 58, 59, 60, 61, 62, 63, 64, 65

OPERANDS = [("v", "a_off"), ("v", "scr_w"), ("v", "sw_w"), ("v", "scr_r"), ("v", "lane_out"), ("v", "voff0"), ("v", "wp0"), ("v", "wp1"),
            ("s", "xb"), ("s", "ob"), ("s", "x_pitch"), ("s", "o_pitch"), ("s", "my_tiles"), ("s", "dma_lds")]
OP = {n: i for i, (_, n) in enumerate(OPERANDS)}


def o(name):
    return f"%{OP[name]}"


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def s(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


def acc(st, blk):
    return (st * 4 + blk) * 16


def wfrag(f):
    return a(128 + 4 * f, 4) if f < 32 else v(V_W + 4 * (f - 32), 4)


def bfrag(b, tb):
    return V_BF + 8 * b + 4 * tb


def e_lds(qg):
    """number of LDS instructions of the epilogue issued in k-group qg (behind its fragment reads)"""
    c, r = qg - W0, qg - R0
    return (1 if 0 <= c < NWC else 0) + (1 if 0 <= r < NRC else 0)


def dma_load(L, i):
    """load i (rows 16 wave + 4 i ..) of the slice the load cursor points at: S_SRC, S_LG"""
    L += [f"s_and_b32 {s(S_TMP)}, {s(S_LG)}, 7", f"s_lshl_b32 {s(S_TMP)}, {s(S_TMP)}, 14", f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {s(S_DMA)}"]
    if i:
        L += [f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, {i * 1024}", f"v_xor_b32 {v(V_TD)}, {i << 6}, {o('voff0')}",
              f"v_add_u32 {v(V_TD)}, {s((S_XP4, S_XP8, S_XP12)[i - 1])}, {v(V_TD)}"]
    L += [f"s_mov_b32 m0, {s(S_TMP)}", "s_nop 0", f"global_load_lds_dwordx4 {v(V_TD) if i else o('voff0')}, {s(S_SRC, 2)}"]
    if i == 3:          # cursor -> next slice (tile S_LT, k-slice S_LK); past the last tile: stay on it (a harmless refill keeps the counts uniform)
        L += [f"s_add_u32 {s(S_LG)}, {s(S_LG)}, 1", f"s_add_u32 {s(S_LK)}, {s(S_LK)}, 1",
              f"s_add_u32 {s(S_SRC)}, {s(S_SRC)}, 256", f"s_addc_u32 {s(S_SRC + 1)}, {s(S_SRC + 1)}, 0",
              f"s_cmp_lt_u32 {s(S_LK)}, {NS}", "s_cbranch_scc1 L_cur_%=_{uid}",
              f"s_mov_b32 {s(S_LK)}, 0", f"s_sub_u32 {s(S_SRC)}, {s(S_SRC)}, {NS * 256}", f"s_subb_u32 {s(S_SRC + 1)}, {s(S_SRC + 1)}, 0",
              f"s_add_u32 {s(S_LT)}, {s(S_LT)}, 1", f"s_cmp_ge_u32 {s(S_LT)}, {s(S_NT)}", "s_cbranch_scc0 L_adv_%=_{uid}",
              f"s_sub_u32 {s(S_LT)}, {s(S_LT)}, 1", "s_branch L_cur_%=_{uid}",
              "L_adv_%=_{uid}:", f"s_add_u32 {s(S_SRC)}, {s(S_SRC)}, {s(S_TXP)}", f"s_addc_u32 {s(S_SRC + 1)}, {s(S_SRC + 1)}, 0",
              "L_cur_%=_{uid}:"]


UID = [0]


def emit_dma(L, i):
    tmp = []
    dma_load(tmp, i)
    UID[0] += 1
    L += [t.replace("{uid}", str(UID[0])) for t in tmp]


def wr_part1(L, st, c):
    b, q4 = c >> 2, c & 3
    for i in range(4):
        L.append(f"v_accvgpr_read_b32 {v(V_D + i)}, {a(acc(st, b) + 4 * q4 + i)}")


def wr_part2(L, c):
    b, q4 = c >> 2, c & 3
    tb, fb = b >> 1, b & 1
    L += [f"v_cvt_pk_bf16_f32 {v(V_PK)}, {v(V_D)}, {v(V_D + 1)}", f"v_cvt_pk_bf16_f32 {v(V_PK + 1)}, {v(V_D + 2)}, {v(V_D + 3)}",
          f"v_xor_b32 {v(V_TB)}, {(fb * 4 + q4) << 4}, {o('sw_w')}", f"v_add_u32 {v(V_TB)}, {v(V_TB)}, {o('scr_w')}",
          f"ds_write_b64 {v(V_TB)}, {v(V_PK, 2)}" + (" offset:4096" if tb else "")]


def rd_chunk(L, r):
    L.append(f"ds_read_b128 {v(V_O, 4)}, {v(V_SR64) if r & 1 else o('scr_r')} offset:{1024 * r}")


def st_chunk(L, r):
    if r:
        L += [f"s_mul_i32 {s(S_TMP + 1)}, {s(S_OP)}, {8 * r}",
              f"s_add_u32 {s(S_DST)}, {s(S_OT)}, {s(S_TMP + 1)}", f"s_addc_u32 {s(S_DST + 1)}, {s(S_OT + 1)}, 0"]
    else:
        L.append(f"s_mov_b64 {s(S_DST, 2)}, {s(S_OT, 2)}")
    # (s_nop: a store of more than 8 bytes must not be followed directly by a write of its data registers)
    L += [f"global_store_dwordx4 {o('lane_out')}, {v(V_O, 4)}, {s(S_DST, 2)}{STORE_POLICY}", "s_nop 1"]


def frag_reads(L, qn, b):
    """fragments of the k-group qn (0 .. 9: 8, 9 = groups 0, 1 of the next slice) -> buffer b"""
    x = (qn & 7) << 5
    if x:
        L += [f"v_xor_b32 {v(V_TA)}, {x}, {o('a_off')}", f"v_add_u32 {v(V_TA)}, {s(S_SB if qn < 8 else S_SB1)}, {v(V_TA)}"]
    else:
        L.append(f"v_add_u32 {v(V_TA)}, {s(S_SB if qn < 8 else S_SB1)}, {o('a_off')}")
    L += [f"ds_read_b128 {v(bfrag(b, 0), 4)}, {v(V_TA)}", f"ds_read_b128 {v(bfrag(b, 1), 4)}, {v(V_TA)} offset:8192"]


def tile_body(par):
    L = []
    for ks in range(NS):
        sync = ((par + ks) & 1) == 1          # global slice index g = 5 t + ks: odd
        # slot bases of this slice and the next one
        L += [f"s_add_u32 {s(S_TMP)}, {s(S_G)}, {ks}", f"s_and_b32 {s(S_SB)}, {s(S_TMP)}, 7", f"s_lshl_b32 {s(S_SB)}, {s(S_SB)}, 14",
              f"s_add_u32 {s(S_TMP)}, {s(S_TMP)}, 1", f"s_and_b32 {s(S_SB1)}, {s(S_TMP)}, 7", f"s_lshl_b32 {s(S_SB1)}, {s(S_SB1)}, 14"]
        for q in range(8):
            qg = ks * 8 + q
            c, r = qg - W0, qg - R0
            if q == QS and sync:
                L += ["s_waitcnt vmcnt(12)", "s_barrier"]
            n_top = e_lds(qg - 2) + 2 + e_lds(qg - 1) if qg >= 2 else (2 + e_lds(qg - 1) if qg == 1 else 2)
            L.append(f"s_waitcnt lgkmcnt({n_top})")
            bcur, bnew = qg % NBUF, (qg + PF) % NBUF
            srcc = lambda blk: "0" if qg == 0 else a(acc(par, blk), 16)
            mf = lambda blk, tb, fb: L.append(f"v_mfma_f32_32x32x16_bf16 {a(acc(par, blk), 16)}, {wfrag(2 * qg + fb)}, {v(bfrag(bcur, tb), 4)}, {srcc(blk)}")
            mf(0, 0, 0)
            frag_reads(L, q + PF, bnew)                                    # gap A
            mf(1, 0, 1)
            if 0 <= r - 1 < NRC:                                           # gap B
                L.append("s_waitcnt lgkmcnt(2)")                            # behind the piece read one group ago: the two fragment reads of gap A
                st_chunk(L, r - 1)
            if 0 <= r < NRC:
                rd_chunk(L, r)
            if 0 <= c < NWC:
                wr_part1(L, par ^ 1, c)
            mf(2, 1, 0)
            if 0 <= c < NWC:                                               # gap C
                wr_part2(L, c)
            dl = (q - QS) * 2
            if sync and q >= QS and PROBE != "nodma":
                emit_dma(L, dl & 3)
            mf(3, 1, 1)
            if sync and q >= QS and PROBE != "nodma":                      # gap D
                emit_dma(L, (dl + 1) & 3)
    return L


def tail_epilogue(st):
    L = ["s_nop 15", "s_nop 15"]               # the last MFMAs' results are read by VALU next
    for c in range(NWC):
        wr_part1(L, st, c)
        wr_part2(L, c)
    for r in range(NRC):
        rd_chunk(L, r)
        L.append("s_waitcnt lgkmcnt(0)")
        st_chunk(L, r)
    return L


def generate():
    L = []
    # ---- prologue: scalars, the first seven slices, the weights, the first fragments
    L += [f"s_mov_b64 {s(S_XB, 2)}, {o('xb')}", f"s_mov_b64 {s(S_OB, 2)}, {o('ob')}", f"s_mov_b32 {s(S_XP)}, {o('x_pitch')}",
          f"s_mov_b32 {s(S_OP)}, {o('o_pitch')}", f"s_mov_b32 {s(S_NT)}, {o('my_tiles')}",
          f"s_mov_b32 {s(S_DMA)}, {o('dma_lds')}", f"s_mov_b32 {s(S_T)}, 0", f"s_mov_b32 {s(S_G)}, 0",
          f"s_lshl_b32 {s(S_XP4)}, {s(S_XP)}, 2", f"s_lshl_b32 {s(S_XP8)}, {s(S_XP)}, 3", f"s_mul_i32 {s(S_XP12)}, {s(S_XP)}, 12",
          f"s_lshl_b32 {s(S_TXP)}, {s(S_XP)}, 6", f"s_lshl_b32 {s(S_TOP)}, {s(S_OP)}, 6",
          f"s_mov_b64 {s(S_SRC, 2)}, {s(S_XB, 2)}", f"s_mov_b64 {s(S_OT, 2)}, {s(S_OB, 2)}",
          f"s_mov_b32 {s(S_LT)}, 0", f"s_mov_b32 {s(S_LK)}, 0", f"s_mov_b32 {s(S_LG)}, 0",
          f"v_xor_b32 {v(V_SR64)}, 64, {o('scr_r')}"]
    for _ in range(7):
        for i in range(4):
            emit_dma(L, i)
    for f in range(2 * KG):          # W fragment f = 2 kg + fb <- rows of feature block fb, 32 kg bytes in
        L.append(f"global_load_dwordx4 {wfrag(f)}, {o('wp1') if f & 1 else o('wp0')}, off offset:{(f >> 1) * 32}")
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    # fragments of groups 0 and 1 of slice 0 (ring slot 0)
    L += [f"ds_read_b128 {v(bfrag(0, 0), 4)}, {o('a_off')}", f"ds_read_b128 {v(bfrag(0, 1), 4)}, {o('a_off')} offset:8192",
          f"v_xor_b32 {v(V_TA)}, 32, {o('a_off')}", f"ds_read_b128 {v(bfrag(1, 0), 4)}, {v(V_TA)}", f"ds_read_b128 {v(bfrag(1, 1), 4)}, {v(V_TA)} offset:8192"]

    def end_of_iteration(tag, set_done):
        return [f"s_cmp_eq_u32 {s(S_T)}, 0", f"s_cbranch_scc1 L_noadv{tag}_%=",
                f"s_add_u32 {s(S_OT)}, {s(S_OT)}, {s(S_TOP)}", f"s_addc_u32 {s(S_OT + 1)}, {s(S_OT + 1)}, 0", f"L_noadv{tag}_%=:",
                f"s_add_u32 {s(S_T)}, {s(S_T)}, 1", f"s_add_u32 {s(S_G)}, {s(S_G)}, {NS}",
                f"s_cmp_ge_u32 {s(S_T)}, {s(S_NT)}", f"s_cbranch_scc1 L_tail{set_done}_%="]
    L.append("L_loop_%=:")
    L += tile_body(0) + end_of_iteration("a", 0)
    L += tile_body(1) + end_of_iteration("b", 1)
    L.append("s_branch L_loop_%=")
    L.append("L_tail0_%=:")
    L += tail_epilogue(0) + ["s_branch L_end_%="]
    L.append("L_tail1_%=:")
    L += tail_epilogue(1)
    L += ["L_end_%=:", "s_waitcnt vmcnt(0)"]
    return L


def clobbers():
    return ["memory", "scc", "vcc"] + [f"v{i}" for i in range(16, 256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(36, 70)]


def emit_inc(path):
    lines = generate()
    with open(path, "w") as fh:
        fh.write("// GENERATED by zigma_amd/csrc/gen/linear_ws_gen.py --emit — do not edit (tests/test_linear4w_gen.py checks it is current).\n")
        fh.write("// The whole of linear_ws_asm_kernel behind its operand set-up as one asm statement; operands in the order of OPERANDS in the generator.\n")
        fh.write("#define ZIGMA_LINEAR_WS_BODY \\\n")
        for ln in lines:
            fh.write(f'    "{ln}\\n" \\\n')
        fh.write('    ""\n')
        fh.write("#define ZIGMA_LINEAR_WS_OPERANDS(" + ", ".join(n for _, n in OPERANDS) + ") \\\n    " +
                 ", ".join(f'"{c}"({n})' for c, n in OPERANDS) + "\n")
        cl = clobbers()
        fh.write("#define ZIGMA_LINEAR_WS_CLOBBERS \\\n")
        for i in range(0, len(cl), 16):
            fh.write("    " + ", ".join(f'"{c}"' for c in cl[i:i + 16]) + (", \\\n" if i + 16 < len(cl) else "\n"))
    return len(lines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--emit", action="store_true")
    ap.add_argument("--stats", action="store_true")
    args = ap.parse_args()
    if args.emit:
        print(emit_inc(os.path.join(HERE, "linear_ws_body.inc")), "instructions")
    if args.stats:
        L = generate()
        print(len(L), "lines,", sum("v_mfma" in l for l in L), "MFMAs")
