"""CPU checks of the generated main loop of linear4w_kernel (zigma_amd/csrc/gen): the committed .inc is what the generator emits,
and the generated TEXT, executed by the simulator (4 waves, LDS, both memory counters, barrier intervals), computes x @ W^T and keeps
the synchronisation discipline the hardware needs (csrc/gen/linear4w_sim.py lists what is checked)."""
import os
import sys

import pytest

from conftest import ROOT

GEN = os.path.join(ROOT, "zigma_amd", "csrc", "gen")
if GEN not in sys.path:
    sys.path.insert(0, GEN)


def test_committed_inc_is_current(tmp_path):
    import linear4w_gen as G
    out = tmp_path / "body.inc"
    G.emit_inc(str(out))
    assert out.read_text() == open(os.path.join(ROOT, "zigma_amd", "csrc", "linear4w_body.inc")).read(), \
        "run `python zigma_amd/csrc/gen/linear4w_gen.py --emit`"


def _cases():
    import linear4w_sim as S
    return S.CASES


@pytest.mark.parametrize("case", range(12))
def test_generated_loop_in_the_simulator(case):
    """every kernel variant (256-wide tiles only / + 128-wide remainder / + gated residual / + bias), tile lists that change width,
    wrap to the next m-tile, odd and even k-step counts, all four wave orders between barriers"""
    import linear4w_sim as S
    M, N, K, n_wg, wg, order, cfg, rpb = S.CASES[case]
    r = S.check(M, N, K, n_wg, wg, order, cfg=cfg, rows_per_batch=rpb)
    assert r is not None and r[0] < 3e-3


def test_register_ranges_are_aligned():
    """the assembler wants 64-bit scalar pairs on even and 128-bit scalar quads on multiples of four (the simulator does not care)"""
    import re
    import linear4w_gen as G
    for cfg in G.VARIANTS.values():
        lines, _ = G.generate(cfg)
        for ln in lines:
            for a, b in re.findall(r"s\[(\d+):(\d+)\]", ln):
                n = int(b) - int(a) + 1
                assert int(a) % (4 if n == 4 else 2) == 0, ln


def test_simulator_catches_a_missing_wait():
    """the checker is not vacuous: dropping one counted wait from the text makes it fail"""
    import linear4w_gen as G
    import linear4w_sim as S
    lines, T = G.generate()
    start = lines.index("L_first0_w_0_%=:")                                    # (a block this problem executes)
    i = next(k for k in range(start, len(lines)) if lines[k].startswith("s_waitcnt vmcnt(0) lgkmcnt(0)") and "s_barrier" in lines[k + 1])
    broken = lines[:i] + ["s_waitcnt lgkmcnt(0)"] + lines[i + 1:]          # the boundary no longer waits for the direct-to-LDS loads
    with pytest.raises(S.SimError):
        S.check(256, 256, 192, 8, 0, gen=(broken, T))
    j = next(k for k in range(start, len(lines)) if lines[k].startswith("s_waitcnt lgkmcnt(0)"))
    broken = lines[:j] + lines[j + 1:]                                    # fragments used before their reads were waited for
    with pytest.raises(S.SimError):
        S.check(256, 256, 192, 8, 0, gen=(broken, T))
