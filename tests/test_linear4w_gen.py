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


@pytest.mark.parametrize("M,N,K,n_wg,wg,order", [
    (256, 256, 192, 8, 0, (0, 1, 2, 3)),          # one tile, three k-steps: FIRST -> NORMAL-before-last -> LAST
    (1024, 768, 192, 8, 0, (3, 2, 1, 0)),         # two tiles, odd k-step count: the stage parity alternates between tiles
    (1024, 768, 320, 8, 1, (0, 1, 2, 3)),         # tile list that wraps to the next m-tile; plain NORMAL steps
    (512, 256, 256, 8, 1, (1, 3, 0, 2)),          # even k-step count
    (4096, 512, 192, 16, 9, (0, 1, 2, 3)),        # two workgroups per XCD: tile stride 2
])
def test_generated_loop_in_the_simulator(M, N, K, n_wg, wg, order):
    import linear4w_sim as S
    r = S.check(M, N, K, n_wg, wg, order)
    assert r is not None and r[0] < 3e-3


def test_simulator_catches_a_missing_wait():
    """the checker is not vacuous: dropping one counted wait from the text makes it fail"""
    import linear4w_gen as G
    import linear4w_sim as S
    lines, T = G.generate()
    start = lines.index("L_first0_%=:")                                    # (a block this problem executes)
    i = next(k for k in range(start, len(lines)) if lines[k].startswith("s_waitcnt vmcnt(0) lgkmcnt(0)") and "s_barrier" in lines[k + 1])
    broken = lines[:i] + ["s_waitcnt lgkmcnt(0)"] + lines[i + 1:]          # the boundary no longer waits for the direct-to-LDS loads
    with pytest.raises(S.SimError):
        S.check(256, 256, 192, 8, 0, gen=(broken, T))
    j = next(k for k in range(start, len(lines)) if lines[k].startswith("s_waitcnt lgkmcnt(0)"))
    broken = lines[:j] + lines[j + 1:]                                    # fragments used before their reads were waited for
    with pytest.raises(S.SimError):
        S.check(256, 256, 192, 8, 0, gen=(broken, T))
