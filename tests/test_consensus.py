"""Samasika chain selection.  The expected values below are the reference README's own worked examples
(README.md:693-705 and img/consensus03.png) plus the decision tables of img/consensus07.png / consensus08.png; random
states are cross-checked against the Python restatement in oracle/consensus_ref.py.  Host-only: no GPU needed."""
import random

import pytest


def mk(m, **kw):
    d = dict(length=100, epoch=5, slot=11 * 7 + 3, min_density=40, window=[5] * 11, staking_cp=b"A" * 32, next_cp=b"B" * 32,
             vrf=b"\x10" * 32, hash=b"\x20" * 32)
    d.update(kw)
    st = m.ConsensusState.make(d["length"], d["epoch"], d["slot"], d["min_density"], d["window"], d["staking_cp"], d["next_cp"], d["vrf"], d["hash"])
    return st, d


def test_projected_window_readme_examples():
    import mina_bridge_amd as m
    # README: current sub-window 11, project to sub-window 15: k = 4 -> shift_count = min(max(4-1,0),11) = 3 zeros,
    # written after the current relative position (ring-shift: overwrite, do not move)
    w = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    st, _ = mk(m, slot=11 * 7, window=w)
    got = m.consensus_project_window(st, 15 * 7)
    cur = 11 % 11
    exp = list(w)
    for j in (1, 2, 3):
        exp[(cur + j) % 11] = 0
    assert got == exp and got.count(0) == 3
    # consensus03.png case 1: same sub-window -> no ring-shift
    assert m.consensus_project_window(st, 11 * 7 + 6) == w
    # case 2: next sub-window -> no zeros shifted in (k - 1 = 0)
    assert m.consensus_project_window(st, 12 * 7) == w
    # case 3: disjoint windows -> the entire window is zeroed
    assert m.consensus_project_window(st, (11 + 12) * 7) == [0] * 11
    assert m.consensus_project_window(st, (11 + 500) * 7) == [0] * 11
    with pytest.raises(m.MinaError):
        m.consensus_project_window(st, 10)                    # into the past


def test_select_longer_and_secure_chain_tables():
    import mina_bridge_amd as m
    tip, _ = mk(m)
    # short range (same epoch, same staking lock checkpoint): consensus08.png
    longer, _ = mk(m, length=101)
    shorter, _ = mk(m, length=99)
    same_hi_vrf, _ = mk(m, vrf=b"\x11" * 32)
    same_lo_vrf, _ = mk(m, vrf=b"\x0f" * 32)
    same_vrf_hi_hash, _ = mk(m, hash=b"\x21" * 32)
    same_vrf_lo_hash, _ = mk(m, hash=b"\x1f" * 32)
    identical, _ = mk(m)
    for cand, exp in ((longer, True), (shorter, False), (same_hi_vrf, True), (same_lo_vrf, False), (same_vrf_hi_hash, True),
                      (same_vrf_lo_hash, False), (identical, False)):
        assert m.consensus_is_short_range(cand, tip)
        assert m.consensus_select_secure_chain(tip, cand) is exp
    # long range (different checkpoints): consensus07.png -- relative minimum window densities decide, ties fall back
    dense, _ = mk(m, staking_cp=b"Z" * 32, min_density=50, window=[9] * 11, length=1)
    sparse, _ = mk(m, staking_cp=b"Z" * 32, min_density=10, window=[9] * 11, length=10 ** 6)
    tie_longer, _ = mk(m, staking_cp=b"Z" * 32, length=101)
    tie_shorter, _ = mk(m, staking_cp=b"Z" * 32, length=99)
    for cand, exp in ((dense, True), (sparse, False), (tie_longer, True), (tie_shorter, False)):
        assert not m.consensus_is_short_range(cand, tip)
        assert m.consensus_select_secure_chain(tip, cand) is exp
    # relative density: a tip that has been offline is projected to the candidate's slot (README "Relative minimum window density")
    stale_tip, _ = mk(m, slot=11 * 7, min_density=55, window=[5] * 11)                      # window density 55 at its own slot
    fresh, _ = mk(m, staking_cp=b"Z" * 32, slot=(11 + 7) * 7, min_density=30, window=[3] * 11)
    assert m.consensus_relative_min_window_density(stale_tip, fresh) == 55 - 6 * 5          # 6 zeros shifted in
    assert m.consensus_relative_min_window_density(fresh, stale_tip) == 30
    assert m.consensus_select_secure_chain(stale_tip, fresh) is True                       # 30 > 25
    # one epoch apart: the later block's previous-epoch checkpoint must equal the earlier block's current-epoch one
    nxt, _ = mk(m, epoch=6, staking_cp=b"B" * 32, next_cp=b"C" * 32)
    assert m.consensus_is_short_range(nxt, tip) and m.consensus_is_short_range(tip, nxt)
    far, _ = mk(m, epoch=8, staking_cp=b"B" * 32)
    assert not m.consensus_is_short_range(far, tip)


def test_random_states_match_restatement():
    import mina_bridge_amd as m
    from oracle import consensus_ref as C
    rng = random.Random(7)
    for _ in range(2000):
        def rnd():
            return dict(length=rng.randrange(90, 110), epoch=rng.randrange(4, 7), slot=rng.randrange(0, 400), min_density=rng.randrange(0, 60),
                        window=[rng.randrange(0, 8) for _ in range(11)], staking_cp=bytes([rng.randrange(65, 68)]) * 32,
                        next_cp=bytes([rng.randrange(65, 68)]) * 32, vrf=bytes([rng.randrange(3)]) * 32, hash=bytes([rng.randrange(3)]) * 32)
        a, b = rnd(), rnd()
        sa, _ = mk(m, **a); sb, _ = mk(m, **b)
        assert m.consensus_is_short_range(sa, sb) == C.is_short_range(a, b)
        assert m.consensus_relative_min_window_density(sa, sb) == C.relative_min_window_density(a, b)
        assert m.consensus_select_secure_chain(sa, sb) == C.select_secure_chain(a, b)
        nxt = max(a["slot"], b["slot"]) + rng.randrange(0, 120)
        assert m.consensus_project_window(sa, nxt) == C.project_window(a, nxt)
