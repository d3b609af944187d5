"""Poseidon constants for the Kimchi-shaped sponge (width 3, 55 full rounds, x^7) -- a *parameter* of the engine.

The real `fp_kimchi` / `fq_kimchi` tables of mina-poseidon (pin core/Cargo.toml:14) are not in the
reference tree and cannot be fetched here, so the default set below is generated with the published
Hades/Poseidon Grain-LFSR procedure (field=1, sbox=0, n=255, t=3, R_F=55, R_P=0; Cauchy MDS from the
same stream).  It is named "hades-grain-UNPINNED": it has the right shape and cost, it is NOT claimed
to equal Mina's tables.  Install the real tables with `MinaContext.poseidon_set_params` when available.
"""
from __future__ import annotations

from functools import lru_cache

P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
NAME = "hades-grain-UNPINNED"


def _grain(prime: int, t: int = 3, r_f: int = 55, r_p: int = 0):
    n = prime.bit_length()

    def tb(v, w):
        return [int(b) for b in bin(v)[2:].zfill(w)]

    state = tb(1, 2) + tb(0, 4) + tb(n, 12) + tb(t, 12) + tb(r_f, 10) + tb(r_p, 10) + [1] * 30

    def step():
        b = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state.pop(0)
        state.append(b)
        return b

    for _ in range(160):
        step()

    def nextbit():
        while True:
            if step() == 1:
                return step()
            step()

    def rand_bits(k):
        v = 0
        for _ in range(k):
            v = (v << 1) | nextbit()
        return v

    rc = []
    for _ in range((r_f + r_p) * t):
        r = rand_bits(n)
        while r >= prime:
            r = rand_bits(n)
        rc.append(r)
    while True:
        rl = [rand_bits(n) % prime for _ in range(2 * t)]
        if len(set(rl)) != len(rl):
            continue
        xs, ys = rl[:t], rl[t:]
        if any((x + y) % prime == 0 for x in xs for y in ys):
            continue
        mds = [[pow((xs[i] + ys[j]) % prime, prime - 2, prime) for j in range(t)] for i in range(t)]
        return mds, [rc[3 * i: 3 * i + 3] for i in range(r_f)]


@lru_cache(maxsize=None)
def default_params_ints(field: int):
    """(mds 3x3, rc 55x3) as Python ints for field 0 (Fp) / 1 (Fq)."""
    return _grain(P if field == 0 else Q)


@lru_cache(maxsize=None)
def default_params_bytes(field: int) -> bytes:
    mds, rc = default_params_ints(field)
    flat = [mds[i][j] for i in range(3) for j in range(3)] + [rc[r][j] for r in range(55) for j in range(3)]
    return b"".join(x.to_bytes(32, "little") for x in flat)
