"""Big-integer check of tools/probes/fe52_probe: every (a, b, r) triple the probe prints with --values must satisfy
r * 2^260 == a * b (mod p) and r < 2^260, for both Pasta fields (ark-ff 0.3 Fp256 arithmetic, /root/reference/core/Cargo.toml:19-21).
Also carries the exact model of the FMA-pair split the probe relies on (`split52`), checked here against Python integers -- runs without a GPU.
usage: fe52_probe --values | python tools/probes/fe52_check.py"""
import json
import sys
from fractions import Fraction

P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


def rz(x: Fraction, ulp_exp: int) -> int:
    """round a non-negative rational toward zero to a multiple of 2^ulp_exp"""
    return (x.numerator // x.denominator) >> ulp_exp << ulp_exp


def split52(a: int, b: int):
    """the probe's FMA pair in exact arithmetic: hi_raw = RZ(a b + 2^104) has ulp 2^52 (the sum lies in [2^104, 2^105)); lo_raw = RZ(a b + (2^104 + 2^52 - hi_raw))
    lies in [2^52, 2^53): ulp 1, exact.  Returns the two mantissa fields."""
    assert 0 <= a < 1 << 52 and 0 <= b < 1 << 52
    hi_raw = rz(Fraction(a * b + (1 << 104)), 52)
    assert 1 << 104 <= hi_raw < 1 << 105
    sub = (1 << 104) + (1 << 52) - hi_raw                      # a multiple of 2^52 of magnitude < 2^105: representable
    lo_raw = a * b + sub
    assert 1 << 52 <= lo_raw < 1 << 53
    return (hi_raw - (1 << 104)) >> 52, lo_raw - (1 << 52)


def self_test():
    import random
    rnd = random.Random(52)
    for _ in range(20000):
        a, b = rnd.getrandbits(52), rnd.getrandbits(52)
        if _ < 16: a, b = ((1 << 52) - 1 if _ & 1 else 0), ((1 << 52) - 1 if _ & 2 else 1)
        hi, lo = split52(a, b)
        assert hi == (a * b) >> 52 and lo == (a * b) & ((1 << 52) - 1)


if __name__ == "__main__":
    self_test()
    n = bad = 0
    for line in sys.stdin:
        if not line.startswith('{"field"'):
            continue
        d = json.loads(line); n += 1
        m = (P, Q)[d["field"]]
        a, b, r = int(d["a"], 16), int(d["b"], 16), int(d["r"], 16)
        if not (r < 1 << 260 and (r << 260) % m == a * b % m):
            bad += 1
    print(json.dumps({"triples": n, "mismatches": bad}))
    sys.exit(1 if bad or n == 0 else 0)
