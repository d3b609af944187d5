# dev tool: randomised differential soak of the Poseidon entry points (Merkle paths, sponge hashes, raw permutations) against
# the C oracle, fresh seeds, batch sizes on both sides of the kernel-form thresholds.  usage: soak_sponge.py SECONDS
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mina_bridge_amd as m
from oracle import oracle as O, pasta_ref as R
from test_merkle import pp_for
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
MOD = {0: P, 1: Q}
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(time.time()) & 0xffffffff
rng = np.random.Generator(np.random.PCG64(seed))
print("seed", seed)
ctx = m.MinaContext(0)
PB = {f: m.poseidon_params.default_params_bytes(f) for f in (0, 1)}
for f in (0, 1): ctx.poseidon_set_params(f, PB[f])
SALT = {}


def salt(field, h):
    if (field, h) not in SALT: SALT[(field, h)] = R.merkle_salt(h, pp_for(field))
    return SALT[(field, h)]


def fe(n, mod):
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8); a[:, 31] &= 0x3f
    return a


t0 = time.time(); it = 0
while time.time() - t0 < budget:
    it += 1
    field = int(rng.integers(0, 2)); mod = MOD[field]
    what = int(rng.integers(0, 3))
    if what == 0:
        n = int(rng.choice([1, 7, 255, 4096, 8192, 8193, 20000])); depth = int(rng.integers(0, 41))
        leaves, sib = fe(n, mod), fe(max(n * depth, 1), mod)[: n * depth]
        dirs = rng.integers(0, 2, n * depth, dtype=np.uint8)
        got = ctx.merkle_roots(field, leaves, sib, dirs, depth)
        for i in rng.integers(0, n, min(n, 12)):
            node = O.le_to_int(leaves[i])
            for h in range(depth):
                s = O.le_to_int(sib[i * depth + h]); st = list(salt(field, h))
                l, r = (node, s) if dirs[i * depth + h] == 0 else (s, node)
                st[0] = (st[0] + l) % mod; st[1] = (st[1] + r) % mod
                node = O.le_to_int(O.poseidon_permute(field, PB[field], O.ints_to_le(st).reshape(1, 96))[0][:32])
            assert O.le_to_int(got[i]) == node, ("merkle", seed, it, n, depth, int(i))
    elif what == 1:
        n = int(rng.choice([1, 100, 8192, 8193, 30000])); length = int(rng.integers(0, 13))
        inp = fe(max(n * length, 1), mod)[: n * length]
        got = ctx.poseidon_hash(field, inp, n, length)
        for i in rng.integers(0, n, min(n, 20)):
            assert (got[i] == O.poseidon_hash(field, PB[field], inp[i * length:(i + 1) * length])).all(), ("hash", seed, it, n, length, int(i))
    else:
        n = int(rng.choice([1, 63, 64, 65, 5000]))
        st = fe(3 * n, mod).reshape(n, 96)
        assert (ctx.poseidon_permute(field, st) == O.poseidon_permute(field, PB[field], st)).all(), ("perm", seed, it, n)
print(f"sponge soak ok: {it} random cases in {time.time() - t0:.0f}s")
