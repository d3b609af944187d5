"""One context per host thread, all on GPU 0 (the integration's threading model): concurrent calls from
different contexts stay bit-exact -- nothing is shared between contexts but the device."""
import threading

import numpy as np
import pytest

from conftest import rand_scalars

pytestmark = pytest.mark.gpu
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


def test_concurrent_contexts_are_bit_exact(oracle, srs_oracle):
    import mina_bridge_amd as m
    T, rounds = 4, 6
    errors, results = [], {}

    def worker(t):
        try:
            curve = t % 2
            mod = Q if curve == 0 else P
            g, _ = srs_oracle[curve]
            c = m.MinaContext(0)
            try:
                c.poseidon_set_params(curve, m.poseidon_params.default_params_bytes(curve))
                c.srs_create(curve, 65536)
                if t >= 2:
                    c.set_pipeline(3)
                for r in range(rounds):
                    n = [65536, 1000, 30000][r % 3]
                    sc = rand_scalars(n, mod, seed=10000 + 97 * t + r)
                    results[(t, r, "fixed")] = (c.msm_srs(curve, sc), curve, n, sc)
                    nv = [300, 5000][r % 2]
                    sv = rand_scalars(nv, mod, seed=15000 + 97 * t + r)
                    results[(t, r, "var")] = (c.msm(curve, g[7:7 + nv], sv), curve, nv, sv)
                    st = rand_scalars(3 * 40, P if curve == 0 else Q, seed=20000 + 31 * t + r).reshape(40, 96)      # field `curve`: 0 = Fp, 1 = Fq
                    results[(t, r, "perm")] = (c.poseidon_permute(curve, st), curve, 40, st)
            finally:
                c.close()
        except Exception as e:                                  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors
    assert len(results) == T * rounds * 3
    for (t, r, kind), (got, curve, n, inp) in results.items():
        g, _ = srs_oracle[curve]
        if kind == "fixed":
            exp = oracle.msm_pippenger(curve, g[:n], inp, threads=8)
        elif kind == "var":
            exp = oracle.msm_pippenger(curve, g[7:7 + n], inp, threads=8)
        else:
            exp = oracle.poseidon_permute(curve, poseidon_pp_bytes(curve), inp)
        assert (np.asarray(got).reshape(-1) == np.asarray(exp).reshape(-1)).all(), (t, r, kind)


def poseidon_pp_bytes(field):
    import mina_bridge_amd as m
    return m.poseidon_params.default_params_bytes(field)
