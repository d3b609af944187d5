"""a12: the batched Fq-sponge transcript tape vs the Python restatement of mina-poseidon's DefaultFqSponge
(oracle/ipa_ref.FqSponge): absorb_fq / absorb_g (incl. infinity) / absorb_fr (both modulus orders) / challenge /
challenge_fq / endo challenge / digest, fresh and resumed sponges, final state hand-over."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ABS_FQ, ABS_G, ABS_FR, CHAL, CHAL_FQ, CHAL_ENDO, DIGEST, CHAL_ENDO_OWN = range(8)


def run_reference(oracle, I, R, curve, pp, tape, inputs_ints, init=None):
    sp = I.FqSponge(curve, pp) if init is None else I.FqSponge(curve, pp, init[0], "squeezed" if init[1] else "absorbed", init[2])
    r = R.scalar_modulus(curve)
    it = iter(inputs_ints)
    outs = []
    for op in tape:
        if op == ABS_FQ:
            sp.absorb_fq([next(it)])
        elif op == ABS_G:
            x, y = next(it), next(it)
            sp.absorb_g([None if (x == 0 and y == 0) else (x, y)])
        elif op == ABS_FR:
            sp.absorb_fr([next(it)])
        elif op == CHAL:
            outs.append(sp.challenge())
        elif op == CHAL_FQ:
            outs.append(sp.challenge_fq())
        elif op == CHAL_ENDO:
            outs.append(R.challenge_to_field(sp.challenge(), R.endo_r(curve), r))
        elif op == CHAL_ENDO_OWN:
            # Fr-sponge: the sponge's field is the scalar field of the OTHER curve; its endo_r lives in that field
            other = 1 - curve
            outs.append(R.challenge_to_field(sp.challenge(), R.endo_r(other), R.scalar_modulus(other)))
        else:
            d = sp.challenge_fq()
            outs.append(d if d < r else 0)
    return outs, sp


@pytest.mark.parametrize("curve", [0, 1])
def test_tape_matches_restatement(ctx, oracle, srs_oracle, curve):
    from ipa_helpers import poseidon_pp
    from oracle import ipa_ref as I, pasta_ref as R
    pp = poseidon_pp(curve)
    q, r = R.base_modulus(curve), R.scalar_modulus(curve)
    g, _ = srs_oracle[curve]
    rng = random.Random(100 + curve)
    tape = [ABS_G, ABS_G, CHAL, ABS_FR, CHAL_ENDO, ABS_FQ, ABS_FQ, ABS_FQ, CHAL_FQ, CHAL, ABS_G, DIGEST, ABS_FR, ABS_FR, CHAL_ENDO, CHAL_FQ, CHAL_FQ, CHAL_FQ, ABS_FQ, DIGEST, ABS_FQ, ABS_FQ, CHAL_ENDO_OWN, CHAL_ENDO_OWN]
    batch = 9
    all_inputs, per_proof = [], []
    for b in range(batch):
        ints = []
        for op in tape:
            if op == ABS_FQ:
                ints.append(rng.randrange(q))
            elif op == ABS_G:
                if b == 3 and len(ints) < 2:
                    ints += [0, 0]                                      # infinity as the first point of proof 3
                else:
                    x, y = oracle.bytes_to_point(g[rng.randrange(1000)]); ints += [x, y]
            elif op == ABS_FR:
                ints.append(rng.choice([0, 1, r - 1, rng.randrange(r)]))
        per_proof.append(ints)
        all_inputs.append(oracle.ints_to_le(ints).reshape(-1))
    got, fstate, fpos = ctx.fq_sponge_run(curve, batch, bytes(tape), np.concatenate(all_inputs), want_final=True)
    for b in range(batch):
        exp, sp = run_reference(oracle, I, R, curve, pp, tape, per_proof[b])
        assert [oracle.le_to_int(x) for x in got[b]] == exp, b
        st, mode, count = sp.raw()
        assert [oracle.le_to_int(fstate[b][32 * i: 32 * i + 32]) for i in range(3)] == st and fpos[b].tolist() == [mode, count]
    # the same transcripts tiled past 8192 sponges: the wave-packed 3-lane form gives the same outputs, final states and positions
    reps = 8200 // batch + 1
    got_big, fs_big, fp_big = ctx.fq_sponge_run(curve, batch * reps, bytes(tape), np.tile(np.concatenate(all_inputs), reps), want_final=True)
    assert (got_big.reshape(reps, batch, -1) == got.reshape(1, batch, -1)).all()
    assert (fs_big.reshape(reps, batch, 96) == fstate.reshape(1, batch, 96)).all() and (fp_big.reshape(reps, batch, 2) == fpos.reshape(1, batch, 2)).all()
    # resume from the handed-over state: continuing on the GPU == continuing in the restatement
    tape2 = [ABS_FQ, CHAL, ABS_G, CHAL_FQ]
    ins2, pp2 = [], []
    for b in range(batch):
        x, y = oracle.bytes_to_point(g[b]); ints = [rng.randrange(q), x, y]
        pp2.append(ints); ins2.append(oracle.ints_to_le(ints).reshape(-1))
    got2 = ctx.fq_sponge_run(curve, batch, bytes(tape2), np.concatenate(ins2), init=(fstate, fpos))
    for b in range(batch):
        _, sp = run_reference(oracle, I, R, curve, pp, tape, per_proof[b])
        exp2, _ = run_reference(oracle, I, R, curve, pp, tape2, pp2[b], init=sp.raw())
        assert [oracle.le_to_int(x) for x in got2[b]] == exp2


def test_tape_rejects_bad_arguments(ctx):
    import mina_bridge_amd as m
    with pytest.raises(m.MinaError):
        ctx.fq_sponge_run(0, 1, bytes([9]), np.zeros(32, np.uint8))              # unknown opcode
    with pytest.raises(m.MinaError):
        ctx.fq_sponge_run(0, 1, bytes([CHAL]), np.zeros(0, np.uint8), init=(np.zeros((1, 96), np.uint8), np.array([[2, 0]], np.uint32)))
