"""PolishToken interpreter semantics (CPU): the one convention for SkipIf / SkipIfNot regions and the cache, against an independent restatement of
upstream's `evaluate` loop."""
import pytest


def test_skipped_tokens_have_no_effect_on_the_cache():
    """kimchi `PolishToken::evaluate` as upstream writes it ([UPSTREAM-RECALL]: `if skip_count > 0 { skip_count -= 1; continue }`, `Store => cache.push(top)`,
    `Load(i) => cache[i]`, `SkipIf(f, n) => if f.is_enabled() { skip_count = n; stack.push(0) }`), restated HERE in a dozen lines independently of
    oracle/kimchi_ref.py, against the oracle's interpreter on programs with a STORE inside a skipped region followed by later STOREs and LOADs --
    the case where "a skipped STORE still takes a slot" (the repo's convention until round 4) and upstream's differ.  CPU-only; the GPU and host
    interpreters are compared with the oracle on such a program in test_verify_boundary.py::test_feature_aware_step_linearization_matches_oracle."""
    import random
    from oracle import kimchi_ref as K
    Q = K.R.Q

    def upstream_evaluate(tokens, features, alpha):
        stack, cache, skip = [], [], 0
        for t in tokens:
            if skip: skip -= 1; continue
            op = t[0]
            if op == K.T_SKIP_IF:
                if (features >> t[1]) & 1: skip = t[2]; stack.append(0)
            elif op == K.T_SKIP_IF_NOT:
                if not (features >> t[1]) & 1: skip = t[2]; stack.append(0)
            elif op == K.T_ALPHA: stack.append(alpha)
            elif op == K.T_LITERAL: stack.append(t[1] % Q)
            elif op == K.T_DUP: stack.append(stack[-1])
            elif op == K.T_ADD: b = stack.pop(); stack.append((stack.pop() + b) % Q)
            elif op == K.T_MUL: b = stack.pop(); stack.append(stack.pop() * b % Q)
            elif op == K.T_SUB: b = stack.pop(); stack.append((stack.pop() - b) % Q)
            elif op == K.T_STORE: cache.append(stack[-1])
            elif op == K.T_LOAD: stack.append(cache[t[1]])            # IndexError = upstream's panic
            else: raise AssertionError(op)
        assert len(stack) == 1
        return stack[0]
    index = type("Ix", (), {"n": 1 << 15, "log2_domain": 15, "zk_rows": 3})()
    rng = random.Random(41)
    region = [(K.T_LITERAL, 7), (K.T_ALPHA,), (K.T_MUL,), (K.T_STORE,), (K.T_LOAD, 1), (K.T_ADD,)]                   # stores INSIDE the region, loads it inside
    prog = [(K.T_ALPHA,), (K.T_STORE,),                                                                            # slot 0
            (K.T_SKIP_IF_NOT, 6, len(region))] + region + [(K.T_ADD,),
            (K.T_LITERAL, 1000), (K.T_STORE,), (K.T_ADD,),                                                         # slot 1 if the region was skipped, 2 if it ran
            (K.T_LOAD, 1), (K.T_MUL,), (K.T_LOAD, 0), (K.T_SUB,)]
    for features in (0, 1 << 6, (1 << 6) | 3):
        for _ in range(4):
            alpha = rng.randrange(Q)
            want = upstream_evaluate(prog, features, alpha)
            got = K.polish_evaluate(prog, index, 5, [], {"alpha": alpha, "features": features}, Q)
            assert got == want, (features, alpha)
    on, off = upstream_evaluate(prog, 1 << 6, 3), upstream_evaluate(prog, 0, 3)
    assert on == ((3 + 21 + 21 + 1000) * 21 - 3) % Q and off == ((3 + 0 + 1000) * 1000 - 3) % Q, "LOAD 1 reads the region's value when it ran, the outer one when it was skipped"
    # a LOAD of a slot that only a SKIPPED store would have filled: upstream panics, the oracle raises, the product fails that proof
    bad = [(K.T_SKIP_IF_NOT, 6, 2), (K.T_ALPHA,), (K.T_STORE,), (K.T_LOAD, 0), (K.T_ADD,)]
    assert upstream_evaluate(bad, 1 << 6, 9) == 18
    with pytest.raises(IndexError):
        upstream_evaluate(bad, 0, 9)
    with pytest.raises(KeyError):
        K.polish_evaluate(bad, index, 5, [], {"alpha": 9, "features": 0}, Q)
