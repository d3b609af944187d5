"""CPU-side pin of tests/golden/statement_k15.json (the full-size input of bench.py's default mode) against oracle/pickles_ref.py"""


def test_statement_fixture_is_what_the_oracle_derives():
    """CPU: the committed statements pack (oracle/pickles_ref.py) to the committed public inputs, and the fixture names the constant set in use"""
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import load_k15_fixture, load_statement_fixture, make_step_index
    from oracle import pickles_ref as PK
    import mina_bridge_amd.poseidon_params as PP
    ix, _, _ = load_k15_fixture()
    items, fx = load_statement_fixture()
    assert fx["poseidon_constants"] == PP.NAME and len(items) >= 4
    comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
    step = make_step_index(99)
    for it in items[:2]:
        assert PK.statement_public_input(it["wrap"], step, comms, it["app"], poseidon_pp(0), poseidon_pp(1))[0] == it["pubs"]


def test_encoded_fixture_is_the_helpers_encoding_of_the_fixture():
    """tests/golden/statement_k15_encoded.json (what bench.py's default mode reads, so that it needs nothing under oracle/) is exactly the
    C-ABI encoding the test helpers produce from statement_k15.json / kimchi_k15.json / make_step_index(99)"""
    import importlib.util, json, os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("encode_statement_fixture", os.path.join(here, "golden", "encode_statement_fixture.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.encode() == json.load(open(os.path.join(here, "golden", "statement_k15_encoded.json")))
