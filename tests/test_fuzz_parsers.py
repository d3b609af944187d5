"""The untrusted-bytes surface under sanitizers (SURVEY.md 5 "race detection / sanitizers"; the proof and public-input bytes cross a network
boundary: core/src/aligned.rs:31-58).  tests/fuzz/fuzz_parsers.cpp builds the library's own host readers -- wire_proof.h, wire_state.h,
wire_account.h, wire_pub.h, loaders_text.h, the headers libminaverify.so compiles -- standalone with -fsanitize=address,undefined:
  * `sweep_parsers` (g++): every seed, every truncation, bit flips and 0xff bytes across it, through every reader the seed is meant for;
  * `fuzz_parsers` (clang++ -fsanitize=fuzzer): a coverage-guided libFuzzer run over all readers, seeded with the same corpus.
Seeds: the committed golden byte fixtures + containers written by the independent Python writers.  Done = no sanitizer report, no assertion."""
import base64
import json
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = os.path.join(ROOT, "tests", "fuzz")
# tests/fuzz/fuzz_parsers.cpp `enum Target`
T = dict(wrap_bincode=0, wrap_binprot=1, state_proof=2, pstate_bincode=3, pstate_binprot=4, account_bincode=5, account_binprot=6, account_proof=7,
         state_pub=8, account_pub=9, poseidon_text=10, tokens_json=11, index_json=12)
SAN_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", FUZZ, "-s", "all"])
    return os.path.join(FUZZ, "sweep_parsers"), os.path.join(FUZZ, "fuzz_parsers")


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    """seed files `<target>_<name>` (sweep_parsers reads the target from the name) + the same with the target byte in front (libFuzzer corpus)"""
    import mina_bridge_amd.poseidon_params as PP
    from oracle import mina_account_ref as A
    from test_protocol_state import bincode_state
    from wire_writers import synth_wrap_proof, wrap_proof_bytes
    d = tmp_path_factory.mktemp("fuzz_corpus")
    sweep, fuzz = d / "sweep", d / "fuzz"
    sweep.mkdir(); fuzz.mkdir()
    seeds = []

    def add(target, name, data: bytes):
        (sweep / f"{T[target]}_{name}").write_bytes(data)
        (fuzz / f"{target}_{name}").write_bytes(bytes([T[target]]) + data)
        seeds.append(str(sweep / f"{T[target]}_{name}"))
    fxb = json.load(open(os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")))
    proof, pub = bytes.fromhex(fxb["proofs"][0]["proof"]), bytes.fromhex(fxb["proofs"][0]["pub"])
    add("state_proof", "k15", proof)                              # the committed full-size bincode MinaStateProof (40 KB)
    add("state_pub", "k15", pub)
    rng = random.Random(5)
    for lookups in (False, True):
        w = synth_wrap_proof(rng, k=15, lookups=lookups)
        add("wrap_bincode", f"lookups{int(lookups)}", wrap_proof_bytes(w, False))
        add("wrap_binprot", f"lookups{int(lookups)}", wrap_proof_bytes(w, True))
    tip = json.load(open(os.path.join(ROOT, "tests", "golden", "tip_protocol_state.json")))
    raw = base64.b64decode(tip["protocol_state_base64"]) if "protocol_state_base64" in tip else bytes.fromhex(tip["protocol_state_hex"])
    add("pstate_binprot", "tip", raw)                             # the reference's own serialized state (core/src/utils/constants.rs:22)
    from oracle import mina_state_ref as S
    add("pstate_bincode", "tip", bincode_state(S.parse_protocol_state(raw)))
    for i, (zk, timed, deleg, vk) in enumerate([(False, False, False, True), (True, True, True, True), (True, False, True, False)]):
        a = A.synth_account(rng, zk, timed, deleg, with_vk=vk)
        add("account_bincode", f"a{i}", A.write_account(a, False))
        add("account_binprot", f"a{i}", A.write_account(a, True))
        path = [(rng.randrange(2), rng.randrange(1 << 254)) for _ in range(35)]
        add("account_proof", f"a{i}", A.write_account_proof(path, a))
        enc = A.abi_encode_account(a)
        add("account_pub", f"a{i}", rng.randrange(1 << 254).to_bytes(32, "little") + len(enc).to_bytes(8, "little") + enc)
    import numpy as np
    tab = np.frombuffer(PP.default_params_bytes(0), np.uint8).reshape(9 + 165, 32)
    ints = [int.from_bytes(r.tobytes(), "little") for r in tab]
    obj = {"mds": [[str(x) for x in ints[3 * i: 3 * i + 3]] for i in range(3)], "roundConstants": [[str(x) for x in ints[9 + 3 * i: 12 + 3 * i]] for i in range(55)], "fullRounds": 55}
    add("poseidon_text", "o1js", json.dumps(obj).encode())
    rust = "mds: vec![" + ",".join("vec![" + ",".join(f'Fp::from_hex("{x.to_bytes(32, "little").hex()}")' for x in ints[3 * i: 3 * i + 3]) + "]" for i in range(3)) + "],\nround_constants: vec![" + \
           ",".join("vec![" + ",".join(f'Fp::from_str("{x}")' for x in ints[9 + 3 * i: 12 + 3 * i]) + "]" for i in range(55)) + "]"
    add("poseidon_text", "rust", rust.encode())
    lit = (123456789).to_bytes(32, "little").hex()
    prog = ["Alpha", {"Mds": {"row": 0, "col": 1}}, {"Literal": lit}, {"Cell": {"col": {"Witness": 3}, "row": "Curr"}}, "Dup", {"Pow": 7}, "Add", "Mul", "Sub",
            "VanishesOnZeroKnowledgeAndPreviousRows", {"UnnormalizedLagrangeBasis": {"zk_rows": True, "offset": -1}}, "Store", {"Load": 0},
            {"SkipIf": [{"LookupPattern": "Xor"}, 2]}, {"Cell": {"col": {"LookupSorted": 1}, "row": "Next"}}, "Mul", {"SkipIfNot": ["RangeCheck0", 1]}, {"Challenge": "JointCombiner"},
            {"Constant": "EndoCoefficient"}, {"Cell": {"col": {"Index": "Poseidon"}, "row": "Curr"}}]
    add("tokens_json", "mixed", json.dumps(prog).encode())
    pt33 = (bytes(range(1, 33)) + b"\x80").hex()
    comm = lambda: {"elems": [pt33]}
    index = {"domain": ((1 << 15).to_bytes(8, "little") + (15).to_bytes(4, "little") + bytes(160)).hex(), "zk_rows": 3, "shift": [lit] * 7, "sigma_comm": [comm()] * 7,
             "coefficients_comm": [comm()] * 15, "generic_comm": comm(), "psm_comm": comm(), "complete_add_comm": comm(), "mul_comm": comm(), "emul_comm": comm(),
             "endomul_scalar_comm": comm(), "range_check0_comm": None, "lookup_index": None, "max_poly_size": 32768}
    add("index_json", "wrap", json.dumps(index).encode())
    return {"seeds": seeds, "fuzz_dir": str(fuzz)}


def test_truncation_and_bit_flip_sweep_is_clean_under_asan_ubsan(built, corpus):
    sweep, _ = built
    r = subprocess.run([sweep] + corpus["seeds"], capture_output=True, text=True, timeout=1500, env=SAN_ENV)
    assert r.returncode == 0 and "sweep ok" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    print(r.stdout.strip())


def test_libfuzzer_run_over_every_reader_is_clean(built, corpus, tmp_path):
    """coverage-guided: $MINA_FUZZ_SECONDS (default 45) of libFuzzer over all readers at once; a crash leaves its input under tests/fuzz/crashes/"""
    _, fuzz = built
    seconds = int(os.environ.get("MINA_FUZZ_SECONDS", "45"))
    crashes = os.path.join(FUZZ, "crashes"); os.makedirs(crashes, exist_ok=True)
    work = tmp_path / "corpus"; work.mkdir()
    r = subprocess.run([fuzz, str(work), corpus["fuzz_dir"], f"-max_total_time={seconds}", "-max_len=49152", "-timeout=20", "-rss_limit_mb=3072", "-print_final_stats=1",
                        f"-artifact_prefix={crashes}/"], capture_output=True, text=True, timeout=seconds + 600, env=SAN_ENV)
    tail = r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, tail
    import re
    stats = dict(re.findall(r"^stat::(\w+):\s+(\d+)", r.stderr, flags=re.M))
    assert int(stats.get("number_of_executed_units", "0")) > 1000, tail
    print("libFuzzer:", stats)


def test_boundary_host_logic_is_race_free_under_thread_sanitizer():
    """the boundary's HOST logic -- api_verify.hip (slots, chunk pipeline, group commit of small callers, shape vote, device sharding, the
    fallback that drains the device before a culprit search), host_core.hip (worker pool, tuning), the parsers and the fork choice -- built from the product's own
    sources with clang -fsanitize=thread against a stand-in HIP runtime, the device layer stubbed with answers that depend on the job's bytes
    (tests/fuzz/tsan_boundary.cpp), and hammered for 30 s: 6 state callers (single proofs that merge into shared jobs, small batches, multi-chunk calls over two
    logical devices), 3 account callers, an installer re-installing the index and flipping the tuning mid-flight, a bad proof (folded failure -> culprit search)
    or an unparseable one in every fifth call.  Done = no ThreadSanitizer report, every verdict as the stub device dictates, at least one culprit search ran.
    The contract: "callable concurrently ... must be re-entrant" (the reference's callers: /root/reference/README.md:277-279)."""
    subprocess.check_call(["make", "-C", FUZZ, "-s", "tsan"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    r = subprocess.run([os.path.join(FUZZ, "tsan_boundary"), os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json"),
                        os.path.join(ROOT, "tests", "golden", "account_proofs_bytes.json"), "30"], capture_output=True, text=True, timeout=600, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["wrong_verdicts"] == 0 and out["failed_calls"] == 0 and out["culprit_searches"] > 0 and out["device_jobs"] > 0 and out["account_jobs"] > 0 and out["installs"] > 10, out
