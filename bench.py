#!/usr/bin/env python3
"""bench.py -- Mina state proofs verified/sec on the MI355X-native verifier hot path (BASELINE.json metric, config C3).

A "step" = one pass of the Proof-of-State job over one batch of `--jobs` synthetic state proofs, all inputs resident in
HBM, through ONE C-ABI call (`mina_state_job_batch_dev`, no host synchronisation inside).  Per proof, default `--mode full` -- the
whole verifier from the PARSED proof, everything on the GPU:
  * the 17 protocol-state hashes (16 candidate-chain states + bridge tip; `MinaHash` = Poseidon over `to_input`), compared
    with the public inputs, plus the chain linkage                                      (README.md:283-288)
  * the Pickles statement -> the wrap circuit's 40 public inputs: endo-expanded challenges, the Tick sponge over the step proof's
    evaluations (xi, r), ft_eval0 of the step proof, derive_plonk, combined inner product, b, the two message digests, packing
                                                                                        (openmina verify_block / compute_deferred_values)
  * the wrap proof's public-input commitment: those 40 scalars over the 2^15 Pallas Lagrange basis
  * kimchi `oracles` + `to_batch` of the wrap proof: Fq-sponge, Fr-sponge, public polynomial, ft_eval0 with the linearization's
    constant term (PolishToken program), the combined inner product, the chunked ft commitment (handed to the MSM as its 8 terms)
  * the wrap proof's combined IPA opening: k = 15 rounds, 47 commitments x 2 evaluation points, folded over the batch
    into one 2^15 fixed-base MSM + one variable-base MSM                                (kimchi batch_verify / SRS::verify)
  * the step accumulator check: b_poly_coefficients of 16 challenges -> 2^16-base Vesta MSM, folded over the batch
`--mode kimchi` leaves the statement stage out (public inputs given), `--mode prepared` also the kimchi stage (BatchEvaluationProof
rows given: round 2's first headline).  The wrap / step verifier indexes are synthetic at the real sizes (the blockchain-snark
indexes are not offline); proofs are minted by the repo's CPU oracle and ACCEPT.  Steps are issued round-robin over
`--pipeline` lanes (independent batches overlap on the GPU).  Not in the job: bin_prot / bincode parsing of the containers and the
consensus pre-checks (host side of the boundary, done by the caller before the timed region).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--jobs B] [--pipeline L]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: proof-level sharding (SURVEY.md 8e variant 1): every rank verifies its own proofs against its replica of the
SRS tables; the ranks' verdict words are all-gathered over RCCL inside the timed region; weak scaling.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# ROCm exposes 4 hardware queues per process by default; the pipeline lanes of the context need one each to overlap, and the
# process holds a few more streams (the null stream, torch's): with exactly 16 queues two lanes share one and run behind each
# other (measured: 16 -> 24 queues = +5 % at 8192 proofs per call, +19 % at 1024, +31 % at 256).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

CURVE_PALLAS, CURVE_VESTA = 0, 1
FIELD_FP = 0
ACC_K, WRAP_K, LOG2_DOMAIN, NPUB, NCOMMS, NPTS, SLOT = 16, 15, 15, 40, 45, 2, 0
STATES_PER_PROOF, PSTATE_SLOTS, PSTATE_BODY_FIELDS = 17, 64, 49
HBM_PEAK_GBPS = 8000.0                                # MI355X_MICROARCH.md: 8 TB/s spec
# VALU side of the roofline (the path is integer-multiply bound).  The dominant kernel runs its Poseidon rounds on 9 limbs of
# 29 bits (fp29.cuh): per lane and round 2 squarings (99 limb multiply-accumulates each), 2 products (135) and one 3-term dot product (297)
# = 765 `v_mad_u64_u32`, which issue at 7.0 cycles per wave64 instruction (profiles/r02_microbench_valu.jsonl, 8 waves per SIMD); the
# round's other ~290 instructions (shifts, masks) take the remaining issue slots.
MADS_PER_LANE_ROUND = 2 * 99 + 2 * 135 + 297
MAD_ISSUE_CYCLES = 7.0
CHIP_SIMDS, CLOCK_HZ = 1024, 2.4e9
PROF_STAGES = {"pstate_hash": 11, "ipa_transcript": 12, "kimchi_to_batch": 13, "pickles_statement": 14, "msm_accumulate": 3}
# HBM-side bytes per protocol-state hash: read from the tracked summary of the rocprofv3 PMC passes (FETCH_SIZE x 2 -- the gfx950 correction
# of MI355X_MICROARCH.md for 16-B-per-lane loads -- + WRITE_SIZE, KiB, over the states of one launch); null when the file is missing
TRAFFIC_FILE = os.path.join("profiles", "pstate_hash_traffic.json")


def pstate_traffic():
    try:
        t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
        return (2 * t["fetch_size_kib"] + t["write_size_kib"]) * 1024 / t["states_per_launch"], t.get("source", TRAFFIC_FILE)
    except (OSError, KeyError, ValueError):
        return None, None


def le32(x: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(32, "little"), np.uint8)


def fresh_randomiser() -> np.ndarray:
    r = np.frombuffer(os.urandom(32), np.uint8).copy(); r[31] &= 0x3F
    return r


def make_chains(ctx, n_chains: int, seed: int):
    """`n_chains` synthetic candidate chains (16 linked states + bridge tip) as the job takes them: flattened protocol-state records
    (include/mina_verify.h: slot 0 = previous_state_hash, slots 1..49 = the body's `to_input` field elements).  Random field content,
    linked through hashes computed by the GPU path itself, state by state (the bin_prot / bincode readers and the `to_input` flattening
    that produce such records from serialized states are the host side of the boundary: tests/test_protocol_state.py,
    tests/test_verify_fullsize.py).  Returns (records[n,17,2048], nfields[n,17], hashes[n,17,32])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    recs = np.zeros((n_chains, STATES_PER_PROOF, PSTATE_SLOTS, 32), np.uint8)
    nf = np.full((n_chains, STATES_PER_PROOF), PSTATE_BODY_FIELDS, np.uint32)
    hashes = np.zeros((n_chains, STATES_PER_PROOF, 32), np.uint8)
    body = rng.integers(0, 256, size=(n_chains, STATES_PER_PROOF, PSTATE_BODY_FIELDS, 32), dtype=np.uint8)
    body[..., 31] &= 0x3F                                       # < 2^254 < p: canonical
    recs[:, :, 1:1 + PSTATE_BODY_FIELDS] = body
    prev = rng.integers(0, 256, size=(n_chains, 32), dtype=np.uint8); prev[:, 31] &= 0x3F
    for s in range(STATES_PER_PROOF):
        if s == 16:                                             # the bridge tip is not linked to the candidate chain
            prev = rng.integers(0, 256, size=(n_chains, 32), dtype=np.uint8); prev[:, 31] &= 0x3F
        recs[:, s, 0] = prev
        hashes[:, s] = ctx.protocol_state_hash_batch(recs[:, s].reshape(n_chains, -1).copy(), nf[:, s].copy())
        prev = hashes[:, s].copy()
    return recs.reshape(n_chains, STATES_PER_PROOF, PSTATE_SLOTS * 32), nf, hashes


def make_accumulators(ctx, count: int, seed: int):
    rng = np.random.Generator(np.random.PCG64(seed))
    pre = rng.integers(0, 256, size=(count, ACC_K, 16), dtype=np.uint8)
    sgs = np.empty((count, 64), np.uint8)
    for i in range(count):
        sgs[i] = ctx.msm_srs(CURVE_VESTA, ctx.b_poly_coefficients(FIELD_FP, ctx.challenge_to_field(FIELD_FP, pre[i])))
    return pre, sgs


def load_encoded_fixture():
    """tests/golden/statement_k15_encoded.json: the four complete wrap proofs in the C-ABI's byte layouts (written by
    tests/golden/encode_statement_fixture.py from statement_k15.json; nothing under oracle/ is needed to read it)"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "statement_k15_encoded.json")))
    un = lambda h: np.frombuffer(bytes.fromhex(h), np.uint8).copy() if h else np.zeros(1, np.uint8)
    return fx, un


def install_fixture_indexes(ctx, fx, un):
    w, s = fx["wrap_index"], fx["step_index"]
    ctx.verifier_index_install(w["log2_domain"], w["zk_rows"], w["perm_alpha_offset"], un(w["shifts"]), un(w["sigma_comm"]), un(w["coefficients_comm"]),
                               un(w["selector_comm"]), bytes.fromhex(w["constant_term"]))
    ctx.step_index_install(s["zk_rows"], s["domains"], un(s["shifts"]), bytes.fromhex(s["constant_term"]))


def build_full_job(ctx, m, B: int, seed: int):
    """host-side `mina_state_jobs` of B complete jobs: 32 distinct chains + the 4 complete wrap proofs of the encoded fixture (statement,
    wrap proof, opening, the accumulator the statement carries), tiled to B.  Returns (StateJobs + keep, KimchiProofs + keep)."""
    import mina_bridge_amd.poseidon_params as PP
    fx, un = load_encoded_fixture()
    assert fx["poseidon_constants"] == PP.NAME, "the fixture was minted under another Poseidon constant set"
    install_fixture_indexes(ctx, fx, un)
    items = fx["proofs"]
    n = len(items)
    idx = np.arange(B) % n
    tile = lambda key, name: np.ascontiguousarray(np.stack([un(it[key][name]) if key else un(it[name]) for it in items])[idx].reshape(-1))
    n_old, n_evals = items[0]["n_old"], items[0]["n_evals"]
    st = m.MinaContext.make_pickles_statements(n_old, n_evals, {name: tile("statement", name) for name in m.lib.PicklesStatements.POINTER_FIELDS})
    # no recursion challenges beside the statement: they are its messages_for_next_wrap_proof.old_bulletproof_challenges (the one source a
    # verifier has; the library expands them and takes kimchi's digest of them on the way through the statement's own sponge)
    karr = {name: tile("kimchi", name) for name in items[0]["kimchi"] if name not in ("prev_prechallenges", "prev_chals")}
    kp = m.MinaContext.make_kimchi_proofs(B, 2, NPUB, karr, statements=st)
    nd = min(B, 32)
    recs, nf, hashes = make_chains(ctx, nd, seed)
    ci = np.arange(B) % nd
    # folding randomisers of the kernel-level job are the CALLER's to supply (include/mina_verify.h): drawn from the OS CSPRNG, as the
    # reference-shaped boundary does per job (api_verify.hip draw_randomisers); their values do not change the work
    rho = np.frombuffer(os.urandom(B * 32), np.uint8).reshape(B, 32).copy(); rho[:, 31] &= 0x3F
    arrays = dict(state_records=recs[ci].reshape(-1), state_nfields=nf[ci].reshape(-1), expected_hashes=hashes[ci].reshape(-1),
                  rand_base=fresh_randomiser(), sg_rand_base=fresh_randomiser(), acc_prechallenges=tile(None, "acc_prechallenges"), acc_sg=tile(None, "acc_sg"), acc_rho=rho.reshape(-1),
                  **{name: tile("opening", name) for name in ("lr", "delta", "sg", "z1", "z2")})
    scal = dict(with_states=1, with_ipa=1, with_accumulator=1, log2_domain=LOG2_DOMAIN, npub=NPUB, k=WRAP_K, n_evalpoints=NPTS, n_comms=NCOMMS + 2, acc_k=ACC_K, kimchi=kp)
    return m.MinaContext.make_state_jobs(B, arrays, **scal), kp, (recs[0], nf[0], hashes[0])


def algorithmic_bytes_per_proof() -> int:
    """what one state proof hands to the verifier, as laid out in HBM ('proof_len + pub_len'; here the kernel-ready
    form): 17 flattened states (50 field elements each) + their 17 expected hashes + 40 public inputs + the opening
    (30 L/R + delta + sg + 45 commitments as 64-B points, 7 scalars + sponge state) + the accumulator (16 x 16 B + sg + rho)"""
    return 17 * 50 * 32 + 17 * 32 + NPUB * 32 + (2 * WRAP_K + 2 + NCOMMS) * 64 + (7 + 3) * 32 + ACC_K * 16 + 64 + 32


def usable_cores() -> int:
    """cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box shared by several jobs reports all
    256 hardware threads in os.cpu_count() while cpu.max grants 16 cores' worth)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            quota = int(parts[0]) if parts[0] != "max" else -1
            period = int(parts[1]) if len(parts) > 1 else int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(baseline_sample, budget_s: float = 12.0):
    """BASELINE config C1 on the CPU restatement, NATIVE: oracle/composite_oracle.c (C, 4 x u64 Montgomery) verifies whole Proof-of-State jobs --
    17 state hashes + linkage, Pickles statement -> 40 public inputs, public-input commitment over cached Lagrange commitments, kimchi oracles +
    to_batch, the k = 15 wrap opening check (one 2^15 + ~100-point ark-style Pippenger MSM), the 2^16 Vesta accumulator MSM -- with pthreads
    ACROSS PROOFS, one thread per core the cgroup really grants (`usable_cores`), every thread a proof at a time.  Same bytes as the GPU job (the
    encoded fixture's four proofs); cross-checked value by value against the Python composite in tests/test_native_composite.py.  kind="port":
    the reference's Rust verifier cannot be built here.  This is the ONLY part of bench.py that touches oracle/."""
    from oracle import composite as C, oracle as O
    import mina_bridge_amd.poseidon_params as PP
    kind, sample = baseline_sample
    if kind != "full":
        return {"skipped": "the CPU leg runs the full job only"}
    recs, nf, hashes = sample
    nproc = os.cpu_count() or 1
    threads = min(usable_cores(), 128)
    fx, _ = load_encoded_fixture()
    srs = {c: O.srs_create(c, 1 << 16, threads=threads) for c in (0, 1)}
    t0 = time.perf_counter()
    C.setup(srs[0], srs[1], PP.default_params_bytes(0), PP.default_params_bytes(1), fx["wrap_index"], fx["step_index"], threads=threads)
    setup_s = time.perf_counter() - t0
    proofs = [C.make_proof(fx["proofs"][i % len(fx["proofs"])], recs, nf, hashes) for i in range(threads)]
    t0 = time.perf_counter(); v1 = C.verify_many(proofs[:1], 1); one_s = time.perf_counter() - t0          # one proof on one thread
    rounds, t0 = 0, time.perf_counter()
    ok = bool(v1.all())
    while True:
        ok = bool(C.verify_many(proofs, threads).all()) and ok; rounds += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    total = rounds * len(proofs)
    return {"value": total / el, "unit": "proofs/s", "cores": threads, "kind": "port", "implementation": "native C restatement (oracle/composite_oracle.c), pthreads across proofs",
            "cpu_model": cpu_model(), "nproc": nproc, "usable_cores": usable_cores(), "single_thread_value": 1.0 / one_s, "setup_s": setup_s,
            "sample": f"{total} full Proof-of-State verifications (BASELINE config C1 = the bench's own job per proof: 17 state hashes + Pickles statement -> public inputs + "
                      f"public-input commitment + kimchi oracles/to_batch + k=15 wrap opening check + 2^16 Vesta accumulator) in {el:.1f} s on {threads} threads, one proof per "
                      f"thread at a time; native C restatement of the o1-labs / arkworks algorithms (NOT the Rust reference, which cannot be built here); verdict ACCEPT: {ok}"}


def boundary_leg(m, local_rank: int, B: int, min_seconds: float = 2.0):
    """Secondary key `boundary_bytes_to_bools`: the reference-shaped boundary itself -- `mina_verify_state_batch` over B full-size bincode
    `MinaStateProof`s + 1057-byte public inputs (tests/golden/state_proofs_k15_bytes.json: the headline's four proofs as the caller of
    core/src/aligned.rs:31-58 would send them, tiled to B), host bytes in, verdict bytes out: parsing, `to_input` flattening, the ledger and
    consensus checks, the page-locked staging, PCIe both ways and the GPU job, back-to-back synchronous calls for >= `min_seconds`; then the
    same with two caller threads (a batcher's tasks).  Never `value`."""
    import ctypes
    import threading
    path = os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")
    if not os.path.exists(path):
        return {"skipped": "tests/golden/state_proofs_k15_bytes.json missing"}
    fxb = json.load(open(path))
    fx, un = load_encoded_fixture()
    import mina_bridge_amd.poseidon_params as PP
    if fxb["poseidon_constants"] != PP.NAME:
        return {"skipped": "byte fixture minted under another Poseidon constant set"}
    os.environ["MINA_VERIFY_DEVICE"] = str(local_rank)
    m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)            # the compiled-in Poseidon tables are the surrogate set (named in the output)
    install_fixture_indexes(m.lib.verify_all_devices(), fx, un)
    items = [(bytes.fromhex(it["proof"]), bytes.fromhex(it["pub"])) for it in fxb["proofs"]]
    P = [items[i % len(items)][0] for i in range(B)]; Q = [items[i % len(items)][1] for i in range(B)]
    lib = m.load_library()
    PP_ = (ctypes.c_char_p * B)(*P); PL = (ctypes.c_size_t * B)(*map(len, P)); QQ = (ctypes.c_char_p * B)(*Q); QL = (ctypes.c_size_t * B)(*map(len, Q))

    def call(out):
        rc = lib.mina_verify_state_batch(ctypes.c_size_t(B), PP_, PL, QQ, QL, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, lib.mina_last_error().decode()
    out = np.zeros(B, np.uint8)
    for _ in range(3):                                                # warm: contexts, tables, slots' page-locked buffers
        call(out)
    assert out.all(), "boundary verdicts must be ACCEPT"
    # one tampered proof in the batch: exactly its verdict byte is 0 (the culprit search of its chunk), untimed
    bad = bytearray(items[0][1]); bad[40] ^= 1
    QQ[B // 3] = bytes(bad)
    call(out)
    assert out.sum() == B - 1 and out[B // 3] == 0, f"a tampered public input must fail exactly its own proof: {int(out.sum())} of {B} accepted, rejected {np.flatnonzero(out == 0)[:16].tolist()}"
    QQ[B // 3] = Q[B // 3]
    calls, t0 = 0, time.perf_counter()
    while True:
        call(out); calls += 1
        el = time.perf_counter() - t0
        if el >= min_seconds:
            break
    assert out.all()
    single = {"proofs_per_s": calls * B / el, "ms_per_call": el / calls * 1e3, "calls": calls}
    def callers(k):
        outs = [np.zeros(B, np.uint8) for _ in range(k)]
        counts = [0] * k; stop = [time.perf_counter() + 1e9]
        def worker(i):
            while time.perf_counter() < stop[0]:
                call(outs[i]); counts[i] += 1
        for warm in (True, False):                                    # every slot the callers use allocates its page-locked staging on first use
            stop[0] = time.perf_counter() + (0.3 if warm else min_seconds)
            for i in range(k): counts[i] = 0
            th = [threading.Thread(target=worker, args=(i,)) for i in range(k)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            el = time.perf_counter() - t0
        assert all(o.all() for o in outs)
        return {"value": sum(counts) * B / el, "unit": "proofs/s", "calls": sum(counts)}
    two, four = callers(2), callers(4)
    # one call of 8 x B proofs: chunks of B, at most four on the GPU at a time
    big = None
    if B == 8192:
        n8 = 8 * B
        P8 = (ctypes.c_char_p * n8)(*(P * 8)); PL8 = (ctypes.c_size_t * n8)(*(list(map(len, P)) * 8)); Q8 = (ctypes.c_char_p * n8)(*(Q * 8)); QL8 = (ctypes.c_size_t * n8)(*(list(map(len, Q)) * 8))
        out8 = np.zeros(n8, np.uint8)
        def call8():
            rc = lib.mina_verify_state_batch(ctypes.c_size_t(n8), P8, PL8, Q8, QL8, out8.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0, lib.mina_last_error().decode()
        call8(); t0 = time.perf_counter(); k8 = 0
        while k8 < 3 or time.perf_counter() - t0 < 1.0:
            call8(); k8 += 1
        el8 = time.perf_counter() - t0
        assert out8.all()
        big = {"value": k8 * n8 / el8, "unit": "proofs/s", "proofs_per_call": n8, "ms_per_call": el8 / k8 * 1e3, "calls": k8}
    res = {"value": single["proofs_per_s"], "unit": "proofs/s", "proofs_per_call": B, "ms_per_call": single["ms_per_call"], "calls_timed": calls,
           "bytes_per_proof": len(P[0]) + len(Q[0]), "host_threads": int(os.environ.get("MINA_HOST_THREADS", min((os.cpu_count() or 2) // 2, 64))), "usable_cores": usable_cores(),
           "two_caller_threads": two, "four_caller_threads": four, "one_call_of_65536": big,
           "entry_point": "mina_verify_state_batch (include/mina_verify.h): bincode MinaStateProof + MinaStatePubInputs bytes -> verdict bytes",
           "poseidon_constants": m.lib.poseidon_params_name(), "process": "a fresh process holding only libminaverify.so (no torch): the operator's verifier process",
           "note": "host bytes in, bools out: parsing, to_input flattening, ledger + consensus checks, pinned staging, PCIe both ways, the GPU job (folding "
                   "randomisers from the OS CSPRNG per chunk); one tampered proof in a warm-up call failed alone"}
    m.lib.verify_shutdown()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--jobs", type=int, default=8192, help="state proofs per step (one mina_state_job_batch_dev call)")
    ap.add_argument("--pipeline", type=int, default=16, help="internal stream lanes over which consecutive steps are issued")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("full", "kimchi", "prepared"), default="full",
                    help="full (default): the whole verifier from parsed proofs -- Pickles statement -> public inputs, kimchi oracles + to_batch, opening check, "
                         "accumulator, 17 state hashes (fixture tests/golden/statement_k15.json, 4 distinct proofs); kimchi: the same without the statement stage "
                         "(public inputs given; tests/golden/kimchi_k15.json); prepared: pre-derived BatchEvaluationProof rows (round 2's first headline)")
    ap.add_argument("--kimchi", action="store_true", help="alias of --mode kimchi")
    ap.add_argument("--no-probes", action="store_true", help="skip the isolated-kernel / C2 probes and the sustained / C5 legs (profiling runs: only the timed loop launches kernels)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the bytes -> bools leg (mina_verify_state_batch on serialized proofs)")
    args = ap.parse_args()
    if args.kimchi:
        args.mode = "kimchi"
    args.kimchi = args.mode != "prepared"                      # the wrap leg starts from the raw proof in both non-prepared modes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # The bytes -> bools leg runs FIRST, in a process of its own that holds only the library (no torch): the operator's verifier process.  It must
    # not share the GPU with another process's queues (a second process holding 24 hardware queues makes the scheduler time-slice them: the
    # same call took 288 ms instead of 55), and inside THIS process the 40-odd streams of the headline's context would populate the runtime's
    # queue pool first (67 - 72 ms).  GPU_MAX_HW_QUEUES=16 there: with the system runtime a lone job's three legs overlap best at <= 16 (54.6 ms;
    # 24: 68.4 ms), while the 16-lane pipeline of the headline below wants one queue per lane plus a few (24).
    boundary = None
    if not args.no_boundary and os.environ.get("MINA_BENCH_SHARE_GPU") != "1" and int(os.environ.get("RANK", "0")) == 0:      # rank 0 reports it; the others wait at the first barrier
        import subprocess
        env = dict(os.environ); env["GPU_MAX_HW_QUEUES"] = os.environ.get("MINA_BOUNDARY_HW_QUEUES", "16")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--boundary-only", str(local_rank), str(args.jobs)], capture_output=True, text=True, timeout=900, env=env)
            boundary = json.loads(r.stdout.strip().split("\n")[-1]) if r.returncode == 0 and r.stdout.strip() else {"error": (r.stderr or r.stdout)[-400:]}
        except (subprocess.TimeoutExpired, ValueError) as e:
            boundary = {"error": repr(e)[:400]}

    import torch
    import mina_bridge_amd as m

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} without a torch.distributed.run launch (WORLD_SIZE=1): measuring 1 GPU", file=sys.stderr)
        args.gpus = world                      # the number of ranks actually running is what is reported
    dist_on = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # MINA_BENCH_SHARE_GPU=1 (test hook): every rank uses GPU 0 and the ranks rendezvous over gloo -- lets the N > 1 code path
    # (barriers, MAX over ranks, verdict all-gather, aggregate value) be exercised on a 1-GPU box; never set by the driver
    share_gpu = os.environ.get("MINA_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = m.MinaContext(local_rank)
    for f in (0, 1):
        ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
    ctx.srs_create(CURVE_VESTA, 1 << 16)                       # both SRS regenerated on the GPU (K4) + window tables
    ctx.srs_create(CURVE_PALLAS, 1 << 16)
    B = args.jobs
    seed = 0x6D696E61 + rank
    if args.mode == "full":
        (hj, keep), kp, chain_sample = build_full_job(ctx, m, B, seed)
        baseline_sample = ("full", chain_sample)
    else:                                                      # partial jobs: inputs through the test helpers (tools/bench_partial_inputs.py)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_partial_inputs as P
        (hj, keep), sample = P.build_batch(ctx, m, B, seed)
        kp = None
        if args.mode == "kimchi":                              # the wrap leg from the raw proofs, public inputs given
            kp, op, kpub = P.build_kimchi_section(ctx, m, B)
            for name in ("sponge_state", "sponge_pos", "cip", "evalpoints", "evalscale", "polyscale", "comms"):
                setattr(hj, name, None)
            keep = [a for a in keep] + [kp]
            for name, arr in list(op.items()) + [("public_inputs", kpub)]:
                arr = np.ascontiguousarray(arr); keep.append(arr); setattr(hj, name, arr.ctypes.data)
            hj.n_comms = NCOMMS + 2
        baseline_sample = ("prepared", sample)                 # the CPU composite of the partial modes starts from the derived rows
    dev = torch.device("cuda", local_rank)
    import ctypes
    dj = m.lib.StateJobs()
    ctypes.memmove(ctypes.byref(dj), ctypes.byref(hj), ctypes.sizeof(m.lib.StateJobs))
    by_addr = {a.ctypes.data: a for a in keep if isinstance(a, np.ndarray)}
    dtensors = []
    for name in m.lib.StateJobs.POINTER_FIELDS:                # every section resident in HBM (torch owns the buffers)
        addr = getattr(hj, name)
        if addr:
            t = torch.from_numpy(np.array(by_addr[addr].view(np.uint8).reshape(-1))).to(dev)
            dtensors.append(t); setattr(dj, name, t.data_ptr())
    if kp is not None:                                         # the kimchi section's arrays live in HBM as well
        import ctypes as ct
        dk = m.lib.KimchiProofs()
        ct.memmove(ct.byref(dk), ct.byref(kp[0]), ct.sizeof(m.lib.KimchiProofs))
        kaddr = {a.ctypes.data: a for a in kp[1] if isinstance(a, np.ndarray)}
        for name in m.lib.KimchiProofs.POINTER_FIELDS:
            addr = getattr(kp[0], name)
            if addr:
                if name == "public_inputs":
                    setattr(dk, name, dj.public_inputs); continue
                t = torch.from_numpy(kaddr[addr].view(np.uint8).reshape(-1)).to(dev); dtensors.append(t); setattr(dk, name, t.data_ptr())
        if kp[0].statements:                                   # the statement sections too
            hst, hkeep = next(a for a in kp[1] if isinstance(a, tuple))
            dst = m.lib.PicklesStatements()
            ct.memmove(ct.byref(dst), ct.byref(hst), ct.sizeof(m.lib.PicklesStatements))
            saddr = {a.ctypes.data: a for a in hkeep}
            for name in m.lib.PicklesStatements.POINTER_FIELDS:
                addr = getattr(hst, name)
                if addr:
                    t = torch.from_numpy(saddr[addr].view(np.uint8).reshape(-1)).to(dev); dtensors.append(t); setattr(dst, name, t.data_ptr())
            dk.statements = ct.addressof(dst)
        dj.kimchi = ct.addressof(dk)
    ctx.state_jobs_prepare(LOG2_DOMAIN, NPUB)
    ctx.set_pipeline(args.pipeline)
    nslots = max(args.pipeline, 1)
    d_out = [torch.zeros(B + 4, dtype=torch.int32, device=dev) for _ in range(nslots)]
    torch.cuda.synchronize()
    it = [0]

    def step():
        o = d_out[it[0] % nslots]; it[0] += 1
        ctx.state_job_batch_dev(dj, o.data_ptr(), o.data_ptr() + 4 * B)

    def verdicts_ok(last):
        # the `last` most recent steps (at most one per output slot): every proof ACCEPT, flags = {folded IPA ok, no malformed input, folded accumulator ok}
        used = [d_out[(it[0] - 1 - i) % nslots] for i in range(min(last, nslots))]
        return all(o.cpu().numpy().tolist() == [1] * B + [1, 0, 1, 0] for o in used)

    n_warm = max(args.warmup, nslots)                          # every lane allocates its workspace during warm-up
    for _ in range(n_warm):
        step()
    ctx.synchronize()
    assert verdicts_ok(n_warm), "warm-up verdicts must be ACCEPT"
    for o in d_out:
        o.zero_()

    def barrier():
        if dist_on:
            dist.barrier()

    gathered = None
    mask = sum(1 << b for b in PROF_STAGES.values())
    ctx.prof_enable(mask)                                      # HIP events around the candidate dominant kernels, on their lane streams
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize(); torch.cuda.synchronize()
    if dist_on:                                                # the shards' verdict words travel over RCCL inside the timed region
        mine = d_out[(it[0] - 1) % nslots][:B].contiguous()
        if share_gpu:
            mine = mine.cpu()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        if not share_gpu:
            torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_read()
    ctx.prof_enable(0)
    assert verdicts_ok(args.steps), "timed-region verdicts must be ACCEPT"
    if gathered is not None:
        assert all(int(g.sum()) == B for g in gathered), "every rank's shard must be ACCEPT"

    # the same loop for >= 2 s whatever --steps was (pipeline fill / drain is then a small part of the sample); secondary key
    sustained = None
    if elapsed < 2.0 and not args.no_probes:
        n_sus = int(args.steps * 2.2 / max(elapsed, 1e-3)) + 1
        barrier(); torch.cuda.synchronize(); ts = time.perf_counter()
        for _ in range(n_sus):
            step()
        ctx.synchronize(); torch.cuda.synchronize(); barrier()
        el_s = time.perf_counter() - ts
        assert verdicts_ok(n_sus)
        sustained = {"steps": n_sus, "seconds": el_s, "value": args.gpus * n_sus * B / el_s, "unit": "proofs/s"}

    # BASELINE config C5 as written: 4096 state proofs in total, split over the ranks (strong scaling) -- 4096 / N per rank and step
    c5 = None
    if B >= 4096 // max(world, 1) and 4096 % max(world, 1) == 0 and not args.no_probes:
        B5 = 4096 // world
        dj5 = m.lib.StateJobs(); ctypes.memmove(ctypes.byref(dj5), ctypes.byref(dj), ctypes.sizeof(m.lib.StateJobs)); dj5.batch = B5
        keep5 = []
        if kp is not None:
            dk5 = m.lib.KimchiProofs(); ctypes.memmove(ctypes.byref(dk5), ctypes.byref(dk), ctypes.sizeof(m.lib.KimchiProofs)); dk5.batch = B5
            dj5.kimchi = ctypes.addressof(dk5); keep5.append(dk5)
        def step5():
            o = d_out[it[0] % nslots]; it[0] += 1
            ctx.state_job_batch_dev(dj5, o.data_ptr(), o.data_ptr() + 4 * B5)
        for _ in range(nslots):
            step5()
        ctx.synchronize()
        n5 = max(32, args.steps)
        barrier(); torch.cuda.synchronize(); t5 = time.perf_counter()
        for _ in range(n5):
            step5()
        ctx.synchronize(); torch.cuda.synchronize(); barrier()
        el5 = time.perf_counter() - t5
        last = d_out[(it[0] - 1) % nslots].cpu().numpy()
        assert last[:B5].tolist() == [1] * B5 and last[B5:B5 + 4].tolist() == [1, 0, 1, 0], "C5 verdicts must be ACCEPT"
        c5 = {"value": n5 * 4096 / el5, "unit": "proofs/s", "proofs_total_per_step": 4096, "proofs_per_rank_per_step": B5, "steps": n5, "scaling": "strong",
              "ms_per_step": el5 / n5 * 1e3}
        for o in d_out:
            o.zero_()

    # the candidate dominant kernels with nothing else on the GPU: one lane, HIP events on that lane's stream
    prof_iso = {}
    if not args.no_probes:
        ctx.set_pipeline(1)
        for _ in range(2):
            step()
        ctx.synchronize()
        ctx.prof_enable(mask)
        for _ in range(6):
            step()
        prof_iso = ctx.prof_read()
        ctx.prof_enable(0)
        # latency of one call (B jobs) alone
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(4):
            step()
        ctx.synchronize()
        call_latency_ms = (time.perf_counter() - t1) / 4 * 1e3
        # secondary key (round 1's headline, BASELINE config C2): 8 independent 2^16 Vesta accumulator checks per call, 16 lanes
        ctx.set_pipeline(16)
        pre8, sg8 = make_accumulators(ctx, 8, 4242 + rank)
        d_pre8 = torch.from_numpy(pre8.reshape(-1)).to(dev); d_sg8 = torch.from_numpy(sg8.reshape(-1)).to(dev)
        d_v8 = torch.zeros(8, dtype=torch.int32, device=dev)
        c2 = lambda: ctx.accumulator_check_multi_dev(CURVE_VESTA, ACC_K, 8, d_pre8.data_ptr(), d_sg8.data_ptr(), d_v8.data_ptr())
        for _ in range(32):
            c2()
        ctx.synchronize(); torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(400):
            c2()
        ctx.synchronize()
        c2_rate = 8 * 400 / (time.perf_counter() - t2)
        assert d_v8.cpu().numpy().tolist() == [1] * 8
        ctx.set_pipeline(1)
    else:
        call_latency_ms = None
        c2_rate = None

    if dist_on:
        t = torch.tensor([elapsed, sustained["seconds"] if sustained else 0.0, c5["ms_per_step"] if c5 else 0.0], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if sustained:
            sustained["seconds"] = float(t[1].item()); sustained["value"] = args.gpus * sustained["steps"] * B / sustained["seconds"]
        if c5:
            c5["ms_per_step"] = float(t[2].item()); c5["value"] = 4096 / (c5["ms_per_step"] * 1e-3)
    ctx.close()
    if boundary is not None and dist_on:                           # every rank ran its own leg on its own GPU; rank 0 reports the sum (a failed leg counts 0)
        tb = torch.tensor([boundary.get("value", 0.0), boundary.get("two_caller_threads", {}).get("value", 0.0)], dtype=torch.float64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        boundary["value_all_ranks"] = float(tb[0].item())
        if "two_caller_threads" in boundary:
            boundary["two_caller_threads"]["value_all_ranks"] = float(tb[1].item())

    if rank == 0:
        def avg_us(p, name):
            n, ms = p.get(name, [0, 0.0])
            return (ms / n) * 1e3 if n else None
        iso = {k: avg_us(prof_iso, k) for k in PROF_STAGES}
        ovl = {k: avg_us(prof, k) for k in PROF_STAGES}
        src = iso if iso.get("pstate_hash") else ovl
        # dominant kernel of the job: the protocol-state hash (17 sponges of ~27 permutations per proof)
        kern_us = src.get("pstate_hash")
        nstates = B * STATES_PER_PROOF
        perms = nstates * (25 + 1)                             # 49 body fields -> 25 permutations, + 1 for H(previous, body)
        hash_bytes = nstates * (50 * 32 + 32)                  # per state: 50 field elements read, one hash written
        achieved = hash_bytes / (kern_us * 1e-6) / 1e9 if kern_us else None
        traffic_per_state, traffic_src = pstate_traffic()
        out = {
            "metric": "Mina state proofs verified/sec (batch)",
            "value": args.gpus * args.steps * B / elapsed,
            "unit": "proofs/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "call_latency_ms": call_latency_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x8-montgomery (255-bit prime fields, integer)", "data": "synthetic",
            "config": {"workload": "C3: full Proof-of-State job per proof -- 17 protocol-state hashes (chain of 16 + bridge tip) vs public inputs + linkage, "
                                   + ("the Pickles statement's deferred values and digests, " if args.mode == "full" else "")
                                   + ("kimchi oracles + to_batch of the wrap proof, " if args.kimchi else "") +
                                   "wrap-proof public-input commitment (40 inputs, 2^15 Pallas domain), wrap IPA opening (k=15, 45+ commitments x 2 points), "
                                   "2^16-base Vesta step-accumulator check; verdict per proof, bit-exact vs the CPU oracle composite (tests/test_state_job.py)",
                       "proofs_per_step": B, "pipeline_lanes": args.pipeline, "warmup_steps_run": n_warm,
                       "mode": args.mode,
                       "distinct_inputs": {"full": "32 chains of flattened protocol-state records (random field content, GPU-linked) and the 4 complete wrap proofs of "
                                                   "tests/golden/statement_k15_encoded.json (statement + proof + its accumulator) per rank; the statements' application "
                                                   "state is the fixture's, not the hash of the chain tiled beside it (that binding, on serialized states through the "
                                                   "parsers: tests/test_verify_fullsize.py)",
                                           "kimchi": "32 chains, 4 wrap proofs (tests/golden/kimchi_k15.json), 32 accumulators per rank",
                                           "prepared": "32 chains, 8 wrap openings (tests/golden/state_job_k15.json), 32 accumulators per rank"}[args.mode],
                       "folding": "IPA and accumulator checks folded over the step's batch (kimchi batch_verify's shape); randomisers drawn from the OS CSPRNG by the caller",
                       "not_in_job": {"full": "bin_prot / bincode parsing of the containers and the consensus pre-checks (host side of the boundary, before the timed region)",
                                      "kimchi": "parsing and the Pickles statement -> public-input derivation (public inputs given)",
                                      "prepared": "kimchi oracles/linearisation, the statement derivation, parsing (BatchEvaluationProof rows given)"}[args.mode],
                       "wrap_leg": {"full": "from the parsed wrap proof, all on the GPU: Pickles statement -> 40 public inputs (compute_deferred_values, the two message digests, "
                                            "packing) -> public-input commitment -> kimchi oracles + to_batch -> combined opening check (synthetic wrap-size verifier index: "
                                            "domain 2^15, 47 commitments; synthetic step index)",
                                    "kimchi": "kimchi oracles + to_batch on the GPU from the raw wrap proofs (synthetic index, domain 2^15, 40 public inputs, 47 commitments)",
                                    "prepared": "pre-derived BatchEvaluationProof rows (45 commitments)"}[args.mode],
                       "sharding": f"proof-level, {args.gpus} rank(s); verdict words all-gathered over RCCL" if dist_on else "single rank",
                       "algorithmic_bytes_per_proof": algorithmic_bytes_per_proof()},
            "roofline": {"bound": "hbm", "kernel": "pstate_hash_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": (nstates * traffic_per_state) if traffic_per_state else None,
                         "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read from {TRAFFIC_FILE} ({traffic_src})" if traffic_per_state else None,
                         "algorithmic_bytes_per_launch": hash_bytes, "states_per_launch": nstates, "avg_launch_us": kern_us,
                         "avg_launch_us_in_timed_region": ovl.get("pstate_hash"),
                         "note": "avg_launch_us: HIP events on the lane stream around the kernel, launches with nothing else on the GPU right after the "
                                 "timed region (second figure: inside it, lanes overlapping).  The path is integer-VALU bound (SURVEY.md 8d): the HBM "
                                 "fraction is reported because the metric asks for it, roofline_valu is the bound that matters; traffic: see profiles/"},
            "stage_us": {"isolated": iso, "in_timed_region": ovl},
            "sustained": sustained,
            "c5_4096_total_strong": c5,
            "boundary_bytes_to_bools": boundary,
            "c2_accumulator_only": {"value": c2_rate, "unit": "accumulator checks/s",
                                    "note": "BASELINE config C2 alone (round 1's headline): un-folded 2^16-base Vesta IPA accumulator checks, 8 per call, 16 lanes",
                                    # the metric's second half ("MSM HBM GB/s vs peak"): one check = one 2^16-base MSM; algorithmic bytes = bases + scalars
                                    "msm_hbm": None if not c2_rate else {"algorithmic_bytes_per_msm": 65536 * (64 + 32), "achieved_GBps": c2_rate * 65536 * 96 / 1e9, "peak_GBps": HBM_PEAK_GBPS,
                                                                         "frac": c2_rate * 65536 * 96 / 1e9 / HBM_PEAK_GBPS,
                                                                         "note": "the bucket MSM is bound by the group law's multiply-accumulates (msm_accumulate_bucket_kernel 58 % of a check, "
                                                                                 "bucket reductions 19 %: tools/c2_rate.py under rocprofv3), not by HBM; with the 16-window fixed-base tables the kernel "
                                                                                 "actually reads 64 MiB per MSM (~10x the algorithmic bytes) and is still far from the HBM roof"}},
        }
        if kern_us:
            peak = CHIP_SIMDS * 64 * CLOCK_HZ / MAD_ISSUE_CYCLES
            got = perms * 3 * 55 * MADS_PER_LANE_ROUND / (kern_us * 1e-6)
            out["roofline_valu"] = {"bound": "64-bit multiply-accumulate issue (v_mad_u64_u32, 7.0 cycles per wave64 instruction)", "kernel": "pstate_hash_kernel",
                                    "achieved": got / 1e12, "peak": peak / 1e12, "unit": "T limb-MAC/s", "frac": got / peak, "permutations_per_launch": perms,
                                    "limb_macs_per_permutation": 3 * 55 * MADS_PER_LANE_ROUND,
                                    "note": "9 x 29-bit limbs, no carry instructions: 765 of a round's ~1055 VALU instructions are multiply-accumulates"}
        if not args.no_cpu_baseline and args.gpus == 1:       # the CPU leg is timed at N = 1 only (rank 0)
            out["cpu_baseline"] = cpu_baseline(baseline_sample)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--boundary-only":     # the bytes -> bools leg in a process of its own (see main): only the library, no torch
        import mina_bridge_amd as _m
        print(json.dumps(boundary_leg(_m, int(sys.argv[2]), int(sys.argv[3]))), flush=True)
    else:
        main()
