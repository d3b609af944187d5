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
`--pipeline` lanes (independent batches overlap on the GPU; 4 since round 6: each lane forks a job's three legs -- state hashes / wrap-proof chain / accumulator --
onto helper streams of its own and joins them before the verdict kernel, and every kernel but the state hash raises its wave priority: profiles/r06_dev_fork.md).  Not in the job: bin_prot / bincode parsing of the containers and the
consensus pre-checks (host side of the boundary, done by the caller before the timed region).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--jobs B] [--pipeline L]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: proof-level sharding (SURVEY.md 8e variant 1): every rank verifies its own proofs against its replica of the
SRS tables; the ranks' verdict words are all-gathered over RCCL inside the timed region; weak scaling.  One JSON line on rank 0.
`python bench.py --gpus N` WITHOUT a torch.distributed launch (WORLD_SIZE unset) starts its N ranks itself (`launch_ranks`): one child
process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's JSON line passed through.  On a box with fewer than N GPUs the
ranks share GPU 0 and rendezvous over gloo (`"shared_gpu": true`, `"gpus_physical"` in the line): the N > 1 code path is exercised, the
figure is NOT a scaling number.  At N > 1 rank 0 also times the PRODUCT's own multi-device path -- ONE process, $MINA_VERIFY_DEVICES = the N
GPUs, `mina_verify_state_batch` on serialized proofs (`boundary_bytes_to_bools.all_devices`) -- while the other ranks wait.

Secondary legs that need a process of their own run in one (rank 0, before the main process touches the GPU): the bytes -> bools boundary, BASELINE C4, `one_proof_per_call` (the
reference's call shape) and BASELINE C2 (`c2_accumulator_only`, incl. ONE check alone on the chip).  In the main process the lone-call latencies and the isolated launches that time the
dominant kernel come BEFORE anything creates more streams than the process has hardware queues (such a process stays slower for life).
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
# 29 bits (fp29.cuh, signed-quotient-digit forms since late round 5): per lane and round 2 squarings (99 limb multiply-accumulates each), 2 products (135) and one
# 3-term dot product with the round constant inside its reduction (306) = 774 `v_mad_u64_u32` / `v_mad_i64_i32` beside ~150 simple instructions (64-bit column shifts,
# limb masks, the doubled limbs of a squaring; the lazy forms' 45 digit negations are gone: 925 VALU instructions per lane-round, 965 before).  ONE model
# (profiles/r04_valu_roofline.md): every instruction takes its issue slot, and the ceiling is the measured rate of THAT MIX -- 153 multiply-accumulates interleaved with
# 16 shifts, 9 masks, 4 32-bit shifts per trip, 8 waves per SIMD (`microbench --ratio`, profiles/r05_microbench_ratio.jsonl): 5.20 nominal-clock cycles per
# multiply-accumulate OF THE MIX (the lazy forms' mix on the same box: 5.47; a pure stream: 4.52).
MADS_PER_LANE_ROUND = 2 * 99 + 2 * 135 + 306
MAD_ISSUE_CYCLES = 5.20                               # the signed-digit round's mix (profiles/r05_microbench_ratio.jsonl; the lazy mix: 5.38 - 5.47 over the boxes of round 5)
# SURVEY.md 8d's peak: a PURE stream of v_mad_u64_u32 (16 independent accumulators, operands on distinct register banks, 8 waves per SIMD) issues one per 4.52
# cycles at the nominal 2.4 GHz = 1.88 ns per wave64 instruction per SIMD (profiles/r05_microbench_ratio.jsonl, the S = 0 row; the same in round 4): 34.8 T limb-MAC/s for the chip.
# `roofline` is quoted against THIS peak (it forgives nothing: the round's ~150 shifts / masks per 774 multiply-accumulates count as lost issue slots);
# `roofline_valu` keeps the own-mix ceiling beside it.
PURE_MAC_ISSUE_CYCLES = 4.52
CHIP_SIMDS, CLOCK_HZ = 1024, 2.4e9
PROF_STAGES = {"pstate_hash": 11, "ipa_transcript": 12, "kimchi_to_batch": 13, "pickles_statement": 14, "msm_accumulate": 3}
# HBM-side bytes per protocol-state hash: read from the tracked summary of the rocprofv3 PMC passes (FETCH_SIZE x 2 -- the gfx950 correction
# of MI355X_MICROARCH.md for 16-B-per-lane loads -- + WRITE_SIZE, KiB, over the states of one launch); null when the file is missing
TRAFFIC_FILE = os.path.join("profiles", "pstate_hash_traffic.json")


class PowerSampler:
    """socket power and shader clock DURING the timed region, read from the amdgpu hwmon files of this rank's GPU every 0.2 s by a host thread (two small sysfs
    reads: nothing is launched, nothing is forked).  Why it rides in the line: the step holds the chip at its socket power cap (1400 W) and the clock settles at
    ~2.25 GHz, while the microbench that measures `roofline.peak` (register-only streams) stays below the cap at 2.39 GHz -- profiles/r05_clock_power.md."""

    def __init__(self, pci_bus_id=None, interval=0.2, drm_root="/sys/class/drm"):
        import glob
        self.dir, self.rows, self._stop, self._th, self.interval = None, [], False, None, interval
        cands = []
        for h in sorted(glob.glob(os.path.join(drm_root, "card*", "device", "hwmon", "hwmon*"))):
            if os.path.exists(os.path.join(h, "power1_input")) and os.path.exists(os.path.join(h, "freq1_input")):
                cands.append((os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))).lower(), h))
        if pci_bus_id:
            cands = [c for c in cands if c[0] == pci_bus_id.lower()] or (cands if len(cands) == 1 else [])
        if len(cands) >= 1 and (pci_bus_id or len(cands) == 1):
            self.dir = cands[0][1]

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except Exception:
            return None

    def _loop(self):
        while not self._stop:
            w, hz = self._read(os.path.join(self.dir, "power1_input")), self._read(os.path.join(self.dir, "freq1_input"))
            if w is not None and hz is not None:
                self.rows.append((w / 1e6, hz / 1e6, time.perf_counter()))
            time.sleep(self.interval)

    def start(self):
        if self.dir:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True); self._th.start()
        return self

    def stop(self):
        self._stop = True
        if self._th: self._th.join()
        if not self.rows:
            return None
        w, f = [r[0] for r in self.rows], [r[1] for r in self.rows]
        cap = self._read(os.path.join(self.dir, "power1_cap"))
        return {"socket_power_w_avg": sum(w) / len(w), "socket_power_w_max": max(w), "power_cap_w": cap / 1e6 if cap else None,
                "sclk_mhz_avg": sum(f) / len(f), "sclk_mhz_min": min(f), "sclk_mhz_max": max(f), "samples": len(w),
                "source": f"amdgpu hwmon power1_input / freq1_input of this rank's GPU, sampled every {self.interval} s inside the timed region",
                "note": "nominal clock 2400 MHz; the cycle figures of roofline_valu / step_valu are quoted at the nominal clock"}


def pstate_traffic():
    try:
        t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
        return (2 * t["fetch_size_kib"] + t["write_size_kib"]) * 1024 / t["states_per_launch"], t.get("source", TRAFFIC_FILE)
    except (OSError, KeyError, ValueError):
        return None, None


def msm_traffic(c2_rate):
    """HBM-side bytes of one 2^16 accumulator check, every kernel of it, from the tracked summary of the rocprofv3 PMC passes (profiles/msm_traffic.json:
    FETCH_SIZE with the gfx950 factor 2 on the coalesced streams and the CALIBRATED factor 1 on the accumulate kernels' random 64-B gathers, + WRITE_SIZE)"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "msm_traffic.json")))
        return {"traffic": t["hbm_bytes_per_check"], "traffic_ratio_to_algorithmic": t["ratio_to_algorithmic"], "traffic_GBps": c2_rate * t["hbm_bytes_per_check"] / 1e9,
                "traffic_frac_of_peak": c2_rate * t["hbm_bytes_per_check"] / 1e9 / HBM_PEAK_GBPS, "traffic_source": "profiles/msm_traffic.json (" + t["source"] + ")"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def clock_normalised(frac, power):
    """`frac` is quoted at the nominal 2.4 GHz; the step holds the chip on its power cap and the clock it gets differs by box (2.10 - 2.30 GHz: profiles/r05_clock_power.md), so the
    same build reads 0.76 on one box and 0.82 on another.  frac_at_sampled_clock = achieved / (peak x sclk_avg / 2400): comparable between boxes (VERDICT r05 next #3a).
    The clock is the timed region's average (the isolated launches that time the kernel run right after it, same thermal state)."""
    if not power or not power.get("sclk_mhz_avg") or frac is None:
        return {"frac_at_sampled_clock": None}
    return {"frac_at_sampled_clock": frac / (power["sclk_mhz_avg"] / (CLOCK_HZ / 1e6)), "sampled_sclk_mhz": power["sclk_mhz_avg"]}


def toolchain():
    """compiler and runtime the library was built with / runs under: the instruction counts, register numbers and s_nop counts the performance contract rests on are
    properties of ONE compiler build (tests/test_code_object.py pins them)"""
    import subprocess
    out = {}
    try:
        v = subprocess.run(["hipcc", "--version"], capture_output=True, text=True, timeout=30).stdout
        out["hipcc"] = " / ".join(l.strip() for l in v.splitlines() if l.startswith("HIP version") or "clang version" in l)[:200]
    except Exception as e:                                       # noqa: BLE001
        out["hipcc"] = "unavailable: " + repr(e)[:80]
    try:
        import torch
        out["torch"] = torch.__version__; out["torch_hip"] = getattr(torch.version, "hip", None)
    except Exception:                                            # noqa: BLE001
        pass
    try:
        out["rocm_runtime"] = open("/opt/rocm/.info/version").read().strip()
    except OSError:
        pass
    return out


def msm_single_traffic(kern_us):
    """HBM-side bytes of ONE lone 2^16 accumulator check (profiles/msm_traffic.json `single_check`: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/c2_single.py)"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "msm_traffic.json")))["single_check"]
        return {"traffic": t["hbm_bytes_per_check"], "traffic_ratio_to_algorithmic": t["ratio_to_algorithmic"], "traffic_GBps_kernels": t["hbm_bytes_per_check"] / kern_us / 1e3 if kern_us else None,
                "traffic_source": "profiles/msm_traffic.json single_check (" + t["source"] + ")"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def step_valu(ms_per_step, proofs_per_step):
    """the whole step against VALU instruction issue: wave-instructions of one step, every kernel (profiles/step_valu.json: rocprofv3 --pmc SQ_INSTS_VALU over a
    single-lane run, setup kernels left out), priced at the measured 4.4 cycles per wave64 instruction per SIMD of the 29-bit mix with the SIMDs full"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "step_valu.json")))
        if t["proofs_per_step"] != proofs_per_step:
            return None
        floor_ms = t["valu_wave_instructions_per_step"] * t["cycles_per_wave_instruction"] / (CHIP_SIMDS * CLOCK_HZ) * 1e3
        return {"wave_instructions_per_step": t["valu_wave_instructions_per_step"], "floor_ms_per_step": floor_ms, "frac": floor_ms / ms_per_step,
                "bound": "VALU instruction issue (every instruction takes its slot: profiles/r04_valu_roofline.md)", "source": "profiles/step_valu.json (" + t["source"] + ")"}
    except (OSError, KeyError, ValueError):
        return None


def msm_valu(c2_rate):
    """the VALU side of the MSM's roofline: wave-instructions of one 2^16 accumulator check, every kernel of it (profiles/msm_valu.json: rocprofv3 --pmc
    SQ_INSTS_VALU, tools/profile_c2_sq.sh), priced at the measured issue rate of the 29-bit mix with the SIMDs full (4.4 cycles per wave64 instruction)"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "msm_valu.json")))
        floor_s = t["valu_wave_instructions_per_check"] * 4.4 / (CHIP_SIMDS * CLOCK_HZ)
        return {"bound": "VALU instruction issue: every instruction takes its slot (profiles/r04_valu_roofline.md), 4.4 cycles at 2.4 GHz per wave64 instruction per SIMD",
                "wave_instructions_per_check": t["valu_wave_instructions_per_check"], "floor_us_per_check": floor_s * 1e6, "achieved_us_per_check": 1e6 / c2_rate,
                "frac": floor_s * c2_rate, "source": "profiles/msm_valu.json (" + t["source"] + ")"}
    except (OSError, KeyError, ValueError):
        return None


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


MANY_FIXTURE = os.path.join("tests", "golden", "statement_k15_many.npz")


def load_wrap_sections():
    """every distinct complete wrap proof the tree holds, in the C-ABI's byte layouts: {section: uint8 [n, bytes per proof]} with sections "statement.<name>",
    "kimchi.<name>", "opening.<name>", "acc_prechallenges", "acc_sg" -- the 4 proofs of tests/golden/statement_k15_encoded.json followed by the proofs of
    tests/golden/statement_k15_many.npz (minted by the same generator under the same indexes: tools/mint_many.sh, tests/golden/encode_statement_fixture.py --many)."""
    fx, un = load_encoded_fixture()
    cols = {}
    for it in fx["proofs"]:
        flat = {"acc_prechallenges": it["acc_prechallenges"], "acc_sg": it["acc_sg"]}
        for grp in ("statement", "kimchi", "opening"):
            flat.update({grp + "." + k: v for k, v in it[grp].items()})
        for k, v in flat.items():
            cols.setdefault(k, []).append(un(v))
    sec = {k: np.stack(v) for k, v in cols.items()}
    try:
        z = np.load(os.path.join(ROOT, MANY_FIXTURE))
        if bytes(z["poseidon_constants"]).decode() == fx["poseidon_constants"]:
            sec = {k: np.concatenate([v, z[k]]) for k, v in sec.items()}
    except (OSError, KeyError, ValueError):
        pass                                                       # only the 4 proofs of the JSON fixture: config.distinct_inputs says so
    return fx, un, sec, fx["proofs"][0]["n_old"], fx["proofs"][0]["n_evals"]


def build_full_job(ctx, m, B: int, seed: int, distinct_chains: int = 0):
    """host-side `mina_state_jobs` of B complete jobs: `distinct_chains` distinct chains (default: one per proof) + every distinct complete wrap proof of the
    fixtures (statement, wrap proof, opening, the accumulator the statement carries: `load_wrap_sections`), tiled to B.  Returns (StateJobs + keep, KimchiProofs +
    keep, a chain sample, {"chains": .., "wrap_proofs": ..})."""
    import mina_bridge_amd.poseidon_params as PP
    fx, un, sec, n_old, n_evals = load_wrap_sections()
    assert fx["poseidon_constants"] == PP.NAME, "the fixture was minted under another Poseidon constant set"
    install_fixture_indexes(ctx, fx, un)
    n = len(sec["acc_sg"])
    idx = np.arange(B) % n
    tile = lambda key, name: np.ascontiguousarray(sec[(key + "." + name) if key else name][idx].reshape(-1))
    st = m.MinaContext.make_pickles_statements(n_old, n_evals, {name: tile("statement", name) for name in m.lib.PicklesStatements.POINTER_FIELDS})
    # no recursion challenges beside the statement: they are its messages_for_next_wrap_proof.old_bulletproof_challenges (the one source a
    # verifier has; the library expands them and takes kimchi's digest of them on the way through the statement's own sponge)
    karr = {name[len("kimchi."):]: tile("kimchi", name[len("kimchi."):]) for name in sec if name.startswith("kimchi.") and name not in ("kimchi.prev_prechallenges", "kimchi.prev_chals")}
    kp = m.MinaContext.make_kimchi_proofs(B, 2, NPUB, karr, statements=st)
    nd = min(B, distinct_chains or B)
    recs, nf, hashes = make_chains(ctx, nd, seed)
    ci = np.arange(B) % nd
    # folding randomisers of the kernel-level job are the CALLER's to supply (include/mina_verify.h): drawn from the OS CSPRNG, as the
    # reference-shaped boundary does per job (api_verify.hip draw_randomisers); their values do not change the work
    rho = np.frombuffer(os.urandom(B * 32), np.uint8).reshape(B, 32).copy(); rho[:, 31] &= 0x3F
    same = nd == B
    arrays = dict(state_records=recs.reshape(-1) if same else recs[ci].reshape(-1), state_nfields=nf.reshape(-1) if same else nf[ci].reshape(-1),
                  expected_hashes=hashes.reshape(-1) if same else hashes[ci].reshape(-1),
                  rand_base=fresh_randomiser(), sg_rand_base=fresh_randomiser(), acc_prechallenges=tile(None, "acc_prechallenges"), acc_sg=tile(None, "acc_sg"), acc_rho=rho.reshape(-1),
                  **{name: tile("opening", name) for name in ("lr", "delta", "sg", "z1", "z2")})
    scal = dict(with_states=1, with_ipa=1, with_accumulator=1, log2_domain=LOG2_DOMAIN, npub=NPUB, k=WRAP_K, n_evalpoints=NPTS, n_comms=NCOMMS + 2, acc_k=ACC_K, kimchi=kp)
    return m.MinaContext.make_state_jobs(B, arrays, **scal), kp, (recs[0].copy(), nf[0].copy(), hashes[0].copy()), {"chains": int(nd), "wrap_proofs": int(min(n, B))}


def device_jobs(m, hj, keep, kp, dev):
    """the job of `build_full_job` with every section resident in HBM: (mina_state_jobs with device pointers, its mina_kimchi_proofs or None, the torch tensors
    that own the buffers -- keep them alive)"""
    import ctypes
    import torch
    dj = m.lib.StateJobs()
    ctypes.memmove(ctypes.byref(dj), ctypes.byref(hj), ctypes.sizeof(m.lib.StateJobs))
    by_addr = {a.ctypes.data: a for a in keep if isinstance(a, np.ndarray)}
    dtensors = []
    for name in m.lib.StateJobs.POINTER_FIELDS:                # every section resident in HBM (torch owns the buffers)
        addr = getattr(hj, name)
        if addr:
            t = torch.from_numpy(np.array(by_addr[addr].view(np.uint8).reshape(-1))).to(dev)
            dtensors.append(t); setattr(dj, name, t.data_ptr())
    dk = None
    if kp is not None:                                         # the kimchi section's arrays live in HBM as well
        ct = ctypes
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
            dtensors.append(dst)
        dj.kimchi = ct.addressof(dk)
    return dj, dk, dtensors


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
    # the like-for-like figure: the SAME fold the GPU job does (kimchi batch_verify's shape) -- per-proof transcripts on all threads, then ONE MSM per curve
    # over the batch (composite_oracle.c oc_verify_folded); batch = 16 proofs per thread, randomisers from the OS CSPRNG per batch
    fbatch = [proofs[i % len(proofs)] for i in range(16 * threads)]
    okf, _ = C.verify_folded(fbatch[:threads], threads)                # warm
    frounds, t1 = 0, time.perf_counter()
    while True:
        ok_b, vf = C.verify_folded(fbatch, threads); okf = okf and ok_b and bool(vf.all()); frounds += 1
        elf = time.perf_counter() - t1
        if elf >= budget_s * 0.6:
            break
    folded = {"value": frounds * len(fbatch) / elf, "unit": "proofs/s", "cores": threads, "kind": "port", "proofs_per_batch": len(fbatch), "batches": frounds, "seconds": elf,
              "sample": f"{frounds} batches of {len(fbatch)} full Proof-of-State verifications, folded as the GPU job folds them: per-proof transcripts (17 state hashes, statement, "
                        f"kimchi oracles, opening transcript, b_poly coefficients) on {threads} threads, then ONE 2^15 + {len(fbatch)} x 81-point Pallas MSM and ONE 2^16 Vesta MSM per batch "
                        f"(ark-style Pippenger, a thread per window); verdict ACCEPT: {okf}",
              "note": "the reference's FFI verifies one proof per call (cpu_baseline: two full MSMs per proof); this is the same algorithm with the batch fold of kimchi batch_verify, "
                      "the shape the GPU headline is measured on"}
    return {"value": total / el, "unit": "proofs/s", "cores": threads, "kind": "port", "implementation": "native C restatement (oracle/composite_oracle.c), pthreads across proofs", "folded": folded,
            "cpu_model": cpu_model(), "nproc": nproc, "usable_cores": usable_cores(), "single_thread_value": 1.0 / one_s, "setup_s": setup_s,
            "sample": f"{total} full Proof-of-State verifications (BASELINE config C1 = the bench's own job per proof: 17 state hashes + Pickles statement -> public inputs + "
                      f"public-input commitment + kimchi oracles/to_batch + k=15 wrap opening check + 2^16 Vesta accumulator) in {el:.1f} s on {threads} threads, one proof per "
                      f"thread at a time; native C restatement of the o1-labs / arkworks algorithms (NOT the Rust reference, which cannot be built here); verdict ACCEPT: {ok}"}


def _boundary_setup(m, devices: str):
    """the verifier process of the boundary legs: contexts on `devices` ("3" = GPU 3; "0,1,2,3" = one context per GPU, a call's proofs cut into
    contiguous shards, api_verify.hip verify_state_many), the fixture's synthetic indexes on every device.  Returns (lib, items, ndev) or a skip record."""
    path = os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")
    if not os.path.exists(path):
        return {"skipped": "tests/golden/state_proofs_k15_bytes.json missing"}
    fxb = json.load(open(path))
    fx, un = load_encoded_fixture()
    import mina_bridge_amd.poseidon_params as PP
    if fxb["poseidon_constants"] != PP.NAME:
        return {"skipped": "byte fixture minted under another Poseidon constant set"}
    os.environ["MINA_VERIFY_DEVICES"] = devices
    ndev = len(devices.split(","))
    m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)            # the compiled-in Poseidon tables are the surrogate set (named in the output)
    assert m.lib.verify_device_count() == ndev
    install_fixture_indexes(m.lib.verify_all_devices(), fx, un)
    items = [(bytes.fromhex(it["proof"]), bytes.fromhex(it["pub"])) for it in fxb["proofs"]]
    return m.load_library(), items, ndev


_CALLERS_C = r"""
#include <pthread.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
bool mina_verify_state(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len);
struct job { const uint8_t *const *proofs; const size_t *pl; const uint8_t *const *pubs; const size_t *ql; int ncases, calls, t; long bad; };
static void *worker(void *a) { struct job *j = a; for (int k = 0; k < j->calls; ++k) { int c = (j->t + k) % j->ncases; if (!mina_verify_state(j->proofs[c], j->pl[c], j->pubs[c], j->ql[c])) j->bad++; } return 0; }
long run_callers(int nthreads, int calls, int ncases, const uint8_t *const *proofs, const size_t *pl, const uint8_t *const *pubs, const size_t *ql) {
  pthread_t th[1024]; struct job jobs[1024]; long bad = 0;
  if (nthreads > 1024) return -1;
  for (int t = 0; t < nthreads; ++t) { jobs[t] = (struct job){proofs, pl, pubs, ql, ncases, calls, t, 0}; pthread_create(&th[t], 0, worker, &jobs[t]); }
  for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], 0); bad += jobs[t].bad; }
  return bad; }
"""


def one_proof_per_call(m, items, thread_counts=(1, 64, 1024), per_thread=4):
    """The reference's real call shape: `verify_mina_state_ffi` / `mina_verify_state` takes ONE proof per call (/root/reference README.md:277-279; the caller is
    core/src/aligned.rs:31-58 through Aligned's operator tasks).  N caller threads -- pthreads of a small C helper, no interpreter lock between a verdict and the next call --
    each make `per_thread` calls; calls that arrive while a job runs leave together as the next job (api_verify.hip group commit).  Same measurement as
    tools/concurrent_callers.py, in the line so that the driver times it."""
    import ctypes, subprocess, tempfile
    try:
        tmp = tempfile.mkdtemp()
        with open(os.path.join(tmp, "callers.c"), "w") as f: f.write(_CALLERS_C)
        libdir = os.path.dirname(m.LIB_PATH)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(tmp, "callers.c"), "-o", os.path.join(tmp, "libcallers.so"), "-L", libdir, "-lminaverify",
                               "-Wl,-rpath," + libdir], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        helper = ctypes.CDLL(os.path.join(tmp, "libcallers.so")); helper.run_callers.restype = ctypes.c_long
        nc = len(items)
        c_pr = (ctypes.c_char_p * nc)(*[c[0] for c in items]); c_pl = (ctypes.c_size_t * nc)(*[len(c[0]) for c in items])
        c_pu = (ctypes.c_char_p * nc)(*[c[1] for c in items]); c_ql = (ctypes.c_size_t * nc)(*[len(c[1]) for c in items])
        rows = {}
        for nt in thread_counts:
            calls = per_thread * (2 if nt == 1 else 1)
            assert helper.run_callers(nt, 2, nc, c_pr, c_pl, c_pu, c_ql) == 0
            t0 = time.perf_counter()
            bad = helper.run_callers(nt, calls, nc, c_pr, c_pl, c_pu, c_ql)
            dt = time.perf_counter() - t0
            assert bad == 0, "a valid proof was rejected"
            rows[str(nt)] = {"caller_threads": nt, "calls": nt * calls, "proofs_per_s": nt * calls / dt, "ms_per_call_seen_by_a_caller": dt / calls * 1e3}
        return {"entry_point": "mina_verify_state (= verify_mina_state_ffi's path): ONE serialized proof per call, pthread callers", "by_caller_threads": rows,
                "note": "the reference's call shape; the batch entry points above are this library's addition (INTEGRATION.md)"}
    except Exception as e:                                            # noqa: BLE001 -- a secondary leg (needs gcc on the box): never takes the line down
        return {"error": repr(e)[:300]}


class _Batch:
    """n serialized proofs (the fixture's four, tiled) as the pointer / length arrays of `mina_verify_state_batch`"""
    def __init__(self, lib, items, n):
        import ctypes
        self.lib, self.n, self.items = lib, n, items
        self.P = [items[i % len(items)][0] for i in range(n)]; self.Q = [items[i % len(items)][1] for i in range(n)]
        self.PP = (ctypes.c_char_p * n)(*self.P); self.PL = (ctypes.c_size_t * n)(*map(len, self.P))
        self.QQ = (ctypes.c_char_p * n)(*self.Q); self.QL = (ctypes.c_size_t * n)(*map(len, self.Q))
        self.out = np.zeros(n, np.uint8)

    def call(self, out=None):
        import ctypes
        out = self.out if out is None else out
        rc = self.lib.mina_verify_state_batch(ctypes.c_size_t(self.n), self.PP, self.PL, self.QQ, self.QL, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, self.lib.mina_last_error().decode()

    def tamper_check(self, positions):
        """a tampered public input at each of `positions`: exactly those verdict bytes are 0 (the culprit search of their chunks), untimed"""
        bad = bytearray(self.items[0][1]); bad[40] ^= 1
        for pos in positions: self.QQ[pos] = bytes(bad)
        self.call()
        got = np.flatnonzero(self.out == 0).tolist()
        assert got == sorted(positions), f"tampered public inputs at {sorted(positions)} must fail exactly those proofs: rejected {got[:16]}"
        for pos in positions: self.QQ[pos] = self.Q[pos]

    def timed(self, min_seconds, min_calls=1):
        calls, t0 = 0, time.perf_counter()
        while True:
            self.call(); calls += 1
            el = time.perf_counter() - t0
            if el >= min_seconds and calls >= min_calls:
                break
        assert self.out.all()
        return {"value": calls * self.n / el, "unit": "proofs/s", "proofs_per_call": self.n, "ms_per_call": el / calls * 1e3, "calls": calls}


def boundary_leg(m, devices: str, B: int, min_seconds: float = 2.0):
    """Secondary key `boundary_bytes_to_bools`: the reference-shaped boundary itself -- `mina_verify_state_batch` over B full-size bincode
    `MinaStateProof`s + 1057-byte public inputs (tests/golden/state_proofs_k15_bytes.json: the headline's four proofs as the caller of
    core/src/aligned.rs:31-58 would send them, tiled to B), host bytes in, verdict bytes out: parsing, `to_input` flattening, the ledger and
    consensus checks, the page-locked staging, PCIe both ways and the GPU job, back-to-back synchronous calls for >= `min_seconds`; then the
    same with two / four caller threads (a batcher's tasks), one call of 8 B proofs, and BASELINE config C5's batch (4096 per call).  Never `value`."""
    import threading
    setup = _boundary_setup(m, devices)
    if isinstance(setup, dict):
        return setup
    lib, items, ndev = setup
    job = _Batch(lib, items, B)
    for _ in range(3):                                                # warm: contexts, tables, slots' page-locked buffers
        job.call()
    assert job.out.all(), "boundary verdicts must be ACCEPT"
    job.tamper_check([B // 3])
    single = job.timed(min_seconds)
    # BASELINE config C5's batch through the boundary: 4096 serialized proofs per call.  Timed HERE, while the process holds one slot's streams: after the caller-thread
    # legs below it holds 4 slots x (lane + three leg streams + upload stream) > its 16 hardware queues and every later call is slower for it (the same call: 35 ms here,
    # 44 ms at the end of this leg -- `c5_4096_per_call_after_culprit_searches` keeps that figure)
    c5 = None
    if B >= 4096:
        c5_job = _Batch(lib, items, 4096)
        for _ in range(2):
            c5_job.call()
        c5 = c5_job.timed(1.0, min_calls=3)

    def callers(k):
        outs = [np.zeros(B, np.uint8) for _ in range(k)]
        counts = [0] * k; stop = [time.perf_counter() + 1e9]
        def worker(i):
            while time.perf_counter() < stop[0]:
                job.call(outs[i]); counts[i] += 1
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
        big_job = _Batch(lib, items, 8 * B)
        big_job.call()
        big = big_job.timed(1.0, min_calls=3)
    # What ONE bad opening per call costs everybody: a folded check that fails sends its chunk through the culprit search -- until round 5 with the device
    # drained and its lock held for the length of the search; now on a view context of the device (api_verify.hip Device::sc).  Caller 1 sends B good proofs per call; caller 2 sends B proofs with ONE bad opening (z1 with a flipped bit: the folded
    # opening check of its chunk fails, the search finds exactly that proof); both for min_seconds.  Reported: the clean caller's rate beside it, and the searcher's call time.
    search_cost = None
    try:
        flat, used = m.lib.wrap_proof_flatten(items[0][0], m.lib.ENC_BINCODE, exact=False)
        z1 = bytes(flat[-192:-160])                                   # the flat form ends with z1 z2 delta sg (include/mina_verify.h)
        at = items[0][0].find(z1)
        if at >= 0 and items[0][0].find(z1, at + 1) < 0:
            badp = bytearray(items[0][0]); badp[at] ^= 1
            bad_job = _Batch(lib, items, B)
            bad_pos = (B // 2) - ((B // 2) % len(items))                  # an entry that holds items[0]
            bad_job.PP[bad_pos] = bytes(badp); bad_job.PL[bad_pos] = len(badp)
            bad_job.call()
            assert np.flatnonzero(bad_job.out == 0).tolist() == [bad_pos], "a bad opening must fail exactly its proof (culprit search)"
            stop = [False]; counts = [0, 0]; lat = []
            def clean_w():
                o = np.zeros(B, np.uint8)
                while not stop[0]:
                    job.call(o); counts[0] += 1
            def bad_w():
                o = np.zeros(B, np.uint8)
                while not stop[0]:
                    t_ = time.perf_counter(); bad_job.call(o); lat.append(time.perf_counter() - t_); counts[1] += 1
            th = [threading.Thread(target=clean_w), threading.Thread(target=bad_w)]
            t0 = time.perf_counter()
            for t in th: t.start()
            time.sleep(min_seconds); stop[0] = True
            for t in th: t.join()
            el = time.perf_counter() - t0
            search_cost = {"clean_caller_beside_a_searching_caller": counts[0] * B / el, "clean_caller_beside_a_clean_caller": two["value"] / 2, "unit": "proofs/s",
                           "searching_caller_ms_per_call": sorted(lat)[len(lat) // 2] * 1e3 if lat else None, "searching_caller_calls": counts[1], "proofs_per_call": B,
                           "note": "one bad opening per call of the second caller: its chunk's folded check fails and the 4-way culprit search runs -- since round 5 on a VIEW context of "
                                   "its device (own lanes and lock: mina_verify_tuning.search_ctx = 1; round 4 drained the device and held its lock: 32 - 51 k proofs/s left here); "
                                   "the first column is what a well-behaved caller keeps beside it"}
    except Exception as e:                                            # noqa: BLE001 -- a diagnostic leg: never takes the line down
        search_cost = {"error": repr(e)[:300]}
    # ... and does a process that HAS searched stay slower?  (Round 3 found exactly that with a 32-way fan-out: more streams than hardware queues, for the life of the
    # process.)  BASELINE C5's call again, after the searches above.  Round 5: yes while a search created its own fan-out streams (45 -> 52 - 54 ms, either search form);
    # not since it borrows the failed chunk's idle streams (45.5 -> 43.7, 45.9 -> 46.0 ms).
    c5_after = None
    if c5 is not None and search_cost and "error" not in search_cost:
        c5_after = c5_job.timed(1.0, min_calls=3)
    res = {"value": single["value"], "unit": "proofs/s", "proofs_per_call": B, "ms_per_call": single["ms_per_call"], "calls_timed": single["calls"],
           "bytes_per_proof": len(job.P[0]) + len(job.Q[0]), "host_threads": int(os.environ.get("MINA_HOST_THREADS", min((os.cpu_count() or 2) // 2, 64))), "usable_cores": usable_cores(),
           "two_caller_threads": two, "four_caller_threads": four, "one_bad_opening_per_call": search_cost, "one_call_of_65536": big, "c5_4096_per_call": c5, "c5_4096_per_call_after_culprit_searches": c5_after, "devices": devices,
           "entry_point": "mina_verify_state_batch (include/mina_verify.h): bincode MinaStateProof + MinaStatePubInputs bytes -> verdict bytes",
           "poseidon_constants": m.lib.poseidon_params_name(), "process": "a fresh process holding only libminaverify.so (no torch): the operator's verifier process",
           "note": "host bytes in, bools out: parsing, to_input flattening, ledger + consensus checks, pinned staging, PCIe both ways, the GPU job (folding "
                   "randomisers from the OS CSPRNG per chunk); one tampered proof in a warm-up call failed alone"}
    m.lib.verify_shutdown()
    return res


def c2_leg(m, device: str, bdf: str = ""):
    """Secondary key `c2_accumulator_only` -- BASELINE config C2 (round 1's headline), in a process of its own that holds only the library: (a) `single_check`: C2 AS WRITTEN
    (/root/reference README.md:534-544: "a single Kimchi proof batch_verify, 2^16-base Vesta IPA MSM") -- ONE un-folded accumulator check per call, one call at a time, nothing
    else on the chip: wall time per call, the sum of its kernels' stage events, the MSM's algorithmic GB/s; (b) the pipelined form: 8 independent checks per call over 16 lanes.
    (Until round 6 this ran at the end of the main process: with the forked lanes' helper streams that process holds more streams than hardware queues by then, and every later
    launch is slower for it -- 12.6 k checks/s there against 13.5 k in a fresh process, same box.)"""
    ctx = m.MinaContext(int(device))
    for f in (0, 1):
        ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
    ctx.srs_create(CURVE_VESTA, 1 << 16)
    pre8, sg8 = make_accumulators(ctx, 8, 4242)
    d_pre8 = ctx.dev_upload(ctx.dev_malloc(pre8.size), pre8.reshape(-1)); d_sg8 = ctx.dev_upload(ctx.dev_malloc(sg8.size), sg8.reshape(-1))
    d_v8 = ctx.dev_upload(ctx.dev_malloc(32), np.zeros(32, np.uint8))
    verdicts = lambda: ctx.dev_download(d_v8, 32).view(np.uint32).tolist()
    one = lambda: ctx.accumulator_check_dev(CURVE_VESTA, ACC_K, 1, d_pre8, d_sg8, 0, d_v8)
    for _ in range(8):
        one()
    ctx.synchronize()
    ctx.prof_enable(0x7ff)                                        # every MSM and b_poly stage (ctx.h ProfStage 0 .. 10)
    n1 = 64; t1 = time.perf_counter()
    for _ in range(n1):
        one(); ctx.synchronize()
    wall_us = (time.perf_counter() - t1) / n1 * 1e6
    p1 = ctx.prof_read(); ctx.prof_enable(0)
    assert verdicts()[0] == 1
    kern_us_1 = sum(ms for _, ms in p1.values()) / n1 * 1e3
    msm_bytes = 65536 * (64 + 32) + 96
    single = {"wall_us": wall_us, "kernel_us_sum": kern_us_1, "stages_us": {k: v[1] / n1 * 1e3 for k, v in p1.items() if v[0]},
              "algorithmic_bytes": msm_bytes, "algorithmic_GBps_wall": msm_bytes / wall_us / 1e3, "algorithmic_GBps_kernels": msm_bytes / kern_us_1 / 1e3 if kern_us_1 else None,
              "frac_of_hbm_peak_kernels": msm_bytes / kern_us_1 / 1e3 / HBM_PEAK_GBPS if kern_us_1 else None, **msm_single_traffic(kern_us_1),
              "note": "ONE un-folded 2^16 Vesta accumulator check per call, one call at a time, nothing else on the GPU (mina_accumulator_check_dev, batch 1: prechallenges -> "
                      "b_poly coefficients -> fixed-base MSM in its one-MSM task form -> comparison); kernel_us_sum = HIP-event stage times on the lane's stream"}
    ctx.set_pipeline(16)
    c2 = lambda: ctx.accumulator_check_multi_dev(CURVE_VESTA, ACC_K, 8, d_pre8, d_sg8, d_v8)
    for _ in range(32):
        c2()
    ctx.synchronize()
    sampler = PowerSampler(bdf or None, interval=0.03).start(); t2 = time.perf_counter()
    for _ in range(400):
        c2()
    ctx.synchronize()
    rate = 8 * 400 / (time.perf_counter() - t2)
    power = sampler.stop()
    assert verdicts() == [1] * 8
    ctx.close()
    return {"value": rate, "single_check": single, "power": power}


def one_proof_leg(m, devices: str):
    """`one_proof_per_call` in a process of its own (the operator's verifier process right after start-up: indexes installed, nothing else has run)"""
    setup = _boundary_setup(m, devices)
    if isinstance(setup, dict):
        return setup
    lib, items, ndev = setup
    out = one_proof_per_call(m, items)
    m.lib.verify_shutdown()
    return out


def account_leg(m, devices: str, min_seconds: float = 1.5):
    """Secondary key `c4_account_256` -- BASELINE config C4: 256 Proof-of-Account verifications per `mina_verify_account_batch` call (the reference's
    verify_account_inclusion_ffi per pair, /root/reference/README.md:358-362; containers core/src/proof/account_proof.rs:9-35), host bytes in, bools out: bincode parsing,
    the Solidity-ABI cross-check (core/src/sol/account.rs:25-314), four dependent Poseidon stages of the account hash and the depth-35 Merkle fold on the GPU.
    256 DISTINCT accounts (tests/golden/account_proofs_bytes.json).  Timed: a lone caller; 16 caller threads (a batcher's tasks: small calls share jobs); and the
    config's "MSM mix" -- 4 account callers beside 2 callers of 8192 serialized state proofs in the same process."""
    import base64
    import ctypes
    import threading
    path = os.path.join(ROOT, "tests", "golden", "account_proofs_bytes.json")
    if not os.path.exists(path):
        return {"skipped": "tests/golden/account_proofs_bytes.json missing"}
    fxa = json.load(open(path))
    setup = _boundary_setup(m, devices)
    if isinstance(setup, dict):
        return setup
    lib, items, ndev = setup
    import mina_bridge_amd.poseidon_params as PP
    if fxa["poseidon_constants"] != PP.NAME:
        return {"skipped": "account fixture minted under another Poseidon constant set"}
    n = 256
    P = [base64.b64decode(fxa["proofs"][i % len(fxa["proofs"])]["proof"]) for i in range(n)]; Q = [base64.b64decode(fxa["proofs"][i % len(fxa["proofs"])]["pub"]) for i in range(n)]
    PP_ = (ctypes.c_char_p * n)(*P); PL = (ctypes.c_size_t * n)(*map(len, P)); QQ = (ctypes.c_char_p * n)(*Q); QL = (ctypes.c_size_t * n)(*map(len, Q))

    def call(out):
        rc = lib.mina_verify_account_batch(ctypes.c_size_t(n), PP_, PL, QQ, QL, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, lib.mina_last_error().decode()
    out0 = np.zeros(n, np.uint8)
    for _ in range(3):
        call(out0)
    assert out0.all(), "account verdicts must be ACCEPT"
    bad = bytearray(Q[7]); bad[3] ^= 1; QQ[7] = bytes(bad)                # another ledger hash for pair 7: it alone fails
    call(out0); assert np.flatnonzero(out0 == 0).tolist() == [7], "a tampered ledger hash must fail exactly its pair"
    QQ[7] = Q[7]

    def callers(k_acc, k_state, seconds):
        stop = [False]; ca = [0] * k_acc; cs = [0] * k_state; lat = [[] for _ in range(k_acc)]
        sjob = _Batch(lib, items, 8192) if k_state else None
        if sjob:
            sjob.call()
        errs = []
        def aw(i):
            o = np.zeros(n, np.uint8)
            try:
                while not stop[0]:
                    t = time.perf_counter(); call(o); lat[i].append(time.perf_counter() - t); ca[i] += 1
                    if not o.all(): raise AssertionError("an account pair was rejected")
            except Exception as e:                                   # a thread's exception does not reach the caller by itself
                errs.append(repr(e)); stop[0] = True
        def sw(i):
            o = np.zeros(8192, np.uint8)
            try:
                while not stop[0]:
                    sjob.call(o); cs[i] += 1
                    if not o.all(): raise AssertionError("a state proof was rejected")
            except Exception as e:
                errs.append(repr(e)); stop[0] = True
        th = [threading.Thread(target=aw, args=(i,)) for i in range(k_acc)] + [threading.Thread(target=sw, args=(i,)) for i in range(k_state)]
        t0 = time.perf_counter()
        for t in th: t.start()
        time.sleep(seconds); stop[0] = True
        for t in th: t.join()
        el = time.perf_counter() - t0
        assert not errs, errs[:2]
        al = sorted(x for l in lat for x in l)
        r = {"value": sum(ca) * n / el, "unit": "account proofs/s", "caller_threads": k_acc, "calls": sum(ca), "ms_per_call_median": al[len(al) // 2] * 1e3 if al else None}
        if k_state:
            r["state_caller_threads"] = k_state; r["state_proofs_per_s"] = sum(cs) * 8192 / el
        return r
    callers(16, 0, 0.3)                                              # warm: every slot's page-locked staging
    lone, many = callers(1, 0, min_seconds), callers(16, 0, min_seconds)
    callers(4, 2, 0.5)
    mix = callers(4, 2, 2 * min_seconds)
    res = {"value": lone["value"], "unit": "account proofs/s", "proofs_per_call": n, "distinct_accounts": len(fxa["proofs"]), "merkle_depth": fxa["merkle_depth"],
           "ms_per_call": lone["ms_per_call_median"], "lone_caller": lone, "sixteen_caller_threads": many, "beside_state_batches": mix,
           "bytes_per_pair": (sum(map(len, P)) + sum(map(len, Q))) // n, "devices": devices,
           "entry_point": "mina_verify_account_batch (include/mina_verify.h): bincode MinaAccountProof + MinaAccountPubInputs bytes -> verdict bytes",
           "poseidon_constants": m.lib.poseidon_params_name(),
           "note": "BASELINE C4.  Poseidon only (a16 has no MSM): ~95 dependent permutations per pair, so ONE 256-pair call is a latency-bound chain on a mostly idle chip; "
                   "callers that overlap share jobs (group commit).  beside_state_batches: the config's 'MSM mix' -- the account jobs ride a lane of their own at raised wave priority "
                   "beside 2 callers x 8192 full state proofs; one tampered pair in a warm-up call failed alone"}
    m.lib.verify_shutdown()
    return res


def boundary_all_devices_leg(m, devices: str, B: int, min_seconds: float = 2.0):
    """`boundary_bytes_to_bools.all_devices`: the PRODUCT's multi-GPU path (SURVEY.md 8e.1 behind the C-ABI) -- ONE process, one context per device of
    `devices`, `mina_verify_state_batch` cuts each call's proofs into contiguous shards, one host thread + pipeline per device, verdict bytes gathered on
    the host (api_verify.hip verify_state_many; no collective: the shards' folded checks are independent).  Two call sizes: ndev x B proofs (every device
    a full chunk) and BASELINE config C5 as written -- 4096 serialized proofs per call over the devices.  A tampered proof in every shard fails alone."""
    setup = _boundary_setup(m, devices)
    if isinstance(setup, dict):
        return setup
    lib, items, ndev = setup
    n_big = ndev * B
    big = _Batch(lib, items, n_big)
    for _ in range(3):
        big.call()
    assert big.out.all(), "boundary verdicts must be ACCEPT"
    big.tamper_check([g * B + B // 3 for g in range(ndev)])           # one per shard
    full = big.timed(min_seconds, min_calls=3)
    c5_job = _Batch(lib, items, 4096)
    c5_job.call(); c5_job.call()
    c5_job.tamper_check([4096 * g // ndev + 5 for g in range(ndev)])
    c5 = c5_job.timed(min_seconds / 2, min_calls=3)
    res = {"devices": devices, "n_devices": ndev, "distinct_gpus": len(set(devices.split(","))), "value_all_devices": full["value"], "unit": "proofs/s",
           "proofs_per_call": n_big, "ms_per_call": full["ms_per_call"], "calls": full["calls"], "c5_4096_per_call": c5,
           "entry_point": "mina_verify_state_batch with $MINA_VERIFY_DEVICES = the devices: contiguous shards, one pipeline per device, no collective",
           "note": "one tampered proof per shard in a warm-up call failed alone" + ("" if len(set(devices.split(","))) == ndev else
                   "; LOGICAL contexts on one GPU (a box with fewer GPUs than --gpus): the sharding code path, not a scaling figure")}
    m.lib.verify_shutdown()
    return res


def physical_gpus() -> int:
    """GPUs this process can see, counted in a child process (the launcher itself never initialises HIP: a process that holds a runtime on the GPUs
    beside the ranks is not the configuration measured)"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().split()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    except (subprocess.TimeoutExpired, ValueError):
        return 0


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no torch.distributed launcher around it: start the N ranks here -- what `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1` would do -- and pass rank 0's JSON line through.  Fewer than N GPUs on the box: the ranks share GPU 0
    (MINA_BENCH_SHARE_GPU=1: gloo rendezvous, the line says "shared_gpu": true) so that the N > 1 path still runs end to end."""
    import socket
    import subprocess
    n = args.gpus
    have = physical_gpus()
    if have < 1:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MINA_BENCH_GPUS_PHYSICAL=str(have), MINA_BENCH_LAUNCHER="bench.py launch_ranks")
        if have < n:
            env["MINA_BENCH_SHARE_GPU"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies takes the others down with it (they would wait for it at the next collective until the backend's timeout)
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read())); reader.start()
    failed = None
    while failed is None and any(p_.poll() is None for p_ in procs):
        for p_ in procs:
            if p_.poll() not in (None, 0): failed = p_.returncode
        time.sleep(0.2)
    if failed is not None:
        for p_ in procs:
            if p_.poll() is None: p_.terminate()                # the exact children started above
    rcs = [p_.wait() for p_ in procs]
    reader.join()
    sys.stdout.write(out0[0] if out0 else ""); sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    # 16384 proofs per step (round 4; rounds 2 - 3: 8192): --steps 20 over 20 lanes 8192 -> 261 k, 16384 -> 275 k, 24576 -> 273 k, 32768 -> 275 k proofs/s; --steps 80: 258 k -> 267 k.
    # The timed region of the driver's --steps 20 is then 1.2 s instead of 0.6 s.  The bytes -> bools legs keep 8192 per call (the boundary's own chunk size).
    ap.add_argument("--jobs", type=int, default=16384, help="state proofs per step (one mina_state_job_batch_dev call)")
    # 20 lanes under 24 hardware queues (round 4 sweep, full mode, one MI355X): --steps 20: 10 lanes 247.5 k, 16: 245.9 k, 20: 262 - 267 k, 24: 250.5 k, 32: 248.2 k proofs/s;
    # --steps 80: 16 lanes 256.2 k, 20: 258.3 k, 24: 242.7 k, 32: 250.9 k (with 32 - 40 queues nothing gains: 20 lanes 258.1 k, 32 lanes 228 - 243 k)
    # at 16384 proofs per step (final build, --steps 40): 12 lanes 273.1 k (29 GiB of HBM in use), 16: 275.3 k (38 GiB), 20: 279.7 / 280.8 k (48 GiB), 24: 260.7 k (56 GiB)
    # round 6: the three legs of a job run on streams of their own (mina_verify_tuning.dev_fork) and every kernel but the state hash raises its wave priority: 4 lanes fill the
    # chip (same box, 16384 per step: 4 lanes 307 k at 13 GiB, 3: 304 k at 10 GiB, 5: 309 k, 8: 297 - 300 k; the round-5 form, one stream per job: 20 lanes 291 k at 50 GiB -- profiles/r06_dev_fork.md)
    ap.add_argument("--pipeline", type=int, default=4, help="internal stream lanes over which consecutive steps are issued")
    ap.add_argument("--distinct-chains", type=int, default=0, help="distinct candidate chains in the batch (default 0: one per proof)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("full", "kimchi", "prepared"), default="full",
                    help="full (default): the whole verifier from parsed proofs -- Pickles statement -> public inputs, kimchi oracles + to_batch, opening check, "
                         "accumulator, 17 state hashes (fixture tests/golden/statement_k15.json, 4 distinct proofs); kimchi: the same without the statement stage "
                         "(public inputs given; tests/golden/kimchi_k15.json); prepared: pre-derived BatchEvaluationProof rows (round 2's first headline)")
    ap.add_argument("--kimchi", action="store_true", help="alias of --mode kimchi")
    ap.add_argument("--no-probes", action="store_true", help="skip the isolated-kernel / C2 probes and the sustained / C5 legs (profiling runs: only the timed loop launches kernels)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the bytes -> bools legs (mina_verify_state_batch on serialized proofs)")
    ap.add_argument("--boundary-jobs", type=int, default=0, help="proofs per call of the bytes -> bools legs (default: min(--jobs, 8192))")
    ap.add_argument("--power-trace", default="", help="write the timed region's raw (seconds, watts, MHz) hwmon samples to this file (sampled every 10 ms instead of 200)")
    ap.add_argument("--preflight", action="store_true",
                    help="the check to run FIRST on a multi-GPU node (VERDICT r04 next #7; no node was available to any round): every rank reports LOCAL_RANK, the device it bound and "
                         "that device's PCI bus id, the rank count the collectives backend saw, runs ONE small step + the verdict all-gather; rank 0 prints one JSON line and the "
                         "exit code is non-zero if two ranks share a device without shared_gpu, the backend saw != N ranks, or a verdict is wrong.  No throughput figure.")
    args = ap.parse_args()
    if args.preflight:
        args.steps, args.warmup, args.jobs, args.pipeline, args.no_probes, args.no_boundary, args.no_cpu_baseline = 1, 1, min(args.jobs, 64), 1, True, True, True
    if args.kimchi:
        args.mode = "kimchi"
    args.kimchi = args.mode != "prepared"                      # the wrap leg starts from the raw proof in both non-prepared modes

    if "WORLD_SIZE" not in os.environ and os.environ.get("MINA_BENCH_FORCE_DIST") == "1":
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:     # no launcher around this process: be the launcher
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MINA_BENCH_SHARE_GPU=1 (set by launch_ranks on a box with fewer GPUs than ranks, and by the tests): every rank uses GPU 0 and the ranks rendezvous
    # over gloo -- the N > 1 code path (barriers, MAX over ranks, verdict all-gather, aggregate value) end to end on a 1-GPU box
    share_gpu = os.environ.get("MINA_BENCH_SHARE_GPU") == "1"
    dist_on = world > 1 or os.environ.get("MINA_BENCH_FORCE_DIST") == "1"      # (test hook: the collectives of the N > 1 path on a 1-rank RCCL group)
    if dist_on:                                              # control plane first, on the CPU: the other ranks wait here while rank 0 runs the boundary legs
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1 and not share_gpu and torch.cuda.device_count() < world:      # an outside launcher (torch.distributed.run) on a box with fewer GPUs than ranks:
            share_gpu = True                                                       # every rank sees the same count and takes the same turn
            os.environ["MINA_BENCH_SHARE_GPU"] = "1"; os.environ.setdefault("MINA_BENCH_GPUS_PHYSICAL", str(torch.cuda.device_count()))
        if share_gpu and world > 4:                          # 20 lanes of 16384 proofs hold ~48 GiB per rank: eight ranks on ONE GPU do not fit 288 GB; fewer lanes each
            args.pipeline = max(4, min(args.pipeline, 100 // world))
        dist.init_process_group("gloo" if share_gpu else "cpu:gloo,cuda:nccl")
    # The bytes -> bools leg runs FIRST, in a process of its own that holds only the library (no torch): the operator's verifier process.  It must
    # not share the GPU with another process's queues (a second process holding 24 hardware queues makes the scheduler time-slice them: the
    # same call took 288 ms instead of 55), and inside THIS process the 40-odd streams of the headline's context would populate the runtime's
    # queue pool first (67 - 72 ms).  GPU_MAX_HW_QUEUES=16 there: with the system runtime a lone job's three legs overlap best at <= 16 (54.6 ms;
    # 24: 68.4 ms), while the 20-lane pipeline of the headline below wants one queue per lane plus a few (24).
    # At N > 1 rank 0 runs a second leg the same way: the product's own multi-device path, ONE process with a context on each of the N GPUs
    # (boundary_all_devices_leg).  The other ranks have not touched their GPUs yet: they wait at the CPU barrier below.
    boundary = c4 = one_pc = c2_own = None
    if not args.no_boundary and rank == 0:
        import subprocess
        env = dict(os.environ); env["GPU_MAX_HW_QUEUES"] = os.environ.get("MINA_BOUNDARY_HW_QUEUES", "16")
        for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"): env.pop(k_, None)
        def leg(*argv):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv), capture_output=True, text=True, timeout=900, env=env)
                return json.loads(r.stdout.strip().split("\n")[-1]) if r.returncode == 0 and r.stdout.strip() else {"error": (r.stderr or r.stdout)[-400:]}
            except (subprocess.TimeoutExpired, ValueError) as e:
                return {"error": repr(e)[:400]}
        bsize = min(args.jobs, args.boundary_jobs) if args.boundary_jobs else min(args.jobs, 8192)
        one_pc = leg("--one-proof-only", "0" if share_gpu else str(local_rank), "0")       # first: a verifier process right after start-up
        boundary = leg("--boundary-only", "0" if share_gpu else str(local_rank), str(bsize))
        c4 = leg("--account-only", "0" if share_gpu else str(local_rank), "256")
        if world > 1:
            devs = ",".join("0" if share_gpu else str(g) for g in range(world))
            boundary["all_devices"] = leg("--boundary-all-devices", devs, str(bsize))
            boundary["value_all_devices"] = boundary["all_devices"].get("value_all_devices")
    if not args.no_probes and rank == 0 and not (dist_on and share_gpu):
        import subprocess
        env2 = dict(os.environ)
        for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"): env2.pop(k_, None)
        try:
            r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--c2-only", str(local_rank), "-"], capture_output=True, text=True, timeout=600, env=env2)
            c2_own = json.loads(r_.stdout.strip().split("\n")[-1]) if r_.returncode == 0 and r_.stdout.strip() else {"error": (r_.stderr or r_.stdout)[-400:]}
        except (subprocess.TimeoutExpired, ValueError) as e:
            c2_own = {"error": repr(e)[:400]}
    if dist_on:
        dist.all_reduce(torch.zeros(1))                      # CPU tensor -> gloo: rank 0 arrives when its boundary legs are done

    import torch
    import mina_bridge_amd as m

    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} under a launcher with WORLD_SIZE={world}: reporting the {world} rank(s) that run", file=sys.stderr)
        args.gpus = world                      # the number of ranks actually running is what is reported
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    if os.environ.get("MINA_TUNE"):                            # experiments only (tools/dev_fork_sweep.sh): fields of mina_verify_tuning over the defaults; named in the line
        m.lib.tune_from_string(os.environ["MINA_TUNE"])
    ctx = m.MinaContext(local_rank)
    for f in (0, 1):
        ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
    ctx.srs_create(CURVE_VESTA, 1 << 16)                       # both SRS regenerated on the GPU (K4) + window tables
    ctx.srs_create(CURVE_PALLAS, 1 << 16)
    B = args.jobs
    seed = 0x6D696E61 + rank
    distinct = {}
    if args.mode == "full":
        (hj, keep), kp, chain_sample, distinct = build_full_job(ctx, m, B, seed, args.distinct_chains)
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
    dj, dk, dtensors = device_jobs(m, hj, keep, kp, dev)
    ctx.state_jobs_prepare(LOG2_DOMAIN, NPUB)
    ctx.set_pipeline(args.pipeline)
    nslots = max(args.pipeline, 8)                             # output slots: one per call in flight (the C5 leg below runs 6 lanes)
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

    n_warm = max(args.warmup, args.pipeline, 1)                          # every lane allocates its workspace during warm-up
    for _ in range(n_warm):
        step()
    ctx.synchronize()
    assert verdicts_ok(n_warm), "warm-up verdicts must be ACCEPT"
    for o in d_out:
        o.zero_()

    def barrier():
        if dist_on:
            if share_gpu: dist.barrier()
            else: dist.barrier(device_ids=[local_rank])          # RCCL

    gathered = None
    mask = sum(1 << b for b in PROF_STAGES.values())
    ctx.prof_enable(mask)                                      # HIP events around the candidate dominant kernels, on their lane streams
    try:
        pr_ = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr_, "pci_domain_id", 0), pr_.pci_bus_id, pr_.pci_device_id) if hasattr(pr_, "pci_bus_id") else None
    except Exception:
        bdf = None
    sampler = PowerSampler(bdf, interval=0.01 if args.power_trace else 0.2)
    barrier(); torch.cuda.synchronize()
    sampler.start()
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
    if dist_on:                                                # MAX over ranks at once: every decision below that depends on `elapsed` is then the same on every rank
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
    power = sampler.stop()
    if args.power_trace and sampler.rows:
        with open(args.power_trace, "w") as f:
            for w_, mhz_, t_ in sampler.rows: f.write(json.dumps({"t_s": round(t_ - t0, 4), "socket_power_w": w_, "sclk_mhz": mhz_}) + "\n")
    prof = ctx.prof_read()
    ctx.prof_enable(0)
    hbm_free, hbm_total = torch.cuda.mem_get_info()            # device-wide: the library's allocations, torch's, every rank's when the GPU is shared
    assert verdicts_ok(args.steps), "timed-region verdicts must be ACCEPT"
    if gathered is not None:
        assert all(int(g.sum()) == B for g in gathered), "every rank's shard must be ACCEPT"

    if args.preflight:
        import socket
        try:
            bus = getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None)
            dom = getattr(torch.cuda.get_device_properties(local_rank), "pci_domain_id", 0)
            dv_ = getattr(torch.cuda.get_device_properties(local_rank), "pci_device_id", 0)
            bus_id = None if bus is None else f"{dom:04x}:{bus:02x}:{dv_:02x}"
        except Exception:                                        # noqa: BLE001 -- a torch without the PCI fields: the uuid names the device as well
            bus_id = None
        if bus_id is None:
            bus_id = str(getattr(torch.cuda.get_device_properties(local_rank), "uuid", f"device{local_rank}"))
        mine = {"rank": rank, "local_rank_env": int(os.environ.get("LOCAL_RANK", "0")), "device_bound": local_rank, "pci_bus_id": bus_id, "host": socket.gethostname(),
                "devices_visible": torch.cuda.device_count(), "hbm_total_GiB": round(hbm_total / 2**30, 1), "verdicts_accept": bool(verdicts_ok(1))}
        facts, backend_ranks, summed = [mine], 1, 1.0
        if dist_on:
            facts = [None] * world
            dist.all_gather_object(facts, mine)
            backend_ranks = dist.get_world_size()
            one = torch.ones(1, dtype=torch.float32, device="cpu" if share_gpu else dev)
            dist.all_reduce(one)                                 # on the data-path backend (RCCL unless the ranks share a GPU): every rank must have contributed
            summed = float(one.item())
        ctx.close()
        rc = 0
        if rank == 0:
            problems = []
            seen = {}
            pretend = os.environ.get("MINA_BENCH_PREFLIGHT_PRETEND_DISTINCT") == "1"      # test hook: shared ranks that claim to own their GPUs -> the same-device rule must fire
            if pretend: share_gpu = False
            for f_ in facts:
                key = (f_["host"], f_["pci_bus_id"])
                if key in seen and not share_gpu: problems.append(f"ranks {seen[key]} and {f_['rank']} are bound to the same device {f_['pci_bus_id']} without shared_gpu")
                seen.setdefault(key, f_["rank"])
                if not f_["verdicts_accept"]: problems.append(f"rank {f_['rank']}: the step's verdicts are not all ACCEPT")
                if not share_gpu and not pretend and f_["device_bound"] != f_["local_rank_env"]: problems.append(f"rank {f_['rank']} bound device {f_['device_bound']}, LOCAL_RANK says {f_['local_rank_env']}")
            if backend_ranks != args.gpus or int(round(summed)) != args.gpus: problems.append(f"the backend saw {backend_ranks} ranks (all-reduce of ones = {summed}), --gpus says {args.gpus}")
            if gathered is not None and len(gathered) != world: problems.append("the verdict all-gather did not return one shard per rank")
            print(json.dumps({"preflight": True, "ok": not problems, "problems": problems, "n_gpus": args.gpus, "world_size": world, "shared_gpu": bool(share_gpu and dist_on),
                              "backend": ("gloo (ranks share one GPU)" if share_gpu else "cpu:gloo,cuda:nccl (RCCL)") if dist_on else "none", "backend_ranks": backend_ranks,
                              "all_reduce_of_ones": summed, "ranks": facts, "launcher": os.environ.get("MINA_BENCH_LAUNCHER", "torch.distributed.run" if dist_on else "none"),
                              "step": {"proofs_per_rank": B, "verdict_all_gather_shards": len(gathered) if gathered is not None else 0}}), flush=True)
            rc = 1 if problems else 0
        if dist_on:
            t = torch.tensor([rc], dtype=torch.int32)
            dist.broadcast(t, src=0)                             # every rank leaves with rank 0's verdict (CPU tensor -> gloo)
            rc = int(t.item())
            dist.destroy_process_group()
        raise SystemExit(rc)

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

    def set_tune(extra=""):                                    # the library's defaults + $MINA_TUNE (experiments) + `extra`, between calls only
        spec = ",".join(x for x in (os.environ.get("MINA_TUNE", ""), extra) if x)
        m.lib.verify_configure_ex(None)
        if spec: m.lib.tune_from_string(spec)

    def sub_job(Bx):                                           # the first Bx proofs of the resident job
        djx = m.lib.StateJobs(); ctypes.memmove(ctypes.byref(djx), ctypes.byref(dj), ctypes.sizeof(m.lib.StateJobs)); djx.batch = Bx
        keepx = [djx]
        if kp is not None:
            dkx = m.lib.KimchiProofs(); ctypes.memmove(ctypes.byref(dkx), ctypes.byref(dk), ctypes.sizeof(m.lib.KimchiProofs)); dkx.batch = Bx
            djx.kimchi = ctypes.addressof(dkx); keepx.append(dkx)
        return djx, keepx

    # latency of ONE call alone on the chip, by call size (the legs of the job forked, wave priorities on: what a caller without a pipeline sees); measured here, while the
    # process holds only the pipeline's streams -- a process with more streams than hardware queues stays slower for life
    call_latency_ms, call_latency_by_size = None, None
    if not args.no_probes and not (dist_on and share_gpu):
        ctx.set_pipeline(1)
        call_latency_by_size = {}
        for Bx in sorted({B, 8192, 4096, 1024}, reverse=True):
            if Bx > B: continue
            djx, keepx = sub_job(Bx)
            o = d_out[0]
            for _ in range(2):
                ctx.state_job_batch_dev(djx, o.data_ptr(), o.data_ptr() + 4 * Bx)
            ctx.synchronize(); t1 = time.perf_counter()
            for _ in range(4):
                ctx.state_job_batch_dev(djx, o.data_ptr(), o.data_ptr() + 4 * Bx); ctx.synchronize()
            call_latency_by_size[str(Bx)] = (time.perf_counter() - t1) / 4 * 1e3
            got = o.cpu().numpy()
            assert got[:Bx].tolist() == [1] * Bx and got[Bx:Bx + 4].tolist() == [1, 0, 1, 0]
        call_latency_ms = call_latency_by_size[str(B)]
        ctx.set_pipeline(args.pipeline)
        for o in d_out:
            o.zero_()
        torch.cuda.synchronize()

    prof_iso = {}
    # Ranks that share ONE GPU have no "nothing else on the GPU": in about every second shared run an isolated launch took 500 - 1000 ms for 35 (the other process's queues are
    # time-sliced in by the hardware scheduler whether or not they hold work) and C2 read 740 - 2500 checks/s for 13 000.  Those keys are left out of a shared-GPU line.
    iso_legs = not args.no_probes and not (dist_on and share_gpu)
    # These legs run HERE, before anything creates more streams: a process holding more streams than the runtime has hardware queues (24 here: GPU_MAX_HW_QUEUES) keeps
    # every later launch slower -- the same isolated state-hash launch read 33.4 ms after 4 forked lanes (20 streams) and 39.5 ms after 6 (30), for the rest of the process
    # (tools/probes/iso_after_fork.py).  The C5 leg (5 lanes) comes after; the C2 and one-proof-per-call legs run in processes of their own.
    # the candidate dominant kernels with nothing else on the GPU: one lane, HIP events on that lane's stream
    if iso_legs:
        ctx.set_pipeline(1)
        set_tune("dev_fork=0")                                    # ONE stream: nothing runs beside the kernel being timed (forked, a lone job's hashes share the chip with its chain)
        for _ in range(2):
            step()
        ctx.synchronize()
        ctx.prof_enable(mask)
        for _ in range(6):
            step()
        prof_iso = ctx.prof_read()
        ctx.prof_enable(0)
        set_tune()
        ctx.set_pipeline(args.pipeline)
        for o in d_out:
            o.zero_()
        torch.cuda.synchronize()

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
        lanes5 = min(max(args.pipeline, 5), 8)                     # a 4096-proof job is 3316 hash waves and 196-wave chain kernels: more of them in flight (4 lanes 235 k, 5 - 6: 254 k);
                                                                   # 5: the process stays within its 24 hardware queues (5 lanes + 15 helper streams + a side stream)
        ctx.set_pipeline(lanes5)
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
              "ms_per_step": el5 / n5 * 1e3, "pipeline_lanes": lanes5}
        ctx.set_pipeline(args.pipeline)
        for o in d_out:
            o.zero_()
        torch.cuda.synchronize()

    # SURVEY.md 8e.2 for the whole job (the multi-GPU variant north_star names): ONE exchange step -- all-to-all of the shards' folded scalar vectors, the two
    # fixed-base MSMs base-sliced over the ranks, all-gather of the partial points (mina_bridge_amd/sharded.py ShardedStateJob).  Secondary key at N > 1: it pays
    # when the batch per GPU is small; at 8192 proofs per rank the proof-level sharding above (no data-path collective) is the faster form.
    exchange = None
    if dist_on and not args.no_probes and args.mode == "full":
        from mina_bridge_amd.sharded import DeviceBackend, ShardedStateJob
        sj = ShardedStateJob(DeviceBackend(ctx, dev), k=WRAP_K, acc_k=ACC_K)
        Bx = min(B, 1024)
        djx = m.lib.StateJobs(); ctypes.memmove(ctypes.byref(djx), ctypes.byref(dj), ctypes.sizeof(m.lib.StateJobs)); djx.batch = Bx
        keepx = []
        if kp is not None:
            dkx = m.lib.KimchiProofs(); ctypes.memmove(ctypes.byref(dkx), ctypes.byref(dk), ctypes.sizeof(m.lib.KimchiProofs)); dkx.batch = Bx
            djx.kimchi = ctypes.addressof(dkx); keepx.append(dkx)
        try:                                                       # a secondary leg on a path no multi-GPU node has ever run (RCCL collectives on the context's stream): it must not take the line down
            vx, okx = sj.verify(djx, Bx)                               # warm
            barrier(); torch.cuda.synchronize(); tx = time.perf_counter()
            nx = 8
            for _ in range(nx):
                vx, okx = sj.verify(djx, Bx)
            torch.cuda.synchronize(); barrier()
            elx = time.perf_counter() - tx
            assert okx and bool(vx.all()), "the exchanged check must ACCEPT"
            exchange = {"value": args.gpus * nx * Bx / elx, "unit": "proofs/s", "proofs_per_rank_per_call": Bx, "calls": nx, "ms_per_call": elx / nx * 1e3,
                        "collectives": "all_to_all_single (2 x folded scalars: 1 MiB + 2 MiB per rank) + all_gather (276 B per rank) over " + ("gloo, host tensors (shared GPU)" if share_gpu else "RCCL, HBM to HBM"),
                        "note": "synchronous calls (no pipelining): the latency of ONE batch spread over the ranks with a single exchange step"}

        except Exception as e:                                     # noqa: BLE001
            exchange = {"error": repr(e)[:400]}
            if dist_on:
                try: barrier()
                except Exception: pass                             # noqa: BLE001

    if dist_on:
        t = torch.tensor([elapsed, sustained["seconds"] if sustained else 0.0, c5["ms_per_step"] if c5 else 0.0], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if sustained:
            sustained["seconds"] = float(t[1].item()); sustained["value"] = args.gpus * sustained["steps"] * B / sustained["seconds"]
        if c5:
            c5["ms_per_step"] = float(t[2].item()); c5["value"] = 4096 / (c5["ms_per_step"] * 1e-3)
    ctx.close()

    c2_rate = (c2_own or {}).get("value"); c2_power = (c2_own or {}).get("power")
    if rank == 0:
        def avg_us(p, name):
            n, ms = p.get(name, [0, 0.0])
            return (ms / n) * 1e3 if n else None
        iso = {k: avg_us(prof_iso, k) for k in PROF_STAGES}
        ovl = {k: avg_us(prof, k) for k in PROF_STAGES}
        src = iso if iso.get("pstate_hash") else ovl
        # dominant kernel of the job: the protocol-state hash (17 sponges of ~27 permutations per proof)
        kern_us = src.get("pstate_hash") if not (dist_on and share_gpu) else None      # no launch of a shared-GPU run is an isolated one
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
            **({"shared_gpu": True, "gpus_physical": int(os.environ.get("MINA_BENCH_GPUS_PHYSICAL", "1")),
                "shared_gpu_note": f"{args.gpus} ranks time-share ONE GPU (a box with fewer GPUs than --gpus, or the test hook): the N > 1 code path end to end, NOT a scaling figure"}
               if share_gpu and dist_on else {}),
            "launcher": os.environ.get("MINA_BENCH_LAUNCHER", "torch.distributed.run" if dist_on else "none"),
            "ms_per_step": elapsed / args.steps * 1e3,
            "call_latency_ms": call_latency_ms,
            "call_latency_ms_by_size": call_latency_by_size,
            "joules_per_proof": (power["socket_power_w_avg"] * elapsed / (args.steps * B)) if power and not (dist_on and share_gpu) else None,
            "power": power,
            "toolchain": toolchain(),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x8-montgomery (255-bit prime fields, integer)", "data": "synthetic",
            "config": {"workload": "C3: full Proof-of-State job per proof -- 17 protocol-state hashes (chain of 16 + bridge tip) vs public inputs + linkage, "
                                   + ("the Pickles statement's deferred values and digests, " if args.mode == "full" else "")
                                   + ("kimchi oracles + to_batch of the wrap proof, " if args.kimchi else "") +
                                   "wrap-proof public-input commitment (40 inputs, 2^15 Pallas domain), wrap IPA opening (k=15, 45+ commitments x 2 points), "
                                   "2^16-base Vesta step-accumulator check; verdict per proof, bit-exact vs the CPU oracle composite (tests/test_state_job.py)",
                       "proofs_per_step": B, "pipeline_lanes": args.pipeline, "hbm_in_use_GiB": round((hbm_total - hbm_free) / 2**30, 1), "hbm_total_GiB": round(hbm_total / 2**30, 1), "warmup_steps_run": n_warm,
                       "mode": args.mode,
                       "distinct_inputs": {"full": {**distinct, "note": "chains: flattened protocol-state records, random field content, GPU-linked, one per proof; wrap_proofs: complete "
                                                             "(statement + wrap proof + opening + its step accumulator), minted by the repo's CPU prover under the synthetic indexes "
                                                             "(tests/golden/statement_k15_encoded.json + statement_k15_many.npz), tiled to the batch; the statement's application state "
                                                             "is the fixture's, not the hash of the chain beside it (that binding: tests/test_verify_fullsize.py)"},
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
                       "sharding": f"proof-level, {args.gpus} rank(s); verdict words all-gathered over {'gloo (ranks share one GPU)' if share_gpu else 'RCCL'}" if dist_on else "single rank",
                       "algorithmic_bytes_per_proof": algorithmic_bytes_per_proof()},
            "roofline": None,                                  # filled below: the binding resource is integer VALU issue, the HBM view rides inside it
            "stage_us": {"isolated": iso, "in_timed_region": ovl},
            "sustained": sustained,
            "c5_4096_total_strong": c5,
            "exchange_variant_8e2": exchange,
            "one_proof_per_call": one_pc,
            "boundary_bytes_to_bools": boundary,
            "c4_account_256": c4,
            "c2_accumulator_only": {"value": c2_rate, "unit": "accumulator checks/s", "single_check": (c2_own or {}).get("single_check"),
                                    **({"error": c2_own["error"]} if c2_own and "error" in c2_own else {}), "process": "a fresh process holding only libminaverify.so",
                                    "note": "BASELINE config C2 alone (round 1's headline): un-folded 2^16-base Vesta IPA accumulator checks, 8 per call, 16 lanes",
                                    # the metric's second half ("MSM HBM GB/s vs peak"): one check = one 2^16-base MSM; algorithmic bytes = bases + scalars
                                    "power": c2_power,
                                    "msm_valu": None if not c2_rate else msm_valu(c2_rate),
                                    "msm_hbm": None if not c2_rate else {"algorithmic_bytes_per_msm": 65536 * (64 + 32) + 96, "achieved_GBps": c2_rate * (65536 * 96 + 96) / 1e9, "peak_GBps": HBM_PEAK_GBPS,
                                                                         "frac": c2_rate * (65536 * 96 + 96) / 1e9 / HBM_PEAK_GBPS, **msm_traffic(c2_rate),
                                                                         "note": "the bucket MSM is bound by the group law's multiply-accumulates (the accumulate kernel: 78 % of a check's instructions, on 29-bit limbs "
                                                                                 "since round 4, six of its nine products lazy since round 5: profiles/r05_k1.md), not by HBM; the fixed-base window tables "
                                                                                 "trade bandwidth for doubling chains: one 64-B point gathered per (base, window)"}},
        }
        hbm_view = {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                    "algorithmic_bytes_per_launch": hash_bytes, "traffic": (nstates * traffic_per_state) if traffic_per_state else None,
                    "traffic_ratio_to_algorithmic": (nstates * traffic_per_state / hash_bytes) if traffic_per_state else None,
                    "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read from {TRAFFIC_FILE} ({traffic_src})" if traffic_per_state else None,
                    "note": "reported because the metric asks for it: 1632 algorithmic bytes per state against ~3.3 M limb multiply-accumulates -- the kernel cannot be HBM-bound"}
        if kern_us:
            peak = CHIP_SIMDS * 64 * CLOCK_HZ / MAD_ISSUE_CYCLES
            pure_peak = CHIP_SIMDS * 64 * CLOCK_HZ / PURE_MAC_ISSUE_CYCLES
            got = perms * 3 * 55 * MADS_PER_LANE_ROUND / (kern_us * 1e-6)
            waves = -(-nstates // 21)                          # 21 sponges per wave64 (3 lanes each)
            out["roofline"] = {"bound": "valu_int32", "kernel": "pstate_hash_kernel", "achieved": got / 1e12, "peak": pure_peak / 1e12, "unit": "T limb-MAC/s", "frac": got / pure_peak,
                               "frac_of_own_mix_ceiling": got / peak, **clock_normalised(got / pure_peak, power), "traffic": hbm_view["traffic"], "hbm": hbm_view,
                               "states_per_launch": nstates, "avg_launch_us": kern_us, "avg_launch_us_in_timed_region": ovl.get("pstate_hash"),
                               "limb_macs_per_launch": perms * 3 * 55 * MADS_PER_LANE_ROUND,
                               "peak_source": f"measured pure v_mad_u64_u32 stream, {PURE_MAC_ISSUE_CYCLES} cycles at 2.4 GHz per wave64 instruction per SIMD, 8 waves per SIMD "
                                              "(profiles/r05_microbench_ratio.jsonl, S = 0; re-measured per round: profiles/README.md)",
                               "note": "bound = the binding resource (SURVEY.md 8d: integer VALU, not HBM, not MFMA).  achieved = 29-bit limb multiply-accumulates per launch / the "
                                       "launch's HIP-event duration on its lane stream (isolated launches right after the timed region; second figure: inside it).  frac counts every "
                                       "non-multiply instruction of the round (151 of 925 per lane-round) as a lost slot; frac_of_own_mix_ceiling prices the kernel's own mix instead. "
                                       "`hbm`: the metric's 'HBM GB/s vs peak' view of the same launch; `traffic` = its PMC bytes"}
            out["roofline_valu"] = {"bound": "issue rate of the kernel's own instruction mix (774 multiply-accumulates + ~150 shifts / masks per lane-round), measured: "
                                             f"{MAD_ISSUE_CYCLES} cycles at 2.4 GHz per wave64 multiply-accumulate per SIMD", "kernel": "pstate_hash_kernel",
                                    "achieved": got / 1e12, "peak": peak / 1e12, "unit": "T limb-MAC/s", "frac": got / peak, **clock_normalised(got / peak, power), "frac_of_pure_mac_peak": got / pure_peak, "pure_mac_peak": pure_peak / 1e12,
                                    "permutations_per_launch": perms,
                                    "limb_macs_per_permutation": 3 * 55 * MADS_PER_LANE_ROUND,
                                    "waves_per_launch": waves, "waves_per_simd_in_launch": waves / CHIP_SIMDS, "resident_waves_per_simd": 5,
                                    "peak_source": "profiles/r05_microbench_ratio.jsonl (the round's mix, 8 waves per SIMD); counters of the kernel: profiles/r04_valu_roofline.md",
                                    "note": "9 x 29-bit limbs, no carry instructions.  The kernel holds 96 VGPRs: 5 waves per SIMD are resident (the mix issues at 5.7 cycles per "
                                            "multiply-accumulate with 4 waves, 5.38 with 8), and a launch whose waves per SIMD are not a multiple of 5 ends on partly filled SIMDs "
                                            "(8192 proofs: 6.5 waves per SIMD, isolated launch 0.82 of the ceiling; 16384: 12.95)"}
        else:
            out["roofline"] = {"bound": "valu_int32", "kernel": "pstate_hash_kernel", "achieved": None, "peak": CHIP_SIMDS * 64 * CLOCK_HZ / PURE_MAC_ISSUE_CYCLES / 1e12, "unit": "T limb-MAC/s",
                               "frac": None, "traffic": hbm_view["traffic"], "hbm": hbm_view, "note": "no isolated kernel timing in this run (--no-probes without the stage events, or ranks that share one GPU)"}
        if args.mode == "full" and not share_gpu:            # (ranks sharing one GPU: a step's time is not one GPU's)
            out["step_valu"] = step_valu(out["ms_per_step"], B)  # the pipelined step as a whole against instruction issue (secondary; `roofline` stays the dominant kernel's)
        if out.get("step_valu"):
            out["step_valu"].update(clock_normalised(out["step_valu"]["frac"], power))
        if not args.no_cpu_baseline and args.gpus == 1:       # the CPU leg is timed at N = 1 only (rank 0)
            out["cpu_baseline"] = cpu_baseline(baseline_sample)
            if isinstance(out["cpu_baseline"], dict) and "folded" in out["cpu_baseline"]:
                out["cpu_baseline_folded"] = out["cpu_baseline"].pop("folded")     # beside it: the same CPU code with the GPU job's batch fold
        line = json.dumps(out)
    if dist_on:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(line, flush=True)                                # the ONE JSON line, after everything a backend may write to stdout (RCCL prints its library path)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] in ("--boundary-only", "--boundary-all-devices", "--account-only", "--one-proof-only", "--c2-only"):     # a bytes -> bools leg in a process of its own (see main): only the library, no torch
        import mina_bridge_amd as _m
        if sys.argv[1] == "--account-only":
            print(json.dumps(account_leg(_m, sys.argv[2])), flush=True)
        elif sys.argv[1] == "--one-proof-only":
            print(json.dumps(one_proof_leg(_m, sys.argv[2])), flush=True)
        elif sys.argv[1] == "--c2-only":
            print(json.dumps(c2_leg(_m, sys.argv[2], "" if sys.argv[3] == "-" else sys.argv[3])), flush=True)
        else:
            fn = boundary_leg if sys.argv[1] == "--boundary-only" else boundary_all_devices_leg
            print(json.dumps(fn(_m, sys.argv[2], int(sys.argv[3]))), flush=True)
    else:
        main()
