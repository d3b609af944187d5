"""The performance contract of the hot kernels, read from the gfx950 code objects inside libminaverify.so (CPU tier: no GPU needed).

Every roofline fraction in the bench line rests on properties of ONE compiler build: registers per lane (-> waves per SIMD), nothing
spilled inside the round loops, the multiply-accumulate count of a lane-round / a mixed add, the matrix-core instruction in the b_poly
fold.  DESIGN.md section 4 used to say "checked in the ISA"; this file is that check.  A compiler bump, a dropped
`amdgpu_waves_per_eu`, or an edit that makes the allocator spill inside a loop fails HERE, not as a silent halving of occupancy.

The arithmetic these kernels implement is ark-ff's `Fp256` Montgomery field of the Pasta primes (/root/reference/core/Cargo.toml:19-21);
correctness is the GPU tier's business -- this tier pins cost.

Budgets are upper bounds with a small allowance (a compiler may schedule a few moves differently); the exact figures of the build in
the tree are printed by `python tools/code_object.py` and recorded in the bench line (`code_object`).
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import code_object as CO  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(CO.LLVM_BIN, "llvm-objdump")), reason="LLVM binutils of the ROCm toolchain not present")

WAVE_VGPR_FILE = 512          # gfx950: 512 VGPRs per lane per SIMD (unified with AGPRs), allocated in blocks of 8


def waves_per_simd(meta) -> int:
    regs = meta["vgpr_count"]                      # on gfx90a+ this is the unified arch + acc count
    blocks = -(-regs // 8) * 8
    return min(8, WAVE_VGPR_FILE // blocks)


@pytest.fixture(scope="module")
def co():
    c = CO.CodeObjects()
    yield c
    c.close()


# kernel -> (max VGPRs, max VGPR spills, max scratch bytes, LDS bytes or None for "do not care", min waves/SIMD)
CONTRACT = {
    # the dominant kernel of the step (C3/C5): 96 VGPRs = 5 waves per SIMD, held by amdgpu_waves_per_eu(5,5) (api_state.hip:27)
    "pstate_hash_kernel<0, 3>": (96, 1, 8, 0, 5),
    "pstate_hash_kernel<0, 8>": (96, 0, 0, 0, 5),
    "pstate_hash_kernel<0, 16>": (96, 0, 0, 0, 5),
    # the wrap-proof transcript kernels in the 3-lane form: 4 waves per SIMD (amdgpu_waves_per_eu(4,8), api_kimchi.hip:81, api_ipa.hip:33)
    "kimchi_fq_kernel<3>": (128, 8, 24, 0, 4),
    "kimchi_fr_kernel<3>": (128, 0, 0, 0, 4),
    "ipa_prepare_kernel<1, 3, 1>": (96, 0, 0, 0, 5),
    "ipa_prepare_kernel<0, 3, 1>": (96, 0, 0, 0, 5),
    "pickles_digest_kernel<3>": (128, 0, 0, 0, 4),
    "pickles_tick_kernel<3>": (128, 0, 0, 0, 4),
    "salted_hash_kernel<0, 3>": (112, 0, 0, 0, 4),
    "merkle_fold_coop_kernel<0, 3>": (112, 0, 0, 0, 4),
    # K1: the 29-bit accumulate kernels (msm.cuh): 128 / 136 VGPRs = 4 / 3 waves per SIMD, nothing in scratch
    "msm_accumulate29_kernel<0, 1>": (128, 0, 0, 0, 4),
    "msm_accumulate29_kernel<1, 1>": (128, 0, 0, 0, 4),
    "msm_accumulate_bucket29_kernel<0, 1>": (136, 0, 0, 0, 3),
    "msm_accumulate_bucket29_kernel<1, 1>": (136, 0, 0, 0, 3),
    "msm_segsum29_kernel<0>": (128, 0, 0, 0, 4),
    "msm_segsum29_kernel<1>": (128, 0, 0, 0, 4),
    "msm_part_sort_kernel": (64, 0, 0, 16512, 8),          # LDS-only counting sort: 16 KiB + 128 B per workgroup
    # the one dense contraction: int8 digit planes on the matrix cores, 80 KiB of LDS tiles
    "bpoly_field_gemm_kernel": (264, 0, 0, 81920, 1),
}


@pytest.mark.parametrize("kernel", sorted(CONTRACT))
def test_registers_spills_scratch_lds(co, kernel):
    ks = co.kernels()
    assert kernel in ks, f"{kernel} not in libminaverify.so (have e.g. {sorted(k for k in ks if kernel.split('<')[0] in k)[:6]})"
    m = ks[kernel]
    vmax, spill_max, scratch_max, lds, waves_min = CONTRACT[kernel]
    total = m["vgpr_count"] + (m.get("agpr_count", 0) if kernel == "bpoly_field_gemm_kernel" else 0)
    assert total <= vmax, f"{kernel}: {total} VGPRs > {vmax}"
    assert m["vgpr_spill_count"] <= spill_max, f"{kernel}: {m['vgpr_spill_count']} VGPR spills > {spill_max}"
    assert m["private_segment_fixed_size"] <= scratch_max, f"{kernel}: {m['private_segment_fixed_size']} B of scratch > {scratch_max}"
    assert not m.get("uses_dynamic_stack", False), f"{kernel} uses a dynamic stack"
    assert m.get("wavefront_size", 64) == 64
    if lds is not None:
        assert m["group_segment_fixed_size"] == lds, f"{kernel}: LDS {m['group_segment_fixed_size']} != {lds}"
    assert waves_per_simd(m) >= waves_min, f"{kernel}: {m['vgpr_count']} VGPRs -> {waves_per_simd(m)} waves per SIMD < {waves_min}"


def _round_loops(co, kernel, min_mac):
    """the innermost loops of a kernel that hold at least `min_mac` 64-bit multiply-accumulates: the Poseidon round loops"""
    loops = co.loops(kernel)
    out = []
    for (s, e) in loops:
        if any(s <= s2 and e2 <= e and (s2, e2) != (s, e) for (s2, e2) in loops):
            continue
        summ = CO.summarize(co.histogram(kernel, (s, e)))
        if summ["mac64"] >= min_mac:
            out.append(((s, e), summ))
    return out


def test_dominant_kernel_round_loops(co):
    """`poseidon_rounds_tri` inside pstate_hash_kernel<0,3> (sponge.cuh): per lane-round 774 multiply-accumulates (+ 1 address
    computation) in <= 925 VALU instructions, <= 52 s_nop, 18 cross-lane moves, and NO scratch access: the values the allocator parks
    in scratch live outside the round loops."""
    k = "pstate_hash_kernel<0, 3>"
    loops = _round_loops(co, k, 700)
    assert len(loops) == 3, f"expected the three round loops (body sponge first block / later blocks, state sponge), found {[(s, x['mac64']) for s, x in loops]}"
    for span, s in loops:
        assert 774 <= s["mac64"] <= 776, (span, s)
        assert s["valu"] <= 925, (span, s)
        assert s["s_nop"] <= 52, (span, s)
        assert s["scratch"] == 0, (span, s)
        assert s["ds_bpermute"] == 18, (span, s)
        assert s["mfma"] == 0
    whole = CO.summarize(co.histogram(k))
    assert whole["scratch"] <= 3, whole                # the parked values: touched outside the loops only


def test_eight_lane_round_loops(co):
    k = "pstate_hash_kernel<0, 8>"
    loops = _round_loops(co, k, 500)
    assert len(loops) == 3
    for span, s in loops:
        assert s["mac64"] <= 596 and s["valu"] <= 772 and s["scratch"] == 0, (span, s)
    assert CO.summarize(co.histogram(k))["scratch"] == 0


@pytest.mark.parametrize("kernel,valu_max", [("msm_accumulate29_kernel<0, 1>", 1673), ("msm_accumulate29_kernel<1, 1>", 1673),
                                             ("msm_accumulate_bucket29_kernel<0, 1>", 1673), ("msm_accumulate_bucket29_kernel<1, 1>", 1673)])
def test_mixed_add_budget(co, kernel, valu_max):
    """the XYZZ mixed add on 29-bit limbs (ec29.cuh `xyzz29_madd`) inside the K1 accumulate loops: <= 1673 VALU instructions per
    added point of which 1248 are multiply-accumulates (8 products + 2 squarings... of the signed-digit forms), nothing in scratch."""
    loops = co.loops(kernel)
    cand = []
    for (s, e) in loops:
        summ = CO.summarize(co.histogram(kernel, (s, e)))
        if summ["mac64"] >= 1200 and summ["global_load"] >= 4:          # the loop that loads a table point and adds it
            cand.append(((s, e), summ))
    assert cand, f"{kernel}: no loop with a table load and a full mixed add"
    span, s = min(cand, key=lambda c: c[0][1] - c[0][0])
    assert s["mac64"] == 1248, (span, s)
    assert s["valu"] <= valu_max, (span, s)
    assert s["scratch"] == 0
    assert CO.summarize(co.histogram(kernel))["scratch"] == 0


def test_bpoly_fold_runs_on_the_matrix_cores(co):
    ins = co.instructions("bpoly_field_gemm_kernel")
    mfma = [mn for _, mn, _ in ins if mn.startswith("v_mfma")]
    assert mfma and all(mn == "v_mfma_i32_32x32x32_i8" for mn in mfma), sorted(set(mfma))
    hot = max((CO.summarize(co.histogram("bpoly_field_gemm_kernel", sp)) for sp in co.loops("bpoly_field_gemm_kernel")), key=lambda s: s["mfma"])
    assert hot["mfma"] >= 16 and hot["scratch"] == 0


def test_no_hot_kernel_uses_scratch_unlisted(co):
    """every kernel with scratch is on this list with its reason; a new entry means the allocator started spilling somewhere"""
    allowed = {
        "pstate_hash_kernel<0, 3>": 8,                  # parked outside the round loops (test above)
        "kimchi_fq_kernel<3>": 24,                      # 8 spills at 128 VGPRs: outside the permutation (checked below)
        "kimchi_pub_kernel<10>": 656, "msm_scan_kernel": 32,
        # the group map's three candidate x indexed by the first square: one-off / per-proof work (groupmap.cuh)
        "ipa_to_group_kernel<0>": 112, "ipa_to_group_kernel<1>": 112, "to_group_kernel<0>": 112, "to_group_kernel<1>": 112,
        "srs_create_kernel<0>": 112, "srs_create_kernel<1>": 112,
        # the test-facing sponge interpreter (api_sponge.hip): not on the job's path
        "sponge_tape_kernel<0, 3>": 24, "sponge_tape_kernel<1, 3>": 16,
    }
    for name, m in co.kernels().items():
        base = name.split("#")[0]
        sz = m["private_segment_fixed_size"]
        if sz == 0:
            continue
        if re.match(r"ipa_prepare_kernel<\d, \d+, 2>", base):       # the latency form with its 16 challenges indexed dynamically
            assert sz <= 1296, (name, sz)
            continue
        if base.startswith(("group_law_selftest", "polish_", "kimchi_scalar", "pickles_scalar")):
            continue
        assert base in allowed and sz <= allowed[base], f"{name}: {sz} B of scratch is not in the contract"


def test_kimchi_fq_spills_stay_outside_the_permutation(co):
    k = "kimchi_fq_kernel<3>"
    for span, s in _round_loops(co, k, 700):
        assert s["scratch"] == 0, (span, s)


def test_wave_priorities(co):
    """round 6: the legs of a device-resident job run on streams of their own, and every kernel of a job except the chip-filling hashes opens with `s_setprio 2`
    (fp.cuh mb_wave_prio): where a wave of the wrap-proof chain shares a SIMD with resident state-hash waves the arbiter takes it first.  Half of the round's headline
    gain (4 forked lanes: 286.6 -> 307.2 k proofs/s, same box) rests on that one instruction being there -- and on the hash kernels NOT raising theirs."""
    def prios(kernel):
        return [op for _, mn, op in co.instructions(kernel) if mn == "s_setprio"]
    for k in ("pstate_hash_kernel<0, 3>", "pstate_hash_kernel<0, 8>", "pstate_hash_kernel<0, 16>"):
        assert prios(k) == [], (k, prios(k))
    # (the Proof-of-Account kernels have raised theirs to 3 since round 4: short dependent chains beside state jobs -- api_account.hip, sponge.cuh merkle_fold_coop_kernel)
    for k in ("pickles_expand_kernel", "pickles_digest_kernel<3>", "pickles_tick_kernel<3>", "pickles_scalar_kernel", "kimchi_fq_kernel<3>", "kimchi_fr_kernel<3>", "kimchi_pub_kernel<10>",
              "kimchi_scalar_kernel", "ipa_prepare_kernel<0, 3, 2>", "ipa_to_group_kernel<0>", "pubcomm_direct29_kernel<0, 8>", "pubcomm_finish16_kernel<0>", "challenge_to_field_kernel<0>",
              "bpoly_tables_digits8_kernel<0>", "bpoly_field_gemm_kernel", "msm_part_sort_kernel", "msm_accumulate29_kernel<1, 1>", "msm_accumulate_bucket29_kernel<1, 1>",
              "msm_segsum_kernel<1, true>", "msm_reduce2d_kernel<1>"):
        p_ = prios(k)
        assert p_ and p_[0].strip() in ("2", "0x2") and len(set(x.strip() for x in p_)) == 1, (k, p_)
        first = [mn for _, mn, _ in co.instructions(k)][:40]
        assert "s_setprio" in first, (k, "the priority is raised at the top of the kernel, before any long-latency work", first[:12])


def test_compiler_recorded():
    v = CO.compiler_version()
    assert v.startswith("HIP "), v


def _compile_state_without_attribute(tmp_path):
    src = open(os.path.join(ROOT, "mina_bridge_amd", "csrc", "api_state.hip")).read()
    pat = "__attribute__((amdgpu_waves_per_eu(LANES == 3 ? 5 : 1, LANES == 3 ? 5 : 8)))"
    assert pat in src, "api_state.hip no longer carries the attribute this test removes"
    csrc = tmp_path / "mina_bridge_amd" / "csrc"                     # ctx.h reaches the header as ../../include/mina_verify.h
    shutil.copytree(os.path.join(ROOT, "mina_bridge_amd", "csrc"), csrc)
    os.makedirs(tmp_path / "include", exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "mina_verify.h"), tmp_path / "include" / "mina_verify.h")
    (csrc / "api_state.hip").write_text(src.replace(pat, ""))
    from mina_bridge_amd import build as B
    obj = tmp_path / "api_state.o"
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-c", str(csrc / "api_state.hip"), "-o", str(obj)])
    return str(obj)


@pytest.mark.skipif(os.environ.get("MINA_CODE_OBJECT_FALSIFY", "1") != "1" or not shutil.which("hipcc"), reason="recompiles api_state.hip (~10 s); MINA_CODE_OBJECT_FALSIFY=0 skips it")
def test_contract_fails_without_waves_per_eu(tmp_path):
    """the falsification: with `amdgpu_waves_per_eu` removed from api_state.hip:27 the dominant kernel takes more registers and this
    file's first test fails"""
    obj = _compile_state_without_attribute(tmp_path)
    c = CO.CodeObjects(obj)
    try:
        m = c.kernels()["pstate_hash_kernel<0, 3>"]
        assert m["vgpr_count"] > CONTRACT["pstate_hash_kernel<0, 3>"][0] or waves_per_simd(m) < 5, m
    finally:
        c.close()
