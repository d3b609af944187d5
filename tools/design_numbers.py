#!/usr/bin/env python3
"""DESIGN.md section 0 ("numbers at a glance") is GENERATED from the tracked bench line (VERDICT r04 next #2d: one source per figure).

    python tools/design_numbers.py [profiles/r06_bench.json] [--write]

Every figure of the table is read from that one JSON line (and from profiles/r06_bench_steps20.json beside it for the driver's --steps 20 shape); nothing is typed by
hand.  tests/test_abi.py checks that the block between the NUMBERS markers of DESIGN.md is what this script prints for the tracked files."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- NUMBERS:BEGIN (tools/design_numbers.py: generated from profiles/r06_bench.json, do not edit) -->", "<!-- NUMBERS:END -->"


def line_of(path):
    for ln in open(path):
        if ln.startswith('{"metric"'):
            return json.loads(ln)
    raise SystemExit(f"{path}: no bench line")


def k(x, d=1):
    return f"{x / 1e3:.{d}f} k"


def table(b, b20=None):
    r, rv, bb, c4 = b["roofline"], b["roofline_valu"], b["boundary_bytes_to_bools"], b["c4_account_256"]
    c2 = b["c2_accumulator_only"]; mv, mh = c2["msm_valu"], c2["msm_hbm"]
    d = b["config"]["distinct_inputs"]
    opc = b.get("one_proof_per_call") or {}
    by = (opc.get("by_caller_threads") or {})
    lat = b.get("call_latency_ms_by_size") or {}
    cpu1 = 1e3 / b["cpu_baseline"]["single_thread_value"] if (b.get("cpu_baseline") or {}).get("single_thread_value") else None
    ref_rows = []
    if by:
        one = by.get("1", {})
        ref_rows.append(("**the reference's call shape**: ONE proof per `verify_mina_state_ffi` / `mina_verify_state` call (`/root/reference/README.md:277-279`; the reference's own "
                         "\"may take 1 second\", `README.md:573`), host bytes in, bool out",
                         f"**{one.get('ms_per_call_seen_by_a_caller', 0):.1f} ms per call** ({one.get('proofs_per_s', 0):.0f} proofs/s per caller)"
                         + (f"; the CPU restatement on one core: {cpu1:.0f} ms" if cpu1 else ""), "`one_proof_per_call.by_caller_threads[\"1\"]`"))
        more = [f"{n} callers: {k(by[n]['proofs_per_s'])}/s at {by[n]['ms_per_call_seen_by_a_caller']:.0f} ms per call" for n in sorted(by, key=int) if n != "1"]
        if more:
            ref_rows.append(("… N concurrent callers, one proof each (calls that arrive while a job runs leave together as the next job)", "; ".join(more), "`one_proof_per_call`"))
    rows = ref_rows + [
        ("**headline `value`**: full Proof-of-State verifications/s from parsed, HBM-resident proofs "
         f"({b['config']['proofs_per_step']} per step, {b['config']['pipeline_lanes']} lanes; {d['chains']} distinct chains and {d['wrap_proofs']} distinct complete wrap proofs in the batch)",
         f"**{k(b['value'])} proofs/s** ({b['ms_per_step']:.1f} ms per step over {b['steps']} steps"
         + (f"; sustained over {b['sustained']['seconds']:.1f} s: {k(b['sustained']['value'])}" if b.get("sustained") else "")
         + (f"; the driver's `--steps 20`: {k(b20['value'])}, sustained {k(b20['sustained']['value'])}" if b20 and b20.get("sustained") else (f"; the driver's `--steps 20`: {k(b20['value'])}" if b20 else "")) + ")",
         "`value`, `sustained`"),
        *([("… one call alone on the chip, by call size (legs forked, wave priorities on)",
            ", ".join(f"{n}: {lat[n]:.1f} ms" for n in sorted(lat, key=int, reverse=True)) + f"; HBM in use with {b['config']['pipeline_lanes']} lanes in flight: {b['config']['hbm_in_use_GiB']} GiB",
            "`call_latency_ms_by_size`, `config.hbm_in_use_GiB`")] if lat else []),
        ("the same through the reference's bytes (`mina_verify_state_batch`, 8192 serialized 41.6 KB proofs per call, host bytes in, bools out)",
         f"{k(bb['value'])}/s lone caller ({bb['ms_per_call']:.1f} ms per call), {k(bb['two_caller_threads']['value'])}/s two callers, {k(bb['four_caller_threads']['value'])}/s four, "
         f"{k(bb['one_call_of_65536']['value'])}/s in one call of 65 536", "`boundary_bytes_to_bools`"),
        ("… beside a caller whose every call carries ONE bad opening (culprit search on a view context)",
         f"a clean caller keeps {k(bb['one_bad_opening_per_call']['clean_caller_beside_a_searching_caller'])} of {k(bb['one_bad_opening_per_call']['clean_caller_beside_a_clean_caller'])} proofs/s; "
         f"the searching caller's call: {bb['one_bad_opening_per_call']['searching_caller_ms_per_call']:.0f} ms", "`boundary_bytes_to_bools.one_bad_opening_per_call`"),
        ("BASELINE C5's batch (4096): device-resident / through the bytes", f"{k(b['c5_4096_total_strong']['value'])}/s / {k(bb['c5_4096_per_call']['value'])}/s ({bb['c5_4096_per_call']['ms_per_call']:.1f} ms per call)",
         "`c5_4096_total_strong`, `boundary….c5_4096_per_call`"),
        ("BASELINE C4 (256 Proof-of-Account pairs per `mina_verify_account_batch` call, 256 distinct accounts)",
         f"{k(c4['lone_caller']['value'])}/s lone caller ({c4['ms_per_call']:.1f} ms per call), {k(c4['sixteen_caller_threads']['value'])}/s from 16 caller threads; beside 2 callers x 8192 state proofs: "
         f"{k(c4['beside_state_batches']['value'])} account + {k(c4['beside_state_batches']['state_proofs_per_s'])} state proofs/s", "`c4_account_256`"),
        ("BASELINE C2 alone (un-folded 2^16 Vesta MSM checks, 8 per call, 16 lanes)",
         f"**{k(c2['value'])} checks/s** = {mv['wave_instructions_per_check'] / 1e6:.1f} M wave-instructions per check at **{mv['frac']:.2f}** of VALU instruction issue; HBM: {mh['frac']:.4f} algorithmic, "
         f"{mh['traffic_frac_of_peak']:.2f} counter traffic ({mh['traffic_ratio_to_algorithmic']:.1f} x algorithmic)", "`c2_accumulator_only` (`msm_valu`, `msm_hbm`)"),
        *([("BASELINE C2 as written: ONE 2^16 Vesta accumulator check per call, alone on the chip",
            f"{c2['single_check']['wall_us']:.0f} us per call (wall), {c2['single_check']['kernel_us_sum']:.0f} us of kernels: {c2['single_check']['algorithmic_GBps_kernels']:.1f} GB/s algorithmic = "
            f"{c2['single_check']['frac_of_hbm_peak_kernels']:.4f} of the HBM peak"
            + (f"; counter traffic {c2['single_check']['traffic'] / 1e6:.0f} MB per check ({c2['single_check']['traffic_ratio_to_algorithmic']:.1f} x) = {c2['single_check']['traffic_GBps_kernels']:.0f} GB/s" if c2['single_check'].get('traffic') else ""),
            "`c2_accumulator_only.single_check`")] if c2.get("single_check") else []),
        (f"dominant kernel `{r['kernel']}<0,3>`: the binding roofline (`roofline.bound = \"{r['bound']}\"`)",
         f"{r['achieved']:.1f} of {r['peak']:.1f} T limb-MAC/s = **{r['frac']:.2f}** of the measured pure `v_mad_u64_u32` peak; **{r['frac_of_own_mix_ceiling']:.2f}** of the issue ceiling of its own "
         f"instruction mix ({rv['peak']:.1f} T); {r['avg_launch_us'] / 1e3:.1f} ms per launch of {r['states_per_launch']} states"
         + (f"; **at the sampled clock ({r['sampled_sclk_mhz']:.0f} MHz): {r['frac_at_sampled_clock']:.2f}** -- the figure that compares boxes" if r.get("frac_at_sampled_clock") else ""), "`roofline`, `roofline_valu`"),
        ("… the HBM view the metric asks for",
         f"{r['hbm']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic per launch = {r['hbm']['achieved']:.1f} GB/s = **{r['hbm']['frac']:.4f}** of 8 TB/s; counter traffic "
         f"{r['hbm']['traffic'] / 1e6:.0f} MB = {r['hbm']['traffic_ratio_to_algorithmic']:.2f} x", "`roofline.hbm`"),
        ("the pipelined step against instruction issue", f"{b['step_valu']['wave_instructions_per_step'] / 1e9:.2f} G wave-instructions per step: floor {b['step_valu']['floor_ms_per_step']:.1f} ms = "
         f"**{b['step_valu']['frac']:.2f}**" + (f" at the nominal 2.4 GHz, {b['step_valu']['frac_at_sampled_clock']:.2f} at the sampled clock" if b['step_valu'].get('frac_at_sampled_clock') else "") if b.get("step_valu") else "n/a", "`step_valu`"),
        *([("socket power and shader clock inside the timed region (amdgpu hwmon, sampled by `bench.py`; `profiles/r05_clock_power.md`)",
            f"{b['power']['socket_power_w_avg']:.0f} W average, {b['power']['socket_power_w_max']:.0f} W peak of the {b['power']['power_cap_w']:.0f} W cap; sclk {b['power']['sclk_mhz_avg']:.0f} MHz average "
            f"({b['power']['sclk_mhz_min']:.0f} – {b['power']['sclk_mhz_max']:.0f}) of the nominal 2400: the step is power-capped"
            + (f"; **{b['joules_per_proof'] * 1e3:.2f} mJ per proof**" if b.get("joules_per_proof") else ""), "`power`, `joules_per_proof`")] if b.get("power") else []),
        (f"CPU restatement, same box ({b['cpu_baseline']['cores']} usable cores): one proof per call / with the GPU job's batch fold",
         f"{b['cpu_baseline']['value']:.1f} proofs/s / {b['cpu_baseline_folded']['value']:.0f} proofs/s", "`cpu_baseline`, `cpu_baseline_folded`"),
    ]
    out = [BEGIN, "| what | value | key of the bench line |", "|---|---|---|"]
    out += [f"| {a} | {v} | {c} |" for a, v, c in rows]
    out.append(END)
    return "\n".join(out)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(ROOT, "profiles", "r06_bench.json")
    p20 = os.path.join(os.path.dirname(path), os.path.basename(path).replace("bench.json", "bench_steps20.json"))
    text = table(line_of(path), line_of(p20) if os.path.exists(p20) else None)
    if "--write" in sys.argv:
        d = os.path.join(ROOT, "DESIGN.md"); s = open(d).read()
        a, e = s.index(BEGIN), s.index(END) + len(END)
        open(d, "w").write(s[:a] + text + s[e:])
    else:
        print(text)
