"""bench.py's N > 1 path (BASELINE config C5): `python bench.py --gpus N` without a launcher starts its own ranks, on a box with
fewer GPUs the ranks share GPU 0 over gloo; rank 0 also runs the product's own multi-device leg (ONE process, one context per device).  The
collectives of the real-GPU path run on a 1-rank RCCL group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--steps", "2", "--warmup", "1", "--jobs", "64", "--pipeline", "2", "--no-probes", "--no-cpu-baseline"]


def json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.strip().split("\n") if ln.startswith('{"metric"')]
    assert len(lines) == 1, f"exactly ONE JSON line: {stdout[-500:]}"
    return json.loads(lines[0])


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MINA_BENCH_SHARE_GPU",
                                                               "MINA_BENCH_FORCE_DIST", "MINA_VERIFY_DEVICES", "MINA_VERIFY_DEVICE")}
    env.update(extra)
    return env


def test_gpus_2_without_a_gpu_fails_loudly():
    """no GPU, no CPU path: the launcher refuses instead of measuring something else"""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=600, env=clean_env())
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks_and_reports_them():
    """`python bench.py --gpus 2` as the driver runs it (no torch.distributed launcher): two ranks, barriers, MAX over ranks, verdict all-gather, the
    aggregate value, and the multi-device boundary leg with a tampered proof in every shard"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--boundary-jobs", "64"] + SMALL, capture_output=True, text=True, timeout=1500, env=clean_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["launcher"] == "bench.py launch_ranks"
    assert line["value"] > 0 and abs(line["value"] - 2 * 2 * 64 / (line["ms_per_step"] * 2 * 1e-3)) < 1e-6 * line["value"], "value = proofs of BOTH ranks / max-over-ranks time"
    import torch
    if torch.cuda.device_count() < 2:
        assert line["shared_gpu"] is True and line["gpus_physical"] == torch.cuda.device_count()
        assert "gloo" in line["config"]["sharding"]
    else:
        assert "shared_gpu" not in line and "RCCL" in line["config"]["sharding"]
    b = line["boundary_bytes_to_bools"]
    assert "error" not in b and b["value"] > 0, b
    ad = b["all_devices"]
    assert "error" not in ad, ad
    assert ad["n_devices"] == 2 and ad["proofs_per_call"] == 128 and ad["value_all_devices"] > 0 and b["value_all_devices"] == ad["value_all_devices"]
    assert ad["c5_4096_per_call"]["proofs_per_call"] == 4096 and ad["c5_4096_per_call"]["value"] > 0


@pytest.mark.gpu
def test_collectives_of_the_multi_gpu_path_on_a_one_rank_rccl_group():
    """the non-shared path (gloo control plane + RCCL barriers / all-gather / MAX all-reduce) on the real backend with one rank"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--no-boundary"] + SMALL, capture_output=True, text=True, timeout=1200, env=clean_env(MINA_BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json_line(r.stdout)
    assert line["n_gpus"] == 1 and "RCCL" in line["config"]["sharding"] and line["value"] > 0


@pytest.mark.gpu
def test_gpus_2_reports_the_exchange_variant():
    """with the probes on, the N > 1 line also carries SURVEY.md 8e.2 for the whole job (`exchange_variant_8e2`: ShardedStateJob over the ranks)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--jobs", "256", "--pipeline", "2", "--no-boundary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, env=clean_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json_line(r.stdout)
    x = line["exchange_variant_8e2"]
    assert line["n_gpus"] == 2 and x and x["value"] > 0 and x["proofs_per_rank_per_call"] == 256 and x["calls"] == 8


@pytest.mark.gpu
def test_gpus_2_under_torch_distributed_run():
    """the driver's own launch for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2`.
    On a box with two GPUs each rank takes its own over RCCL; with one, the ranks find out by themselves and share GPU 0 (no rank asks for a GPU that is not there)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        BENCH, "--gpus", "2", "--no-boundary"] + SMALL, capture_output=True, text=True, timeout=1500, env=clean_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["launcher"] == "torch.distributed.run" and line["value"] > 0
    import torch
    if torch.cuda.device_count() < 2:
        assert line["shared_gpu"] is True and line["gpus_physical"] == torch.cuda.device_count()
    else:
        assert "shared_gpu" not in line and "RCCL" in line["config"]["sharding"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8])
def test_preflight_reports_every_rank_and_passes_on_shared_ranks(n):
    """`bench.py --gpus N --preflight` (the check to run first on a real multi-GPU node): one JSON line with every rank's LOCAL_RANK, bound device and
    PCI bus id, the backend's rank count, one small step + the verdict all-gather; exit code 0.  On this 1-GPU box the N ranks share GPU 0 and the line says so."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--preflight"], capture_output=True, text=True, timeout=1800, env=clean_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"preflight"')][-1])
    assert line["ok"] is True and line["problems"] == [] and line["n_gpus"] == n == line["world_size"] == line["backend_ranks"] and line["all_reduce_of_ones"] == float(n)
    assert sorted(f["rank"] for f in line["ranks"]) == list(range(n)) and all(f["verdicts_accept"] and f["pci_bus_id"] for f in line["ranks"])
    assert line["step"]["verdict_all_gather_shards"] == n
    import torch
    if torch.cuda.device_count() < n:
        assert line["shared_gpu"] is True and len({f["pci_bus_id"] for f in line["ranks"]}) == torch.cuda.device_count()
    else:
        assert line["shared_gpu"] is False and len({f["pci_bus_id"] for f in line["ranks"]}) == n


@pytest.mark.gpu
def test_preflight_fails_when_two_ranks_bind_one_device_without_saying_so():
    """two ranks forced onto device 0 while claiming their own GPUs: the preflight must exit non-zero and name the ranks (a 2-GPU box would run this as a real
    mis-binding; on a 1-GPU box MINA_BENCH_PREFLIGHT_PRETEND_DISTINCT makes the shared ranks claim to be distinct)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--preflight"], capture_output=True, text=True, timeout=1800, env=clean_env(MINA_BENCH_PREFLIGHT_PRETEND_DISTINCT="1"))
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"preflight"')][-1])
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two real GPUs: the ranks ARE distinct")
    assert r.returncode != 0 and line["ok"] is False and any("same device" in p_ for p_ in line["problems"]), line


@pytest.mark.gpu
def test_c_abi_all_devices_leg_over_eight_logical_contexts():
    """the product's own multi-device path (ONE process, $MINA_VERIFY_DEVICES = 0 x 8: eight logical contexts on this box's GPU): a call's proofs cut into eight shards,
    one pipeline per context, a tampered proof in every shard fails alone, BASELINE C5's 4096-proof call over the eight"""
    r = subprocess.run([sys.executable, BENCH, "--boundary-all-devices", "0,0,0,0,0,0,0,0", "64"], capture_output=True, text=True, timeout=1800, env=clean_env(GPU_MAX_HW_QUEUES="16"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    ad = json.loads(r.stdout.strip().splitlines()[-1])
    assert "error" not in ad and "skipped" not in ad, ad
    assert ad["n_devices"] == 8 and ad["distinct_gpus"] == 1 and ad["proofs_per_call"] == 512 and ad["value_all_devices"] > 0 and ad["c5_4096_per_call"]["value"] > 0


@pytest.mark.gpu
def test_exchange_variant_leg_on_a_one_rank_rccl_group():
    """with the probes on, the forced 1-rank RCCL run also times `exchange_variant_8e2` over the real backend: nccl collectives of HBM tensors on the context's pinned
    stream -- the leg's code path on a real node (the 2-rank runs on this box move host tensors over gloo)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--no-boundary", "--no-cpu-baseline", "--steps", "2", "--warmup", "1", "--jobs", "1024", "--pipeline", "2"],
                       capture_output=True, text=True, timeout=1500, env=clean_env(MINA_BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    x = json_line(r.stdout)["exchange_variant_8e2"]
    assert x and "error" not in x and x["value"] > 0 and "RCCL" in x["collectives"], x


def test_power_sampler_reads_the_hwmon_files_of_the_rank_s_own_gpu(tmp_path):
    """bench.py's `power` key (profiles/r05_clock_power.md): socket power and sclk from amdgpu's hwmon files, sampled by a host thread inside the timed region.  On a fake
    sysfs tree with two cards the sampler picks the card whose PCI address is the rank's device, averages what it read, and is silent (None) when nothing matches."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    drm = tmp_path / "drm"; drm.mkdir()
    for i, (bdf, uw, hz) in enumerate([("0000:05:00.0", 1300000000, 2250000000), ("0000:15:00.0", 240000000, 95000000)]):
        hw = tmp_path / "pci" / bdf / "hwmon" / f"hwmon{i + 3}"; hw.mkdir(parents=True)
        (hw / "power1_input").write_text(f"{uw}\n"); (hw / "freq1_input").write_text(f"{hz}\n"); (hw / "power1_cap").write_text("1400000000\n")
        (drm / f"card{i}").mkdir(); os.symlink(tmp_path / "pci" / bdf, drm / f"card{i}" / "device")
    s = b.PowerSampler("0000:05:00.0", interval=0.01, drm_root=str(drm)).start(); time.sleep(0.1); r = s.stop()
    assert r["samples"] >= 3 and r["socket_power_w_avg"] == 1300.0 and r["sclk_mhz_avg"] == 2250.0 and r["power_cap_w"] == 1400.0
    s = b.PowerSampler("0000:15:00.0", interval=0.01, drm_root=str(drm)).start(); time.sleep(0.05); r = s.stop()
    assert r["socket_power_w_max"] == 240.0 and r["sclk_mhz_min"] == 95.0
    assert b.PowerSampler("0000:99:00.0", interval=0.01, drm_root=str(drm)).start().stop() is None      # two cards, neither is ours: no guess
    assert b.PowerSampler(None, interval=0.01, drm_root=str(drm)).start().stop() is None                # no address and more than one card: no guess
    assert b.PowerSampler(None, interval=0.01, drm_root=str(tmp_path / "nothing")).start().stop() is None
