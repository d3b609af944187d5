"""Build libminaverify.so (HIP, gfx950 only) in-tree with hipcc.  No CPU fallback exists."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libminaverify.so")
OBJ_DIR = os.path.join(HERE, "build")

SOURCES = ["host_core.hip", "api_core.hip", "api_msm.hip", "api_srs.hip", "api_sponge.hip", "api_ipa.hip", "api_wire.hip", "api_consensus.hip", "api_state.hip", "api_kimchi.hip", "api_verify.hip", "api_account.hip", "api_shard.hip", "api_pickles.hip", "api_loaders.hip"]
HEADERS = ["fp.cuh", "fp29.cuh", "ec.cuh", "ec29.cuh", "msm.cuh", "groupmap.cuh", "sponge.cuh", "lagrange.cuh", "ctx.h", "wire_state.h", "wire_proof.h", "wire_account.h", "polish.h", "loaders_text.h", "wire_pub.h", "kimchi_dev.cuh", "bpoly_mfma.cuh", "poseidon_tables.inc"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"] + os.environ.get("MINA_BUILD_EXTRA_FLAGS", "").split()   # experiments: -DMB_CHAIN_PRIO=0


def _hipcc() -> str:
    h = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(h):
        raise RuntimeError("hipcc not found: libminaverify.so can only be built with the ROCm toolchain")
    return h


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "mina_verify.h")]
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


def build_microbench() -> str:
    """the VALU instruction-rate probe (mina_bridge_amd/microbench), not part of the library"""
    out = os.path.join(HERE, "microbench")
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ftemplate-depth=2048", "-o", out, os.path.join(CSRC, "microbench.hip")])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--microbench" in sys.argv:
        print(build_microbench())
