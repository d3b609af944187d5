#!/usr/bin/env python3
"""Read the gfx950 code objects inside libminaverify.so: kernel metadata (registers, spills, scratch, LDS) and disassembly
statistics (instruction histograms of whole kernels and of their loops).

The numbers DESIGN.md quotes as "checked in the ISA" are properties of one compiler build; `tests/test_code_object.py` asserts them
through this module so that a compiler bump or a dropped launch-bound attribute fails in the CPU tier instead of halving occupancy
silently.  Only LLVM's own binutils (/opt/rocm/llvm/bin) are used; nothing is executed on a GPU.

    python tools/code_object.py                         # table of every kernel
    python tools/code_object.py --kernel 'pstate_hash_kernel<0, 3>' --loops
"""
from __future__ import annotations

import argparse
import collections
import functools
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mina_bridge_amd", "libminaverify.so")
LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/llvm/bin")

NOTE_KEYS = ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
             "group_segment_fixed_size", "max_flat_workgroup_size", "wavefront_size", "uses_dynamic_stack")


def _tool(name: str) -> str:
    p = os.path.join(LLVM_BIN, name)
    if not os.path.exists(p):
        p = shutil.which(name) or ""
    if not p:
        raise RuntimeError(f"{name} not found (LLVM_BIN={LLVM_BIN})")
    return p


def demangle(names):
    cxxfilt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not cxxfilt or not names:
        return list(names)
    out = subprocess.run([cxxfilt], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    return out[:len(names)]


def short_name(demangled: str) -> str:
    """`void mb::pstate_hash_kernel<0, 3>(unsigned int, ...)` -> `pstate_hash_kernel<0, 3>`"""
    s = demangled
    depth = 0
    for i, ch in enumerate(s):                       # cut the argument list: the first '(' at template depth 0
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            s = s[:i]
            break
    s = re.sub(r"^void\s+", "", s).strip()
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    return s.split("::")[-1] if "<" not in s else re.sub(r"^(?:\w+::)+", "", s)


class CodeObjects:
    """The code objects of one shared library, unbundled into a scratch directory that lives as long as this object."""

    def __init__(self, lib: str = LIB):
        if not os.path.exists(lib):
            raise FileNotFoundError(lib)
        self.lib = lib
        self._tmp = tempfile.TemporaryDirectory(prefix="mina_co_")
        local = os.path.join(self._tmp.name, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([_tool("llvm-objdump"), "--offloading", local], cwd=self._tmp.name, check=True, capture_output=True)
        self.files = sorted(glob.glob(os.path.join(self._tmp.name, "lib.so.*gfx950*")))
        if not self.files:
            raise RuntimeError(f"{lib} holds no gfx950 code object")
        self._kernels = None
        self._disasm = {}

    def close(self):
        self._tmp.cleanup()

    # ---- metadata ---------------------------------------------------------------------------------------------------------------
    def kernels(self) -> dict:
        """{short name: {vgpr_count, ..., 'file': code object, 'symbol': mangled}}; a kernel instantiated in several translation
        units (static msm_* helpers) appears once per distinct metadata under `name`, `name#2`, ..."""
        if self._kernels is not None:
            return self._kernels
        import yaml
        rows = []
        for f in self.files:
            txt = subprocess.run([_tool("llvm-readelf"), "--notes", f], capture_output=True, text=True, check=True).stdout
            for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", txt, flags=re.S | re.M):     # one YAML document per metadata note
                for k in (yaml.safe_load(doc) or {}).get("amdhsa.kernels", []):
                    row = {key: k.get("." + key) for key in NOTE_KEYS if "." + key in k}
                    row["file"] = f
                    rows.append(row)
        out = {}
        dem = demangle([r["name"] for r in rows])
        for r, d in zip(rows, dem):
            meta = dict(r)
            meta["symbol"] = r["name"]
            sn = short_name(d)
            key, n = sn, 1
            while key in out:
                same = all(out[key].get(k) == meta.get(k) for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"))
                if same:
                    break
                n += 1
                key = f"{sn}#{n}"
            out.setdefault(key, meta)
        self._kernels = out
        return out

    # ---- disassembly ------------------------------------------------------------------------------------------------------------
    def _disassemble(self, f: str) -> dict:
        """{mangled symbol: [(address, mnemonic, operands)]} for one code object"""
        if f in self._disasm:
            return self._disasm[f]
        txt = subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", f], capture_output=True, text=True, check=True).stdout
        funcs, cur = {}, None
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                cur = funcs.setdefault(m.group(1), [])
                continue
            if cur is None:
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if m:
                cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
        self._disasm[f] = funcs
        return funcs

    def instructions(self, kernel: str):
        meta = self.kernels()[kernel]
        return self._disassemble(meta["file"])[meta["symbol"]]

    def loops(self, kernel: str):
        """Natural loops found as backward branches: [(first index, last index)] into instructions(kernel), innermost first by
        size.  s_cbranch/s_branch encode a signed 16-bit dword offset from the NEXT instruction; llvm-objdump prints it unsigned."""
        ins = self.instructions(kernel)
        addr_to_idx = {a: i for i, (a, _, _) in enumerate(ins)}
        out = []
        for i, (a, mn, ops) in enumerate(ins):
            if not (mn.startswith("s_cbranch") or mn == "s_branch"):
                continue
            try:
                off = int(ops.split()[0])
            except (ValueError, IndexError):
                continue
            if off >= 0x8000:
                off -= 0x10000
            target = a + 4 + 4 * off
            if target <= a and target in addr_to_idx:
                out.append((addr_to_idx[target], i))
        out.sort(key=lambda se: se[1] - se[0])
        return out

    def histogram(self, kernel: str, span=None) -> collections.Counter:
        ins = self.instructions(kernel)
        if span is not None:
            ins = ins[span[0]:span[1] + 1]
        return collections.Counter(mn for _, mn, _ in ins)


# ---- statistics the tests and the bench line use --------------------------------------------------------------------------------
def is_valu(mn: str) -> bool:
    return mn.startswith("v_") and not mn.startswith(("v_mfma", "v_smfma", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_nop"))


def is_mac64(mn: str) -> bool:
    return mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32"))


def is_scratch(mn: str) -> bool:
    return mn.startswith(("scratch_", "buffer_load_dword", "buffer_store_dword")) and True


def summarize(hist: collections.Counter) -> dict:
    valu = sum(n for mn, n in hist.items() if is_valu(mn))
    return {
        "instructions": sum(hist.values()),
        "valu": valu,
        "mac64": sum(n for mn, n in hist.items() if is_mac64(mn)),
        "s_nop": hist.get("s_nop", 0),
        "scratch": sum(n for mn, n in hist.items() if mn.startswith("scratch_")),
        "ds_bpermute": hist.get("ds_bpermute_b32", 0),
        "s_waitcnt": hist.get("s_waitcnt", 0),
        "mfma": sum(n for mn, n in hist.items() if mn.startswith("v_mfma")),
        "global_load": sum(n for mn, n in hist.items() if mn.startswith("global_load")),
    }


def hottest_loop(co: CodeObjects, kernel: str, min_mac64: int = 1):
    """The innermost loop (no other loop nested inside) holding the most 64-bit multiply-accumulates."""
    loops = co.loops(kernel)
    best = None
    for (s, e) in loops:
        if any(s <= s2 and e2 <= e and (s2, e2) != (s, e) for (s2, e2) in loops):
            continue
        summ = summarize(co.histogram(kernel, (s, e)))
        if summ["mac64"] >= min_mac64 and (best is None or summ["mac64"] > best[1]["mac64"]):
            best = ((s, e), summ)
    return best


@functools.lru_cache(maxsize=1)
def compiler_version() -> str:
    try:
        out = subprocess.run([shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return "unknown"
    hip = re.search(r"HIP version:\s*(\S+)", out)
    clang = re.search(r"clang version\s*(\S+)", out)
    return f"HIP {hip.group(1) if hip else '?'} / clang {clang.group(1) if clang else '?'}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--kernel")
    ap.add_argument("--loops", action="store_true")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    co = CodeObjects(a.lib)
    try:
        ks = co.kernels()
        if a.kernel:
            meta = {k: v for k, v in ks[a.kernel].items() if k != "file"}
            res = {"kernel": a.kernel, "meta": meta, "whole": summarize(co.histogram(a.kernel))}
            if a.loops:
                res["loops"] = [{"span": [s, e], **summarize(co.histogram(a.kernel, (s, e)))} for s, e in co.loops(a.kernel)]
            print(json.dumps(res, indent=None if a.json else 1))
            return
        print(f"# {compiler_version()}  {os.path.relpath(a.lib, ROOT)}")
        print(f"{'kernel':58s} vgpr agpr vspill sspill scratch    lds")
        for name in sorted(ks):
            m = ks[name]
            print(f"{name[:58]:58s} {m.get('vgpr_count', 0):4d} {m.get('agpr_count', 0):4d} {m.get('vgpr_spill_count', 0):6d} {m.get('sgpr_spill_count', 0):6d} "
                  f"{m.get('private_segment_fixed_size', 0):7d} {m.get('group_segment_fixed_size', 0):6d}")
    finally:
        co.close()


if __name__ == "__main__":
    main()
