"""The symbol names Aligned's operator binds (README.md:277-279, 358-362): `verify_mina_state_ffi` / `verify_account_inclusion_ffi` with
fixed-size caller-owned buffers (48 KiB proof, 6 KiB public input) + used lengths, and the u32-length variants of later Aligned versions.
CPU leg: the shared library exports them (dynamic symbol table) and a plain-C consumer links against them.  GPU leg: that consumer installs the
fixture's indexes from raw binary blobs -- no Python in the process -- and verifies the committed full-size byte fixtures through the `_ffi` names."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FFI = ["verify_mina_state_ffi", "verify_account_inclusion_ffi", "verify_mina_state_ffi_u32", "verify_account_inclusion_ffi_u32"]

CONSUMER = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mina_verify.h"
/* the operator's buffers (aligned_layer/operator/mina: MAX_PROOF_SIZE, MAX_PUB_INPUT_SIZE) */
static unsigned char proof_buffer[MINA_FFI_MAX_PROOF_SIZE], pub_input_buffer[MINA_FFI_MAX_PUB_INPUT_SIZE];
static unsigned char *slurp(const char *dir, const char *name, size_t *n) { char p[4096]; snprintf(p, sizeof p, "%s/%s", dir, name); FILE *f = fopen(p, "rb"); if (!f) { *n = 0; return 0; }
  fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); rewind(f); unsigned char *b = malloc(*n + 1); if (fread(b, 1, *n, f) != *n) { fclose(f); return 0; } fclose(f); return b; }
static int load(const char *dir, const char *name, unsigned char *buf, size_t cap, size_t *len) { unsigned char *b = slurp(dir, name, len); if (!b || *len > cap) return 0; memset(buf, 0xA5, cap); memcpy(buf, b, *len); free(b); return 1; }
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const char *d = argv[1]; const int nproofs = atoi(argv[2]);
  size_t n;
  /* the fixture's synthetic wrap / step index, as raw arrays (a deployment loads the real ones: mina_verifier_index_load_json) */
  mina_verifier_index vi; memset(&vi, 0, sizeof vi);
  unsigned char *meta = slurp(d, "wrap_meta.bin", &n); if (!meta || n != 12) return 3;
  memcpy(&vi.log2_domain, meta, 4); memcpy(&vi.zk_rows, meta + 4, 4); memcpy(&vi.perm_alpha_offset, meta + 8, 4);
  vi.shifts = slurp(d, "wrap_shifts.bin", &n); vi.sigma_comm = slurp(d, "wrap_sigma.bin", &n); vi.coefficients_comm = slurp(d, "wrap_coeff.bin", &n);
  vi.selector_comm = slurp(d, "wrap_sel.bin", &n); vi.constant_term = slurp(d, "wrap_ct.bin", &n); vi.constant_term_len = n;
  mina_step_index si; memset(&si, 0, sizeof si);
  unsigned char *smeta = slurp(d, "step_meta.bin", &n); if (!smeta || n != 8) return 3;
  memcpy(&si.zk_rows, smeta, 4); memcpy(&si.n_domains, smeta + 4, 4);
  si.domain_log2 = (const uint32_t *)slurp(d, "step_domains.bin", &n); si.shifts = slurp(d, "step_shifts.bin", &n); si.constant_term = slurp(d, "step_ct.bin", &n); si.constant_term_len = n;
  mina_verify_configure(MINA_VERIFY_ALLOW_SURROGATE);
  if (mina_verify_install_verifier_index(&vi) != MINA_OK || mina_verify_install_step_index(&si) != MINA_OK) { fprintf(stderr, "install: %s\n", mina_last_error()); return 4; }
  int bad = 0;
  for (int i = 0; i < nproofs; ++i) {
    char a[32], b[32]; size_t pl, ql; snprintf(a, sizeof a, "proof%d.bin", i); snprintf(b, sizeof b, "pub%d.bin", i);
    if (!load(d, a, proof_buffer, sizeof proof_buffer, &pl) || !load(d, b, pub_input_buffer, sizeof pub_input_buffer, &ql)) return 5;
    const bool ok = verify_mina_state_ffi(proof_buffer, pl, pub_input_buffer, ql);
    const bool ok32 = verify_mina_state_ffi_u32(proof_buffer, (uint32_t)pl, pub_input_buffer, (uint32_t)ql);
    pub_input_buffer[40] ^= 1;                                              /* candidate_chain_state_hashes[0] */
    const bool tampered = verify_mina_state_ffi(proof_buffer, pl, pub_input_buffer, ql);
    pub_input_buffer[40] ^= 1;
    const bool too_long = verify_mina_state_ffi(proof_buffer, sizeof proof_buffer + 1, pub_input_buffer, ql);   /* a length beyond the buffer */
    const bool truncated = verify_mina_state_ffi(proof_buffer, pl - 1, pub_input_buffer, ql);
    const bool acct = verify_account_inclusion_ffi(proof_buffer, pl, pub_input_buffer, ql) || verify_account_inclusion_ffi_u32(proof_buffer, (uint32_t)pl, pub_input_buffer, (uint32_t)ql);   /* a state proof is no account proof */
    printf("proof %d: ffi=%d ffi_u32=%d tampered=%d too_long=%d truncated=%d as_account=%d\n", i, ok, ok32, tampered, too_long, truncated, acct);
    if (!ok || !ok32 || tampered || too_long || truncated || acct) ++bad;
  }
  if (verify_mina_state_ffi(NULL, 0, pub_input_buffer, 0) || verify_account_inclusion_ffi(NULL, 0, NULL, 0)) ++bad;
  printf("ffi_consumer %s\n", bad ? "FAILED" : "ok");
  return bad ? 1 : 0; }
'''


def build_consumer(tmp_path):
    import mina_bridge_amd as m
    src = tmp_path / "ffi_consumer.c"; src.write_text(CONSUMER)
    exe = tmp_path / "ffi_consumer"
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.dirname(m.LIB_PATH), "-lminaverify", "-Wl,-rpath," + os.path.dirname(m.LIB_PATH)])
    return exe


def test_ffi_names_are_in_the_dynamic_symbol_table(tmp_path):
    import mina_bridge_amd as m
    out = subprocess.run(["nm", "-D", "--defined-only", m.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert all(s in exported for s in FFI), [s for s in FFI if s not in exported]
    build_consumer(tmp_path)                          # -Werror: the declarations are plain C99 and every symbol resolves at link time


@pytest.mark.gpu
def test_plain_c_consumer_verifies_the_byte_fixtures_through_the_ffi_names(tmp_path):
    import numpy as np
    fxb = json.load(open(os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")))
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "statement_k15_encoded.json")))
    import mina_bridge_amd.poseidon_params as PP
    assert fxb["poseidon_constants"] == PP.NAME == fx["poseidon_constants"]
    w, s = fx["wrap_index"], fx["step_index"]
    blobs = {"wrap_meta.bin": np.array([w["log2_domain"], w["zk_rows"], w["perm_alpha_offset"]], np.uint32).tobytes(), "wrap_shifts.bin": bytes.fromhex(w["shifts"]),
             "wrap_sigma.bin": bytes.fromhex(w["sigma_comm"]), "wrap_coeff.bin": bytes.fromhex(w["coefficients_comm"]), "wrap_sel.bin": bytes.fromhex(w["selector_comm"]),
             "wrap_ct.bin": bytes.fromhex(w["constant_term"]), "step_meta.bin": np.array([s["zk_rows"], len(s["domains"])], np.uint32).tobytes(),
             "step_domains.bin": np.array(s["domains"], np.uint32).tobytes(), "step_shifts.bin": bytes.fromhex(s["shifts"]), "step_ct.bin": bytes.fromhex(s["constant_term"])}
    for i, it in enumerate(fxb["proofs"]):
        blobs[f"proof{i}.bin"] = bytes.fromhex(it["proof"]); blobs[f"pub{i}.bin"] = bytes.fromhex(it["pub"])
        assert len(blobs[f"proof{i}.bin"]) <= 48 * 1024 and len(blobs[f"pub{i}.bin"]) == 1057
    for name, data in blobs.items():
        (tmp_path / name).write_bytes(data)
    exe = build_consumer(tmp_path)
    r = subprocess.run([str(exe), str(tmp_path), str(len(fxb["proofs"]))], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ffi_consumer ok" in r.stdout, r.stdout + r.stderr
