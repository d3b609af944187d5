#!/bin/bash
# VALU issue slots of ONE fe52_mul: the loop body of fe52_chain from the s_nop 7 marker of tools/probes/fe52_probe.hip to the loop's branch (the loop is not unrolled:
# the body is one product), by mnemonic.  Needs only hipcc (cross-compiles): runs on the CPU box.
set -e
cd "$(dirname "$0")/../.."
OBJ=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mina_bridge_amd/csrc --cuda-device-only -S -o $OBJ/fe52.s tools/probes/fe52_probe.hip
awk '/^_Z10fe52_chain/ {f=1} f && /s_nop 7/ {g=1; next} f && g && /s_cbranch_scc0/ {exit} f && g {print}' $OBJ/fe52.s | grep -E "^\s+[vs]_|^\s+ds_|^\s+global_|^\s+buffer_" | awk '{print $1}' | sort | uniq -c | sort -rn
echo "--- totals (v_* = VALU issue slots; s_* ride the scalar unit)"
awk '/^_Z10fe52_chain/ {f=1} f && /s_nop 7/ {g=1; next} f && g && /s_cbranch_scc0/ {exit} f && g {print}' $OBJ/fe52.s | grep -E "^\s+v_" | wc -l
rm -rf $OBJ
