#!/usr/bin/env python3
"""Host check of tools/probes/sg_probe --values: the signed-digit chain and the lazy chain end in the SAME field element; limbs 0..7 of a signed-digit result are below 2^29,
the top limb is non-negative and the value lies in (p, 3 p) after a product of values below 3 p (offset (1 p, 2 p] built into digit 8)."""
import json, sys
P = [0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001]
val = lambda l: sum(int(x) << (29 * i) for i, x in enumerate(l))
n = bad = 0; vmax = 0.0; vmin = 1e9
for ln in sys.stdin:
    t = ln.split()
    if len(t) != 29: continue
    f, it = int(t[0]), int(t[1]); p = P[f]
    a, b = t[11:20], t[20:29]
    va, vb = val(a), val(b)
    ok = va % p == vb % p and all(int(x) < (1 << 29) for x in a[:8]) and int(a[8]) < (1 << 24) and p < va < 3 * p
    vmax = max(vmax, va / p); vmin = min(vmin, va / p)
    n += 1; bad += not ok
print(json.dumps({"chains": n, "mismatches": bad, "signed_result_over_p_min": round(vmin, 4), "signed_result_over_p_max": round(vmax, 4)}))
sys.exit(1 if bad or not n else 0)
