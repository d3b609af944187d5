# dev tool: four single-opening mina_ipa_batch_check calls (for rocprofv3 kernel traces of the B = 1 path)
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import mina_bridge_amd as m
fx = json.load(open(os.path.join(ROOT, 'tests/golden/ipa_pallas_k15.json')))
a = {k: (np.frombuffer(bytes.fromhex(v), dtype=np.uint8).copy() if isinstance(v, str) else v) for k, v in fx['fields'].items()}
ctx = m.MinaContext(0)
for f in (0, 1): ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(0, 65536)
rb = np.zeros(32, np.uint8); rb[:8] = 7; sb = np.zeros(32, np.uint8); sb[:8] = 9
for i in range(4): assert ctx.ipa_batch_check(0, [a], rb, sb)
