# dev tool: four mina_ipa_batch_check calls on IPA_BATCH copies of a committed opening (for rocprofv3 kernel traces)
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import mina_bridge_amd as m
fx = json.load(open(os.path.join(ROOT, 'tests/golden/' + os.environ.get('IPA_FIXTURE', 'ipa_pallas_k15') + '.json')))
a = {k: (np.frombuffer(bytes.fromhex(v), dtype=np.uint8).copy() if isinstance(v, str) else v) for k, v in fx['fields'].items()}
ctx = m.MinaContext(0)
for f in (0, 1): ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(0, 65536)
rb = np.zeros(32, np.uint8); rb[:8] = 7; sb = np.zeros(32, np.uint8); sb[:8] = 9
B = int(os.environ.get('IPA_BATCH', '1'))
ops = ctx.pack_ipa_openings([a] * B)
for i in range(4): assert ctx.ipa_batch_check(0, ops, rb, sb)
