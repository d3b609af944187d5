"""Copies the one serialized protocol state the reference tree holds -- the base64 `MINA_TIP_PROTOCOL_STATE` constant and its
`MINA_TIP_STATE_HASH_FIELD` (core/src/utils/constants.rs:22-24) -- into tests/golden/tip_protocol_state.json.  Data only
(an input and its expected output); run in the build container where /root/reference exists."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open("/root/reference/core/src/utils/constants.rs").read()
b64 = re.search(r'MINA_TIP_PROTOCOL_STATE: &str = "([^"]+)"', src).group(1)
h = re.search(r'MINA_TIP_STATE_HASH_FIELD: &str =\s*"(\d+)"', src).group(1)
json.dump({"source": "core/src/utils/constants.rs:22-24", "protocol_state_base64": b64, "state_hash_field": h},
          open(os.path.join(ROOT, "tests/golden/tip_protocol_state.json"), "w"), indent=0)
