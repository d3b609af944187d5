"""mina_bridge_amd -- MI355X-native (gfx950) Kimchi/Pickles IPA batch-verifier hot path.

The product is `libminaverify.so` (hand-written HIP, C-ABI in include/mina_verify.h); this package
is the thin Python host mirror of that ABI used by the tests and bench.py.  No CPU fallback.
"""
from .lib import (ConsensusParams, ConsensusState, MAINNET_CONSENSUS, consensus_is_short_range, consensus_project_window,
                  consensus_relative_min_window_density, consensus_select_secure_chain, combined_inner_product, parse_account_pub_inputs, parse_merkle_path, parse_state_pub_inputs,
                  CURVE_PALLAS, CURVE_VESTA, EXPORTS, FIELD_FP, FIELD_FQ, LIB_PATH, MinaContext, MinaError,
                  base_field_of, load_library, scalar_field_of)
from . import poseidon_params

__all__ = ["MinaContext", "MinaError", "load_library", "LIB_PATH", "EXPORTS", "FIELD_FP", "FIELD_FQ",
           "CURVE_PALLAS", "CURVE_VESTA", "scalar_field_of", "base_field_of", "poseidon_params"]
