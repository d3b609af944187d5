"""shared by the IPA tests: oracle-side instance minting and conversion to the C-ABI `mina_ipa_opening` layout"""
import numpy as np


def poseidon_pp(curve):
    from oracle import pasta_ref as R, oracle as O
    import mina_bridge_amd.poseidon_params as PP
    mds, rc = PP.default_params_ints(O.base_field_of(curve))
    return R.PoseidonParams(R.base_modulus(curve), mds, rc, PP.NAME)


def mint(curve, g, h, k, n_polys, n_points, seed, xi=None):
    from oracle import ipa_ref as I, oracle as O
    entry, sponge = I.make_instance(curve, g, O.bytes_to_point(h), poseidon_pp(curve), k, n_polys, n_points, seed, xi=xi)
    return entry, sponge


def to_abi(entry, sponge):
    """oracle instance -> dict of numpy buffers for MinaContext.ipa_batch_check"""
    from oracle import oracle as O
    op = entry["opening"]
    state, mode, count = sponge.raw()
    lr = np.concatenate([np.concatenate([O.point_to_bytes(L), O.point_to_bytes(R)]) for (L, R) in op["lr"]])
    return {
        "k": entry["k"], "lr": lr, "delta": O.point_to_bytes(op["delta"]), "sg": O.point_to_bytes(op["sg"]),
        "z1": O.int_to_le(op["z1"]), "z2": O.int_to_le(op["z2"]),
        "n_evalpoints": len(entry["evalpoints"]), "evalpoints": O.ints_to_le(entry["evalpoints"]).reshape(-1),
        "n_comms": len(entry["comms"]), "comms": np.concatenate([O.point_to_bytes(c) for c in entry["comms"]]),
        "combined_inner_product": O.int_to_le(entry["combined_inner_product"]),
        "polyscale": O.int_to_le(entry["polyscale"]), "evalscale": O.int_to_le(entry["evalscale"]),
        "sponge_state": O.ints_to_le(state).reshape(-1), "sponge_mode": mode, "sponge_count": count,
    }
