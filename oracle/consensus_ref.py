"""Python restatement of the chain-selection rules the reference specifies in README.md:619-735 and
img/consensus0{3,7,8}.png (TEST INFRASTRUCTURE ONLY).  States are plain dicts."""


def project_window(st, next_slot, v=7, n=11):
    w = list(st["window"][:n])
    sw_cur, sw_next = st["slot"] // v, next_slot // v
    k = sw_next - sw_cur
    shift = min(max(k - 1, 0), n)
    i = sw_cur % n
    for _ in range(shift):
        i = (i + 1) % n
        w[i] = 0
    return w


def relative_min_window_density(a, b, v=7, n=11):
    return min(a["min_density"], sum(project_window(a, max(a["slot"], b["slot"]), v, n)))


def is_short_range(a, b):
    if a["epoch"] == b["epoch"]:
        return a["staking_cp"] == b["staking_cp"]
    if a["epoch"] == b["epoch"] + 1:
        return a["staking_cp"] == b["next_cp"]
    if b["epoch"] == a["epoch"] + 1:
        return b["staking_cp"] == a["next_cp"]
    return False


def select_longer(tip, cand):
    if tip["length"] < cand["length"]:
        return True
    if tip["length"] == cand["length"]:
        if cand["vrf"] > tip["vrf"]:
            return True
        if cand["vrf"] == tip["vrf"] and cand["hash"] > tip["hash"]:
            return True
    return False


def select_secure_chain(tip, cand, v=7, n=11):
    if is_short_range(cand, tip):
        return select_longer(tip, cand)
    td, cd = relative_min_window_density(tip, cand, v, n), relative_min_window_density(cand, tip, v, n)
    if cd > td:
        return True
    if cd == td:
        return select_longer(tip, cand)
    return False
