// api_consensus.hip -- Samasika chain selection between the bridge's tip and a candidate tip, host C++.
//
// Follows the reference's own specification, which is in-tree prose + figures:
//   /root/reference/README.md:619-735  (short/long-range fork rules, decentralized checkpointing, sliding-window density,
//                                       ring-shift, projected window, relative minimum window density)
//   img/consensus07.png  selectSecureChain      img/consensus08.png  selectLongerChain     img/consensus03.png  special cases
// README.md:290-294 says the verifier runs this between `candidate_tip` and `bridge_tip` before the Pickles check.
// The fields come from MinaStateProtocolStateValueStableV2.body.consensus_state (un-vendored type; the binprot parser
// is row 8f-2), so the caller passes them pre-extracted.  Branchy integer logic: it stays on the host.
#include <cstring>

#include "ctx.h"

static uint32_t window_sum(const uint32_t *w, uint32_t n) { uint32_t s = 0; for (uint32_t i = 0; i < n; ++i) s += w[i]; return s; }

// README.md "Projected window": W projected to global slot `next` = ring-shift in
//   shift_count = min(max(k - 1, 0), sub_windows_per_window)   zero densities,  k = subwindow(next) - subwindow(W)
// starting after W's most recent sub-window; k == 0: unchanged; disjoint windows: everything zeroed.
extern "C" int mina_consensus_project_window(const mina_consensus_params *p, const mina_consensus_state *s, uint32_t next_global_slot,
                                             uint32_t *out_window /* sub_windows_per_window */) {
    if (!p || !s || !out_window) return fail(MINA_ERR_ARG, "null argument");
    const uint32_t n = p->sub_windows_per_window;
    if (n == 0 || n > MINA_MAX_SUB_WINDOWS || p->slots_per_sub_window == 0) return fail(MINA_ERR_ARG, "bad consensus parameters");
    if (next_global_slot < s->curr_global_slot) return fail(MINA_ERR_ARG, "cannot project a window into the past");
    for (uint32_t i = 0; i < n; ++i) out_window[i] = s->sub_window_densities[i];
    const uint32_t sw_cur = s->curr_global_slot / p->slots_per_sub_window, sw_next = next_global_slot / p->slots_per_sub_window;
    const uint32_t k = sw_next - sw_cur;
    uint32_t shift = k > 0 ? k - 1 : 0;
    if (shift > n) shift = n;
    uint32_t i = sw_cur % n;                                   // relative sub-window of W's most recent sub-window
    while (shift--) { i = (i + 1) % n; out_window[i] = 0; }
    return MINA_OK;
}

// README.md "Relative minimum window density": project `a`'s window to the later of the two global slots and take
// min(a.min_window_density, density(projected window)).
extern "C" int mina_consensus_relative_min_window_density(const mina_consensus_params *p, const mina_consensus_state *a,
                                                          const mina_consensus_state *b, uint32_t *out) {
    if (!p || !a || !b || !out) return fail(MINA_ERR_ARG, "null argument");
    const uint32_t max_slot = a->curr_global_slot > b->curr_global_slot ? a->curr_global_slot : b->curr_global_slot;
    uint32_t w[MINA_MAX_SUB_WINDOWS];
    int rc = mina_consensus_project_window(p, a, max_slot, w);
    if (rc) return rc;
    const uint32_t d = window_sum(w, p->sub_windows_per_window);
    *out = d < a->min_window_density ? d : a->min_window_density;
    return MINA_OK;
}

// README.md "Short-range fork check": same epoch -> same lock_checkpoint in the previous (staking) epoch data; one epoch
// apart -> the later block's previous-epoch lock_checkpoint equals the earlier block's current(next)-epoch one.
extern "C" int mina_consensus_is_short_range(const mina_consensus_state *a, const mina_consensus_state *b) {
    if (!a || !b) return 0;
    if (a->epoch_count == b->epoch_count) return memcmp(a->staking_lock_checkpoint, b->staking_lock_checkpoint, 32) == 0;
    if (a->epoch_count == b->epoch_count + 1) return memcmp(a->staking_lock_checkpoint, b->next_lock_checkpoint, 32) == 0;
    if (b->epoch_count == a->epoch_count + 1) return memcmp(b->staking_lock_checkpoint, a->next_lock_checkpoint, 32) == 0;
    return 0;
}

// img/consensus08.png.  Returns 1 if the candidate is selected, 0 if the tip is kept.
static int select_longer(const mina_consensus_state *tip, const mina_consensus_state *cand) {
    if (tip->blockchain_length < cand->blockchain_length) return 1;
    if (tip->blockchain_length == cand->blockchain_length) {
        const int v = memcmp(cand->last_vrf_output_hash, tip->last_vrf_output_hash, 32);      // lexicographic
        if (v > 0) return 1;
        if (v == 0 && memcmp(cand->state_hash, tip->state_hash, 32) > 0) return 1;
    }
    return 0;
}

// img/consensus07.png for one candidate: *candidate_selected = 1 if the candidate replaces the tip.
extern "C" int mina_consensus_select_secure_chain(const mina_consensus_params *p, const mina_consensus_state *tip,
                                                  const mina_consensus_state *candidate, int *candidate_selected) {
    if (!p || !tip || !candidate || !candidate_selected) return fail(MINA_ERR_ARG, "null argument");
    if (mina_consensus_is_short_range(candidate, tip)) { *candidate_selected = select_longer(tip, candidate); return MINA_OK; }
    uint32_t td = 0, cd = 0;
    int rc;
    if ((rc = mina_consensus_relative_min_window_density(p, tip, candidate, &td))) return rc;
    if ((rc = mina_consensus_relative_min_window_density(p, candidate, tip, &cd))) return rc;
    if (cd > td) *candidate_selected = 1;
    else if (cd == td) *candidate_selected = select_longer(tip, candidate);
    else *candidate_selected = 0;
    return MINA_OK;
}
