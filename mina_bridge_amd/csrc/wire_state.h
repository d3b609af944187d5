// wire_state.h -- host-side readers of the protocol-state record and its `to_input` flattening.
//
// Replaces, for the verifier behind core/src/aligned.rs:31-58:
//   * `MinaStateProtocolStateValueStableV2::binprot_read` (core/src/mina.rs:143-168 reads every state of the chain with it;
//     the one serialized state in the tree, core/src/utils/constants.rs:22, is consumed exactly: tests/test_protocol_state.py)
//   * the serde/bincode form of the same record inside `MinaStateProof` (core/src/proof/state_proof.rs:28-41,
//     `bincode::serialize` at core/src/aligned.rs:33) -- same field order, fixed-width integers  [UPSTREAM-RECALL for
//     mina-p2p-messages' non-human-readable serde impls: BigInt = 32 raw bytes, Number<T> = T, strings = u64 length + bytes,
//     enums = u32 variant index]
//   * mina `Protocol_state.Body.to_input` / openmina `ToInput` + `Inputs::to_fields` (pin core/Cargo.toml:23-24)  [UPSTREAM-RECALL]
// Host C++ only (microseconds per state); the Poseidon work on the flattened fields is the GPU's (api_state.hip).
#pragma once
#include <cstdint>
#include <array>
#include <cstring>
#include <vector>

namespace mw {

struct B32 { uint8_t b[32]; };
static inline bool b32_eq(const B32 &a, const B32 &b) { return memcmp(a.b, b.b, 32) == 0; }

// Fixed-capacity vector: the containers are parsed once per proof on the boundary's host threads, so nothing on that path allocates.
// push_back beyond N is dropped and sets `overflow` (the readers bound every length before they push).
template <class T, size_t N> struct SmallVec {
    T d[N]; size_t n = 0; bool overflow = false;
    void clear() { n = 0; overflow = false; }
    void push_back(const T &x) { if (n < N) d[n++] = x; else overflow = true; }
    T &emplace_back() { if (n < N) return d[n++]; overflow = true; return d[N - 1]; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return d[i]; }
    const T &operator[](size_t i) const { return d[i]; }
    T *begin() { return d; } T *end() { return d + n; }
    const T *begin() const { return d; } const T *end() const { return d + n; }
    T *data() { return d; } const T *data() const { return d; }
};
// a length-prefixed byte string that is 32 bytes in every well-formed state (longer ones are cut and keep their length: rejected later)
struct Str32 { uint8_t d[32] = {0}; size_t n = 0; const uint8_t *data() const { return d; } size_t size() const { return n; } };

static inline bool fp_canonical(const uint8_t *b) {
    static const uint8_t P_LE[32] = {0x01, 0x00, 0x00, 0x00, 0xed, 0x30, 0x2d, 0x99, 0x1b, 0xf9, 0x4c, 0x09, 0xfc, 0x98, 0x46, 0x22,
                                     0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x40};
    for (int i = 31; i >= 0; --i) { if (b[i] != P_LE[i]) return b[i] < P_LE[i]; }
    return false;
}
static inline bool fq_canonical(const uint8_t *b) {
    static const uint8_t Q_LE[32] = {0x01, 0x00, 0x00, 0x00, 0x21, 0xeb, 0x46, 0x8c, 0xdd, 0xa8, 0x94, 0x09, 0xfc, 0x98, 0x46, 0x22,
                                     0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x40};
    for (int i = 31; i >= 0; --i) { if (b[i] != Q_LE[i]) return b[i] < Q_LE[i]; }
    return false;
}

// ---------------------------------------------------------------------------------------------- the two codecs
// Both expose the same primitive readers; every record parser below is a template over the codec.
struct Cursor {
    const uint8_t *p; size_t n, pos = 0; bool ok = true;
    Cursor(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    const uint8_t *take(size_t k) { if (!ok || n - pos < k) { ok = false; return nullptr; } const uint8_t *r = p + pos; pos += k; return r; }
    uint8_t u8() { const uint8_t *q = take(1); return q ? *q : 0; }
    uint64_t le(size_t k) { const uint8_t *q = take(k); uint64_t v = 0; if (q) for (size_t i = k; i-- > 0;) v = (v << 8) | q[i]; return v; }
    void fail() { ok = false; }
};

struct Binprot : Cursor {                    // OCaml bin_prot, as mina-p2p-messages' `BinProtRead` derives it
    using Cursor::Cursor;
    uint64_t nat() {                         // Nat0 / non-negative int
        const uint8_t c = u8();
        if (c < 0x80) return c;
        if (c == 0xfe) return le(2);
        if (c == 0xfd) return le(4);
        if (c == 0xfc) return le(8);
        fail(); return 0;
    }
    uint32_t u32() { const uint64_t v = nat(); if (v > 0xffffffffull) fail(); return (uint32_t)v; }
    uint64_t u64() { return nat(); }
    int64_t i64() {                          // signed int: small non-negative inline, 0xff = negative byte, then the sized codes
        const uint8_t c = u8();
        if (c < 0x80) return c;
        if (c == 0xff) return (int8_t)u8();
        if (c == 0xfe) return (int16_t)le(2);
        if (c == 0xfd) return (int32_t)le(4);
        if (c == 0xfc) return (int64_t)le(8);
        fail(); return 0;
    }
    uint32_t variant() { return u8(); }      // constructor index of an ordinary variant: one byte
    size_t length() { return (size_t)nat(); }
    bool boolean() { const uint8_t v = u8(); if (v > 1) fail(); return v == 1; }
    bool option() { return boolean(); }
    void unit() { if (u8() != 0) fail(); }
    B32 big() { B32 r{}; const uint8_t *q = take(32); if (q) memcpy(r.b, q, 32); return r; }
    std::vector<uint8_t> string() { const size_t k = length(); const uint8_t *q = take(k); return q ? std::vector<uint8_t>(q, q + k) : std::vector<uint8_t>(); }
    Str32 str32() { Str32 r; r.n = length(); const uint8_t *q = take(r.n); if (q) memcpy(r.d, q, r.n < 32 ? r.n : 32); return r; }
    uint8_t chr() { return u8(); }
    void padded_end() { unit(); }            // `PaddedSeq<T, N>` = OCaml vector: N elements, then the unit that ends the nested pairs
};

struct Bincode : Cursor {                    // bincode 1.3 default options (fixed-width little-endian) of the serde derives
    using Cursor::Cursor;
    uint32_t u32() { return (uint32_t)le(4); }
    uint64_t u64() { return le(8); }
    int64_t i64() { return (int64_t)le(8); }
    uint32_t variant() { return (uint32_t)le(4); }
    size_t length() { const uint64_t v = le(8); if (v > n) fail(); return (size_t)v; }
    bool boolean() { const uint8_t v = u8(); if (v > 1) fail(); return v == 1; }
    bool option() { return boolean(); }
    void unit() {}
    B32 big() { B32 r{}; const uint8_t *q = take(32); if (q) memcpy(r.b, q, 32); return r; }
    std::vector<uint8_t> string() { const size_t k = length(); const uint8_t *q = take(k); return q ? std::vector<uint8_t>(q, q + k) : std::vector<uint8_t>(); }
    Str32 str32() { Str32 r; r.n = length(); const uint8_t *q = take(r.n); if (q) memcpy(r.d, q, r.n < 32 ? r.n : 32); return r; }
    uint8_t chr() { return u8(); }           // mina-p2p-messages `Char(u8)`: one byte
    void padded_end() {}                     // `PaddedSeq` = [T; N] in serde: a tuple, no terminator
};

// ---------------------------------------------------------------------------------------------- the record
struct SignedAmount { uint64_t magnitude = 0; uint8_t sgn = 0; };                // sgn: 0 = Pos, 1 = Neg
struct LocalState { B32 stack_frame, call_stack, transaction_commitment, full_transaction_commitment; SignedAmount excess, supply_increase;
                    B32 ledger; bool success; uint32_t account_update_index; bool will_succeed; };
struct Registers { B32 first_pass_ledger, second_pass_ledger, pc_data, pc_init, pc_curr; LocalState local; };
struct EpochData { B32 ledger_hash; uint64_t total_currency; B32 seed, start_checkpoint, lock_checkpoint; uint32_t epoch_length; };
struct PubKey { B32 x; bool is_odd; };
struct ProtocolState {
    B32 previous_state_hash, genesis_state_hash;
    B32 staged_ledger_hash, pending_coinbase_hash, genesis_ledger_hash;
    Str32 aux_hash, pending_coinbase_aux, body_reference, last_vrf_output;
    Registers source, target;
    B32 connecting_ledger_left, connecting_ledger_right;
    SignedAmount supply_increase, fee_excess_l, fee_excess_r; B32 fee_token_l, fee_token_r;
    uint64_t timestamp;
    uint32_t blockchain_length, epoch_count, min_window_density; SmallVec<uint32_t, 64> sub_window_densities;
    uint64_t total_currency; uint32_t slot_number, slots_per_epoch, global_slot_since_genesis;
    EpochData staking, next;
    bool has_ancestor_in_same_checkpoint_window, supercharge_coinbase; PubKey block_stake_winner, block_creator, coinbase_receiver;
    uint32_t k, c_slots_per_epoch, slots_per_sub_window, grace_period_slots, delta; uint64_t genesis_state_timestamp;
    // `Blockchain_state.snarked_ledger_hash`: the ledger hashes of MinaStatePubInputs (core/src/mina.rs:203-213)
    const B32 &snarked_ledger_hash() const { return target.first_pass_ledger; }
};

template <class C> static void rd_signed(C &c, SignedAmount &a) { a.magnitude = c.u64(); const uint32_t v = c.variant(); if (v > 1) c.fail(); a.sgn = (uint8_t)v; }
template <class C> static void rd_local(C &c, LocalState &l) {
    l.stack_frame = c.big(); l.call_stack = c.big(); l.transaction_commitment = c.big(); l.full_transaction_commitment = c.big();
    rd_signed(c, l.excess); rd_signed(c, l.supply_increase); l.ledger = c.big(); l.success = c.boolean(); l.account_update_index = c.u32();
    const size_t rows = c.length();                              // failure_status_tbl: not hashed; constructor tags only (empty in every block state)
    for (size_t i = 0; i < rows && c.ok; ++i) { const size_t m = c.length(); for (size_t j = 0; j < m && c.ok; ++j) (void)c.variant(); }
    l.will_succeed = c.boolean();
}
template <class C> static void rd_registers(C &c, Registers &g) {
    g.first_pass_ledger = c.big(); g.second_pass_ledger = c.big(); g.pc_data = c.big(); g.pc_init = c.big(); g.pc_curr = c.big(); rd_local(c, g.local);
}
template <class C> static void rd_epoch(C &c, EpochData &e) {
    e.ledger_hash = c.big(); e.total_currency = c.u64(); e.seed = c.big(); e.start_checkpoint = c.big(); e.lock_checkpoint = c.big(); e.epoch_length = c.u32();
}
template <class C> static void rd_pk(C &c, PubKey &k) { k.x = c.big(); k.is_odd = c.boolean(); }
template <class C> static uint32_t rd_tagged_u32(C &c) { if (c.variant() != 0) c.fail(); return c.u32(); }   // `Since_hard_fork of u32` / `Since_genesis of u32`

template <class C> static bool read_protocol_state(C &c, ProtocolState &s) {
    s.previous_state_hash = c.big(); s.genesis_state_hash = c.big();
    s.staged_ledger_hash = c.big(); s.aux_hash = c.str32(); s.pending_coinbase_aux = c.str32(); s.pending_coinbase_hash = c.big();
    s.genesis_ledger_hash = c.big();
    rd_registers(c, s.source); rd_registers(c, s.target);
    s.connecting_ledger_left = c.big(); s.connecting_ledger_right = c.big(); rd_signed(c, s.supply_increase);
    s.fee_token_l = c.big(); rd_signed(c, s.fee_excess_l); s.fee_token_r = c.big(); rd_signed(c, s.fee_excess_r);
    c.unit();                                                     // sok_digest
    s.timestamp = c.u64(); s.body_reference = c.str32();
    s.blockchain_length = c.u32(); s.epoch_count = c.u32(); s.min_window_density = c.u32();
    { const size_t m = c.length(); if (m > 64) c.fail(); s.sub_window_densities.clear(); for (size_t i = 0; i < m && c.ok; ++i) s.sub_window_densities.push_back(c.u32()); }
    s.last_vrf_output = c.str32(); s.total_currency = c.u64();
    s.slot_number = rd_tagged_u32(c); s.slots_per_epoch = c.u32(); s.global_slot_since_genesis = rd_tagged_u32(c);
    rd_epoch(c, s.staking); rd_epoch(c, s.next);
    s.has_ancestor_in_same_checkpoint_window = c.boolean();
    rd_pk(c, s.block_stake_winner); rd_pk(c, s.block_creator); rd_pk(c, s.coinbase_receiver);
    s.supercharge_coinbase = c.boolean();
    s.k = c.u32(); s.c_slots_per_epoch = c.u32(); s.slots_per_sub_window = c.u32(); s.grace_period_slots = c.u32(); s.delta = c.u32();
    s.genesis_state_timestamp = c.u64();
    if (!c.ok) return false;
    if (s.aux_hash.size() != 32 || s.pending_coinbase_aux.size() != 32 || s.body_reference.size() != 32 || s.last_vrf_output.size() != 32) return false;
    // every hash is a base-field element: ark's deserialiser rejects non-canonical encodings
    const B32 *fes[] = {&s.previous_state_hash, &s.genesis_state_hash, &s.staged_ledger_hash, &s.pending_coinbase_hash, &s.genesis_ledger_hash,
                        &s.source.first_pass_ledger, &s.source.second_pass_ledger, &s.source.pc_data, &s.source.pc_init, &s.source.pc_curr,
                        &s.source.local.stack_frame, &s.source.local.call_stack, &s.source.local.transaction_commitment, &s.source.local.full_transaction_commitment, &s.source.local.ledger,
                        &s.target.first_pass_ledger, &s.target.second_pass_ledger, &s.target.pc_data, &s.target.pc_init, &s.target.pc_curr,
                        &s.target.local.stack_frame, &s.target.local.call_stack, &s.target.local.transaction_commitment, &s.target.local.full_transaction_commitment, &s.target.local.ledger,
                        &s.connecting_ledger_left, &s.connecting_ledger_right, &s.fee_token_l, &s.fee_token_r,
                        &s.staking.ledger_hash, &s.staking.seed, &s.staking.start_checkpoint, &s.staking.lock_checkpoint,
                        &s.next.ledger_hash, &s.next.seed, &s.next.start_checkpoint, &s.next.lock_checkpoint,
                        &s.block_stake_winner.x, &s.block_creator.x, &s.coinbase_receiver.x};
    for (const B32 *f : fes) if (!fp_canonical(f->b)) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------- SHA-256 (staged-ledger non-snark digest)
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64]; size_t fill = 0; uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t *p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
            0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
            0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
            0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
            0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t *p, size_t n) {
        total += n;
        while (n) { const size_t k = 64 - fill < n ? 64 - fill : n; memcpy(buf + fill, p, k); fill += k; p += k; n -= k; if (fill == 64) { block(buf); fill = 0; } }
    }
    void finish(uint8_t out[32]) {
        const uint64_t bits = total * 8; uint8_t pad[72] = {0x80}; const size_t padn = (fill < 56 ? 56 : 120) - fill;
        uint8_t len[8]; for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(pad, padn); update(len, 8);
        for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};

// ---------------------------------------------------------------------------------------------- to_input
// openmina `Inputs`: whole field elements first, then (value, bits) chunks packed greedily into elements of < 255 bits.
// Allocation-free (the boundary flattens 17 states per proof on the host): fields and the packed elements live in fixed arrays and the
// greedy packing runs as the chunks arrive -- it never looks at the whole fields, so streaming it changes nothing.  `overflow` = more
// elements than a record could ever hold (callers treat it as malformed input).
struct Inputs {
    static constexpr size_t MAX_FIELDS = 128, MAX_PACKED = 64;
    B32 fields[MAX_FIELDS]; size_t nfields = 0;
    B32 packed_out[MAX_PACKED]; size_t npacked = 0;
    uint64_t cur[4] = {0, 0, 0, 0}; uint32_t nbits = 0; bool overflow = false;
    void field(const B32 &x) { if (nfields < MAX_FIELDS) fields[nfields++] = x; else overflow = true; }
    void flush() {
        if (npacked >= MAX_PACKED) { overflow = true; return; }
        B32 &r = packed_out[npacked++];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r.b[8 * i + j] = (uint8_t)(cur[i] >> (8 * j));
    }
    void shift_in(uint64_t x, uint32_t b) {                          // cur = (cur << b) + x, 1 <= b <= 64
        if (b == 64) { cur[3] = cur[2]; cur[2] = cur[1]; cur[1] = cur[0]; cur[0] = x; }
        else { cur[3] = (cur[3] << b) | (cur[2] >> (64 - b)); cur[2] = (cur[2] << b) | (cur[1] >> (64 - b)); cur[1] = (cur[1] << b) | (cur[0] >> (64 - b)); cur[0] = (cur[0] << b) | x; }
    }
    void packed(uint64_t x, uint32_t b) {
        nbits += b;
        if (nbits < 255) shift_in(x, b);
        else { flush(); cur[0] = x; cur[1] = cur[2] = cur[3] = 0; nbits = b; }
    }
    void boolean(bool b) { packed(b ? 1 : 0, 1); }
    void u32(uint32_t x) { packed(x, 32); }
    void u64(uint64_t x) { packed(x, 64); }
    // nbits single-bit chunks, least-significant bit of each byte first.  Same result as `boolean()` per bit: a run of one-bit chunks fills an
    // element up to 254 bits, so up to 64 of them are shifted in at once.
    void bytes_lsb_first(const uint8_t *p, size_t nbytes, size_t nb) {
        if (nb > nbytes * 8) nb = nbytes * 8;
        size_t i = 0;
        while (i < nb) {
            const uint32_t room = 254 - nbits;
            if (room == 0) { flush(); cur[0] = cur[1] = cur[2] = cur[3] = 0; nbits = 0; continue; }
            uint32_t t = (uint32_t)(nb - i < 64 ? nb - i : 64); if (t > room) t = room;
            uint64_t v = 0;
            for (uint32_t j = 0; j < t; ++j, ++i) v = (v << 1) | ((p[i >> 3] >> (i & 7)) & 1);
            shift_in(v, t); nbits += t;
        }
    }
    size_t count() const { return nfields + npacked + (nbits > 0 ? 1 : 0); }
    // the flattened elements into `dst` (count() * 32 bytes); returns the count
    size_t write(uint8_t *dst) {
        if (nbits > 0) { flush(); nbits = 0; cur[0] = cur[1] = cur[2] = cur[3] = 0; }
        memcpy(dst, fields, nfields * 32); memcpy(dst + nfields * 32, packed_out, npacked * 32);
        return nfields + npacked;
    }
    void to_fields(std::vector<B32> &out) {
        if (nbits > 0) { flush(); nbits = 0; cur[0] = cur[1] = cur[2] = cur[3] = 0; }
        out.assign(fields, fields + nfields); out.insert(out.end(), packed_out, packed_out + npacked);
    }
};

static inline void in_signed(Inputs &in, const SignedAmount &a) { in.u64(a.magnitude); in.boolean(a.sgn == 0); }   // Pos -> 1
static inline void in_local(Inputs &in, const LocalState &l) {
    in.field(l.stack_frame); in.field(l.call_stack); in.field(l.transaction_commitment); in.field(l.full_transaction_commitment);
    in_signed(in, l.excess); in_signed(in, l.supply_increase); in.field(l.ledger); in.u32(l.account_update_index); in.boolean(l.success); in.boolean(l.will_succeed);
}
static inline void in_registers(Inputs &in, const Registers &g) {
    in.field(g.first_pass_ledger); in.field(g.second_pass_ledger); in.field(g.pc_data); in.field(g.pc_init); in.field(g.pc_curr); in_local(in, g.local);
}
static inline void in_epoch(Inputs &in, const EpochData &e) {
    in.field(e.seed); in.field(e.start_checkpoint); in.u32(e.epoch_length); in.field(e.ledger_hash); in.u64(e.total_currency); in.field(e.lock_checkpoint);
}
static inline void in_pk(Inputs &in, const PubKey &k) { in.field(k.x); in.boolean(k.is_odd); }

// body `to_input` -> the field elements `hash_with_kimchi("MinaProtoStateBody", .)` absorbs
static inline void protocol_state_body_inputs(const ProtocolState &s, Inputs &in) {
    in.field(s.genesis_state_hash);
    {   // Staged_ledger_hash.Non_snark: SHA-256(ledger hash as 32 big-endian bytes || aux_hash || pending_coinbase_aux), bit by bit
        uint8_t be[32], dg[32]; for (int i = 0; i < 32; ++i) be[i] = s.staged_ledger_hash.b[31 - i];
        Sha256 h; h.update(be, 32); h.update(s.aux_hash.data(), s.aux_hash.size()); h.update(s.pending_coinbase_aux.data(), s.pending_coinbase_aux.size()); h.finish(dg);
        in.bytes_lsb_first(dg, 32, 256);
    }
    in.field(s.pending_coinbase_hash); in.field(s.genesis_ledger_hash);
    in_registers(in, s.source); in_registers(in, s.target);
    in.field(s.connecting_ledger_left); in.field(s.connecting_ledger_right); in_signed(in, s.supply_increase);
    in.field(s.fee_token_l); in_signed(in, s.fee_excess_l); in.field(s.fee_token_r); in_signed(in, s.fee_excess_r);
    in.u64(s.timestamp); in.bytes_lsb_first(s.body_reference.data(), s.body_reference.size(), 256);
    in.u32(s.blockchain_length); in.u32(s.epoch_count); in.u32(s.min_window_density);
    for (uint32_t d : s.sub_window_densities) in.u32(d);
    in.bytes_lsb_first(s.last_vrf_output.data(), s.last_vrf_output.size(), 253);      // truncated VRF output
    in.u64(s.total_currency); in.u32(s.slot_number); in.u32(s.slots_per_epoch); in.u32(s.global_slot_since_genesis);
    in.boolean(s.has_ancestor_in_same_checkpoint_window); in.boolean(s.supercharge_coinbase);
    in_epoch(in, s.staking); in_epoch(in, s.next);
    in_pk(in, s.block_stake_winner); in_pk(in, s.block_creator); in_pk(in, s.coinbase_receiver);
    in.u32(s.k); in.u32(s.delta); in.u32(s.c_slots_per_epoch); in.u32(s.slots_per_sub_window); in.u32(s.grace_period_slots); in.u64(s.genesis_state_timestamp);
}
static inline void protocol_state_body_fields(const ProtocolState &s, std::vector<B32> &out) { Inputs in; protocol_state_body_inputs(s, in); in.to_fields(out); }

// 20-byte '*'-padded hash prefix as a little-endian field element (mina `Hash_prefix_create.salt` input)
static inline B32 prefix_field(const char *s) {
    B32 r{}; size_t n = strlen(s); for (size_t i = 0; i < 20; ++i) r.b[i] = (uint8_t)(i < n ? s[i] : '*'); return r;
}

}  // namespace mw
