// fib_air verifier (host only): p3_uni_stark::verify + TwoAdicFriPcs::verify + p3_fri::verifier for FibonacciAir,
// the second half of the reference's run_fib_air_zk (native/src/fib_air.rs:70-72: prove, then verify, then
// "fib_air zk ok").  Verification is a few thousand permutations — it is the verifier's role, runs on the host
// exactly as in the reference, and is not a fallback for any device kernel.  Written against the wire format
// of prover.hip; independent of the test oracle.
#include <cstring>
#include <vector>

#include "bb31.hip.h"
#include "challenger.h"
#include "common.h"
#include "poseidon2.hip.h"
#include "prover.h"

namespace p3 {

using bb::Ext;

namespace {

struct Reader {
    const uint8_t* p; size_t len, pos = 0; bool bad = false;
    uint32_t u32() { uint32_t v = 0; if (pos + 4 > len) { bad = true; return 0; } memcpy(&v, p + pos, 4); pos += 4; return v; }
    uint32_t felt() { uint32_t v = u32(); if (v >= bb::P) bad = true; return v; }
    void felts(uint32_t* w, size_t n) { for (size_t i = 0; i < n; i++) w[i] = felt(); }
    Ext ext() { Ext e; felts(e.c, 4); return e; }
    // n digests: 8 field elements each (Poseidon2) or raw [u64; 4] bytes (Keccak)
    void digests(int hash, uint32_t* w, size_t n) { if (hash == HASH_POSEIDON2) felts(w, 8 * n); else for (size_t i = 0; i < 8 * n; i++) w[i] = u32(); }
};

void hash_row(const uint32_t* items, size_t n, uint32_t out[8]) {  // PaddingFreeSponge<_, 16, 8, 8>
    uint32_t st[16] = {0};
    for (size_t i = 0; i < n; i += 8) { size_t take = n - i < 8 ? n - i : 8; memcpy(st, items + i, take * 4); p2::permute(st); }
    memcpy(out, st, 32);
}
void compress(const uint32_t* l, const uint32_t* r, uint32_t out[8]) {  // TruncatedPermutation<_, 2, 8, 16>
    uint32_t st[16]; memcpy(st, l, 32); memcpy(st + 8, r, 32); p2::permute(st); memcpy(out, st, 32);
}
// MerkleTreeMmcs::verify_batch for a single matrix
bool verify_opening(int hash, const uint32_t root[8], size_t index, const uint32_t* row, size_t width, const uint32_t* path, unsigned depth) {
    uint32_t cur[8], nxt[8];
    auto cmp = hash == HASH_KECCAK ? keccak_compress_host : compress;
    if (hash == HASH_KECCAK) keccak_hash_row_host(row, width, cur); else hash_row(row, width, cur);
    for (unsigned l = 0; l < depth; l++) {
        const uint32_t* sib = path + 8 * (size_t)l;
        if ((index >> l) & 1) cmp(sib, cur, nxt); else cmp(cur, sib, nxt);
        memcpy(cur, nxt, 32);
    }
    return memcmp(cur, root, 32) == 0;
}
size_t rev_bits_host(size_t x, unsigned bits) { size_t y = 0; for (unsigned i = 0; i < bits; i++) { y = (y << 1) | (x & 1); x >>= 1; } return y; }

}  // namespace

// 0 = accept; otherwise a positive code naming the failed check (same numbering as the error strings below).
int verify_fib_air(const uint8_t* proof, size_t len, uint64_t a_pub, uint64_t b_pub, uint64_t x_pub, uint32_t log_n,
                   const FriParams& fp, std::string* why, int hash) {
    if (hash != HASH_POSEIDON2 && hash != HASH_KECCAK) { if (why) *why = "unknown hash configuration"; return -1; }
    auto reject = [&](int code, const char* msg) { if (why) *why = msg; return code; };
    // the same parameter gates as FibProver::init: nothing below may shift by >= 32 or leave the two-adic subgroup
    if (log_n < 1 || fp.log_blowup < 1 || log_n + fp.log_blowup > bb::TWO_ADICITY) return reject(-1, "bad parameters: LDE height outside [2^2, 2^27]");
    if (fp.log_final_poly_len >= log_n && !(fp.log_final_poly_len == 0 && log_n >= 1)) return reject(-1, "bad parameters: log_final_poly_len must be below the trace's log height");
    if (fp.proof_of_work_bits > 30) return reject(-1, "bad parameters: proof_of_work_bits too large");
    if (fp.num_queries == 0) return reject(-1, "bad parameters: num_queries must be positive");
    Reader rd{proof, len};
    const uint32_t log_big = log_n + fp.log_blowup;
    const uint64_t n = 1ull << log_n;
    const uint32_t gen = bb::to_monty(bb::GEN);
    if (rd.u32() != 0x42463350u || rd.u32() != 1) return reject(1, "bad header");
    if (rd.u32() != log_n) return reject(2, "degree_bits mismatch");
    uint32_t root_t[8], root_q[8];
    rd.digests(hash, root_t, 1); rd.digests(hash, root_q, 1);
    Ext t_loc[2], t_nxt[2], q_z[4];
    if (rd.u32() != 2) return reject(3, "opened values shape");
    for (auto& e : t_loc) e = rd.ext();
    if (rd.u32() != 2) return reject(3, "opened values shape");
    for (auto& e : t_nxt) e = rd.ext();
    if (rd.u32() != 1 || rd.u32() != 4) return reject(3, "opened values shape");
    for (auto& e : q_z) e = rd.ext();
    if (rd.bad) return reject(4, "truncated proof");
    uint32_t pis[3] = {bb::to_monty((uint32_t)(a_pub % bb::P)), bb::to_monty((uint32_t)(b_pub % bb::P)), bb::to_monty((uint32_t)(x_pub % bb::P))};
    Challenger ch(hash);
    ch.observe(bb::to_monty(log_n)); ch.observe(bb::to_monty(log_n));
    ch.observe_digest(root_t); ch.observe_n(pis, 3);
    Ext alpha = ch.sample_ext();
    ch.observe_digest(root_q);
    Ext zeta = ch.sample_ext();
    const uint32_t g_n = bb::two_adic_generator(log_n);
    Ext zeta_next = bb::scale(zeta, g_n);
    {   // constraints at zeta (FibonacciAir, fib_air.rs:232-264) against the opened quotient
        Ext zh = bb::sub(bb::pow(zeta, n), bb::ext_one());
        Ext ginv = bb::ext_from_base(bb::inv(g_n));
        Ext first = bb::mul(zh, bb::inv(bb::sub(zeta, bb::ext_one())));
        Ext last = bb::mul(zh, bb::inv(bb::sub(zeta, ginv)));
        Ext trans = bb::sub(zeta, ginv);
        Ext c[5] = {bb::mul(first, bb::sub(t_loc[0], bb::ext_from_base(pis[0]))), bb::mul(first, bb::sub(t_loc[1], bb::ext_from_base(pis[1]))),
                    bb::mul(trans, bb::sub(t_loc[1], t_nxt[0])), bb::mul(trans, bb::sub(bb::add(t_loc[0], t_loc[1]), t_nxt[1])),
                    bb::mul(last, bb::sub(t_loc[1], bb::ext_from_base(pis[2])))};
        Ext folded = bb::ext_zero();
        for (auto& ck : c) folded = bb::add(bb::mul(folded, alpha), ck);
        Ext quot = bb::ext_zero();
        for (int e = 0; e < 4; e++) { Ext be = bb::ext_zero(); be.c[e] = bb::ONE; quot = bb::add(quot, bb::mul(be, q_z[e])); }
        if (!bb::eq(bb::mul(folded, bb::inv(zh)), quot)) return reject(10, "OodEvaluationMismatch");
    }
    for (auto& e : t_loc) ch.observe_ext(e);
    for (auto& e : t_nxt) ch.observe_ext(e);
    for (auto& e : q_z) ch.observe_ext(e);
    Ext al = ch.sample_ext();
    Ext alp[8]; alp[0] = bb::ext_one();
    for (int k = 1; k < 8; k++) alp[k] = bb::mul(alp[k - 1], al);
    const uint32_t n_rounds = rd.u32();
    if (rd.bad || n_rounds != log_big - fp.log_blowup - fp.log_final_poly_len) return reject(5, "commit phase length");
    std::vector<uint32_t> froots((size_t)n_rounds * 8);
    std::vector<Ext> betas(n_rounds);
    rd.digests(hash, froots.data(), froots.size() / 8);
    for (uint32_t r = 0; r < n_rounds; r++) { ch.observe_digest(&froots[(size_t)r * 8]); betas[r] = ch.sample_ext(); }
    if (rd.u32() != fp.num_queries) return reject(6, "query count");
    const size_t qstart = rd.pos;
    for (uint32_t q = 0; q < fp.num_queries && !rd.bad; q++) {  // skip to the final polynomial
        if (rd.u32() != 2) rd.bad = true;
        for (int m = 0; m < 2 && !rd.bad; m++) { rd.u32(); uint32_t w = rd.u32(); rd.pos += 4 * (size_t)w; uint32_t pl = rd.u32(); rd.pos += 32 * (size_t)pl; }
        uint32_t nr = rd.u32();
        for (uint32_t r = 0; r < nr && !rd.bad; r++) { rd.pos += 16; uint32_t pl = rd.u32(); rd.pos += 32 * (size_t)pl; }
    }
    const uint32_t fpl = rd.u32();
    if (rd.bad || fpl != (1u << fp.log_final_poly_len)) return reject(7, "final polynomial length");
    std::vector<Ext> fpoly(fpl);
    for (auto& e : fpoly) { e = rd.ext(); ch.observe_ext(e); }
    const uint32_t witness = rd.felt();
    if (rd.bad || rd.pos != len) return reject(8, "trailing or missing bytes");
    ch.observe(witness);
    if (ch.sample_bits(fp.proof_of_work_bits) != 0) return reject(11, "InvalidPowWitness");
    rd.pos = qstart;
    std::vector<uint32_t> path((size_t)(log_big + 1) * 8);
    for (uint32_t q = 0; q < fp.num_queries; q++) {
        const size_t index = ch.sample_bits(log_big);
        uint32_t trow[2], qrow[4];
        if (rd.u32() != 2) return reject(12, "query shape");  // one BatchOpening per commitment round
        if (rd.u32() != 1 || rd.u32() != 2) return reject(12, "query shape");
        rd.felts(trow, 2);
        if (rd.u32() != log_big) return reject(12, "query shape"); rd.digests(hash, path.data(), log_big);
        if (!verify_opening(hash, root_t, index, trow, 2, path.data(), log_big)) return reject(13, "trace opening");
        if (rd.u32() != 1 || rd.u32() != 4) return reject(12, "query shape");
        rd.felts(qrow, 4);
        if (rd.u32() != log_big) return reject(12, "query shape"); rd.digests(hash, path.data(), log_big);
        if (!verify_opening(hash, root_q, index, qrow, 4, path.data(), log_big)) return reject(13, "quotient opening");
        const uint32_t xi = bb::mul(gen, bb::pow(bb::two_adic_generator(log_big), rev_bits_host(index, log_big)));
        Ext d0 = bb::inv(bb::sub(zeta, bb::ext_from_base(xi))), d1 = bb::inv(bb::sub(zeta_next, bb::ext_from_base(xi)));
        Ext ro = bb::ext_zero();
        int k = 0;
        for (int j = 0; j < 2; j++, k++) ro = bb::add(ro, bb::mul(alp[k], bb::mul(bb::sub(t_loc[j], bb::ext_from_base(trow[j])), d0)));
        for (int j = 0; j < 2; j++, k++) ro = bb::add(ro, bb::mul(alp[k], bb::mul(bb::sub(t_nxt[j], bb::ext_from_base(trow[j])), d1)));
        for (int j = 0; j < 4; j++, k++) ro = bb::add(ro, bb::mul(alp[k], bb::mul(bb::sub(q_z[j], bb::ext_from_base(qrow[j])), d0)));
        if (rd.u32() != n_rounds) return reject(12, "query shape");
        Ext folded = ro;
        size_t idx = index;
        for (uint32_t r = 0; r < n_rounds; r++) {
            const uint32_t lfh = log_big - 1 - r;
            Ext sib = rd.ext();
            if (rd.u32() != lfh) return reject(12, "query shape");
            rd.digests(hash, path.data(), lfh);
            Ext ev[2];
            ev[idx & 1] = folded; ev[(idx & 1) ^ 1] = sib;
            const size_t pair = idx >> 1;
            uint32_t row8[8];
            memcpy(row8, ev[0].c, 16); memcpy(row8 + 4, ev[1].c, 16);
            if (!verify_opening(hash, &froots[(size_t)r * 8], pair, row8, 8, path.data(), lfh)) return reject(14, "FRI layer opening");
            const uint32_t s = bb::pow(bb::two_adic_generator(lfh + 1), rev_bits_host(pair, lfh));
            Ext num = bb::mul(bb::sub(betas[r], bb::ext_from_base(s)), bb::sub(ev[1], ev[0]));
            folded = bb::add(ev[0], bb::scale(num, bb::inv(bb::sub(bb::neg(s), s))));
            idx = pair;
        }
        const uint32_t lfinal = fp.log_blowup + fp.log_final_poly_len;
        const uint32_t xf = bb::pow(bb::two_adic_generator(lfinal), rev_bits_host(idx, lfinal));
        Ext evf = bb::ext_zero();
        for (uint32_t i = fpl; i-- > 0;) evf = bb::add(bb::scale(evf, xf), fpoly[i]);
        if (rd.bad) return reject(9, "truncated proof");
        if (!bb::eq(evf, folded)) return reject(15, "FinalPolyMismatch");
    }
    if (why) why->clear();
    return 0;
}


// ---- verifier of HIDING proofs (wire format version 2; prover_hiding.hip.inc): p3_uni_stark::verify with SC::Pcs::ZK over
// HidingFriPcs + MerkleTreeHidingMmcs as the reference configures them (native/src/fib_air.rs:40-72).  Leaves are the
// opened values followed by their salts, matrix by matrix; the quotient is recomposed from the four blinded chunks
// (the blinding cancels in sum_c zps_c(zeta) chunk_c(zeta)); every opened column, the random ones included, enters the
// FRI batch.  [UPSTREAM-RECALL] in structure, parity unpinned.
namespace {
constexpr uint32_t VH_NRC = 4, VH_SALT = 4, VH_D = 4, VH_TW = 2 + VH_NRC, VH_RW = VH_NRC + VH_D, VH_CH = 4;
constexpr uint32_t VH_OPEN = VH_RW + 2 * VH_TW + VH_CH * VH_D;

void hash_any(int hash, const uint32_t* items, size_t n, uint32_t out[8]) {
    if (hash == HASH_KECCAK) keccak_hash_row_host(items, n, out); else hash_row(items, n, out);
}
// one hiding BatchOpening: values per matrix, salts per matrix, sibling path; leaf = m0 || s0 || m1 || s1 ...
bool read_check_hiding_opening(Reader& rd, int hash, const uint32_t root[8], size_t index, unsigned depth, uint32_t n_mats,
                               const uint32_t* widths, uint32_t* vals, std::vector<uint32_t>& path, int* code) {
    uint32_t salts[VH_CH * VH_SALT];
    if (rd.u32() != n_mats) { *code = 12; return false; }
    size_t off = 0;
    for (uint32_t m = 0; m < n_mats; m++) {
        if (rd.u32() != widths[m]) { *code = 12; return false; }
        rd.felts(vals + off, widths[m]);
        off += widths[m];
    }
    for (uint32_t m = 0; m < n_mats; m++) {
        if (rd.u32() != VH_SALT) { *code = 12; return false; }
        rd.felts(salts + m * VH_SALT, VH_SALT);
    }
    if (rd.u32() != depth) { *code = 12; return false; }
    rd.digests(hash, path.data(), depth);
    if (rd.bad) { *code = 9; return false; }
    uint32_t row[VH_CH * (VH_D + VH_SALT) + VH_RW + VH_SALT];
    size_t p = 0;
    off = 0;
    for (uint32_t m = 0; m < n_mats; m++) {
        memcpy(row + p, vals + off, widths[m] * 4); p += widths[m]; off += widths[m];
        memcpy(row + p, salts + m * VH_SALT, VH_SALT * 4); p += VH_SALT;
    }
    uint32_t cur[8], nxt[8];
    hash_any(hash, row, p, cur);
    auto cmp = hash == HASH_KECCAK ? keccak_compress_host : compress;
    for (unsigned l = 0; l < depth; l++) {
        const uint32_t* sib = path.data() + 8 * (size_t)l;
        if ((index >> l) & 1) cmp(sib, cur, nxt); else cmp(cur, sib, nxt);
        memcpy(cur, nxt, 32);
    }
    if (memcmp(cur, root, 32) != 0) { *code = 13; return false; }
    return true;
}
}  // namespace

int verify_fib_air_hiding(const uint8_t* proof, size_t len, uint64_t a_pub, uint64_t b_pub, uint64_t x_pub, uint32_t log_n,
                          const FriParams& fp, std::string* why, int hash) {
    if (hash != HASH_POSEIDON2 && hash != HASH_KECCAK) { if (why) *why = "unknown hash configuration"; return -1; }
    auto reject = [&](int code, const char* msg) { if (why) *why = msg; return code; };
    const uint32_t log_ext = log_n + 1, log_big = log_ext + fp.log_blowup;
    if (log_n < 1 || fp.log_blowup < 1 || log_big > bb::TWO_ADICITY) return reject(-1, "bad parameters: LDE height outside the two-adic subgroup");
    if (fp.log_final_poly_len >= log_ext) return reject(-1, "bad parameters: log_final_poly_len must be below the randomized trace's log height");
    if (fp.proof_of_work_bits > 30) return reject(-1, "bad parameters: proof_of_work_bits too large");
    if (fp.num_queries == 0) return reject(-1, "bad parameters: num_queries must be positive");
    Reader rd{proof, len};
    const uint64_t h = 1ull << log_n;
    const uint32_t gen = bb::to_monty(bb::GEN);
    if (rd.u32() != 0x42463350u || rd.u32() != 2) return reject(1, "bad header");
    if (rd.u32() != log_n) return reject(2, "degree_bits mismatch");
    uint32_t root_t[8], root_q[8], root_r[8];
    rd.digests(hash, root_t, 1); rd.digests(hash, root_q, 1); rd.digests(hash, root_r, 1);
    Ext opened[VH_OPEN];  // random (8), trace @ zeta (6), trace @ zeta g (6), chunks (4 x 4)
    {
        uint32_t k = 0;
        if (rd.u32() != VH_RW) return reject(3, "opened values shape");
        for (uint32_t i = 0; i < VH_RW; i++) opened[k++] = rd.ext();
        if (rd.u32() != VH_TW) return reject(3, "opened values shape");
        for (uint32_t i = 0; i < VH_TW; i++) opened[k++] = rd.ext();
        if (rd.u32() != VH_TW) return reject(3, "opened values shape");
        for (uint32_t i = 0; i < VH_TW; i++) opened[k++] = rd.ext();
        if (rd.u32() != VH_CH) return reject(3, "opened values shape");
        for (uint32_t c = 0; c < VH_CH; c++) {
            if (rd.u32() != VH_D) return reject(3, "opened values shape");
            for (uint32_t i = 0; i < VH_D; i++) opened[k++] = rd.ext();
        }
    }
    if (rd.bad) return reject(4, "truncated proof");
    const Ext* t_z = opened + VH_RW;
    const Ext* t_zn = t_z + VH_TW;
    const Ext* q_z = t_zn + VH_TW;
    uint32_t pis[3] = {bb::to_monty((uint32_t)(a_pub % bb::P)), bb::to_monty((uint32_t)(b_pub % bb::P)), bb::to_monty((uint32_t)(x_pub % bb::P))};
    Challenger ch(hash);
    ch.observe(bb::to_monty(log_ext)); ch.observe(bb::to_monty(log_n));
    ch.observe_digest(root_t); ch.observe_n(pis, 3);
    Ext alpha = ch.sample_ext();
    ch.observe_digest(root_q);
    ch.observe_digest(root_r);
    Ext zeta = ch.sample_ext();
    const uint32_t g_h = bb::two_adic_generator(log_n);
    Ext zeta_next = bb::scale(zeta, g_h);
    {   // constraints at zeta against the quotient recomposed from the blinded chunks
        Ext zh_pow = bb::pow(zeta, h);
        Ext zh = bb::sub(zh_pow, bb::ext_one());
        Ext ginv = bb::ext_from_base(bb::inv(g_h));
        Ext first = bb::mul(zh, bb::inv(bb::sub(zeta, bb::ext_one())));
        Ext last = bb::mul(zh, bb::inv(bb::sub(zeta, ginv)));
        Ext trans = bb::sub(zeta, ginv);
        Ext c[5] = {bb::mul(first, bb::sub(t_z[0], bb::ext_from_base(pis[0]))), bb::mul(first, bb::sub(t_z[1], bb::ext_from_base(pis[1]))),
                    bb::mul(trans, bb::sub(t_z[1], t_zn[0])), bb::mul(trans, bb::sub(bb::add(t_z[0], t_z[1]), t_zn[1])),
                    bb::mul(last, bb::sub(t_z[1], bb::ext_from_base(pis[2])))};
        Ext folded = bb::ext_zero();
        for (auto& ck : c) folded = bb::add(bb::mul(folded, alpha), ck);
        // chunk cosets D_c = s_c <g_h>: s_c^h = GENERATOR^h w4^c; zps_c(zeta) = prod_{j != c} (zeta^h - s_j^h) / (s_c^h - s_j^h)
        uint32_t sh[VH_CH];
        { uint32_t gh = bb::pow(gen, h), w4 = bb::two_adic_generator(2), p = bb::ONE;
          for (uint32_t k = 0; k < VH_CH; k++) { sh[k] = bb::mul(gh, p); p = bb::mul(p, w4); } }
        Ext quot = bb::ext_zero();
        for (uint32_t ci = 0; ci < VH_CH; ci++) {
            uint32_t kc = bb::ONE;
            Ext zp = bb::ext_one();
            for (uint32_t j = 0; j < VH_CH; j++)
                if (j != ci) { kc = bb::mul(kc, bb::sub(sh[ci], sh[j])); zp = bb::mul(zp, bb::sub(zh_pow, bb::ext_from_base(sh[j]))); }
            zp = bb::scale(zp, bb::inv(kc));
            Ext v = bb::ext_zero();
            for (int e = 0; e < 4; e++) { Ext be = bb::ext_zero(); be.c[e] = bb::ONE; v = bb::add(v, bb::mul(be, q_z[ci * VH_D + e])); }
            quot = bb::add(quot, bb::mul(zp, v));
        }
        if (!bb::eq(bb::mul(folded, bb::inv(zh)), quot)) return reject(10, "OodEvaluationMismatch");
    }
    for (uint32_t k = 0; k < VH_OPEN; k++) ch.observe_ext(opened[k]);
    Ext al = ch.sample_ext();
    Ext alp[VH_OPEN]; alp[0] = bb::ext_one();
    for (uint32_t k = 1; k < VH_OPEN; k++) alp[k] = bb::mul(alp[k - 1], al);
    const uint32_t n_rounds = rd.u32();
    if (rd.bad || n_rounds != log_big - fp.log_blowup - fp.log_final_poly_len) return reject(5, "commit phase length");
    std::vector<uint32_t> froots((size_t)n_rounds * 8);
    std::vector<Ext> betas(n_rounds);
    rd.digests(hash, froots.data(), froots.size() / 8);
    for (uint32_t r = 0; r < n_rounds; r++) { ch.observe_digest(&froots[(size_t)r * 8]); betas[r] = ch.sample_ext(); }
    if (rd.u32() != fp.num_queries) return reject(6, "query count");
    const size_t qstart = rd.pos;
    {   // every query has the same length: skip to the final polynomial
        const uint32_t nm[3] = {1, 1, VH_CH}, wsum[3] = {VH_RW, VH_TW, VH_CH * VH_D};
        size_t qlen = 4 + 4;
        for (int k = 0; k < 3; k++) qlen += 4 + 4 * (size_t)(nm[k] + wsum[k]) + 4 * (size_t)nm[k] * (1 + VH_SALT) + 4 + 32 * (size_t)log_big;
        for (uint32_t r = 0; r < n_rounds; r++) qlen += 16 + 4 + 4 * VH_SALT + 4 + 32 * (size_t)(log_big - 1 - r);
        rd.pos += qlen * fp.num_queries;
        if (rd.pos > len) rd.bad = true;
    }
    const uint32_t fpl = rd.u32();
    if (rd.bad || fpl != (1u << fp.log_final_poly_len)) return reject(7, "final polynomial length");
    std::vector<Ext> fpoly(fpl);
    for (auto& e : fpoly) { e = rd.ext(); ch.observe_ext(e); }
    const uint32_t witness = rd.felt();
    if (rd.bad || rd.pos != len) return reject(8, "trailing or missing bytes");
    ch.observe(witness);
    if (ch.sample_bits(fp.proof_of_work_bits) != 0) return reject(11, "InvalidPowWitness");
    rd.pos = qstart;
    std::vector<uint32_t> path((size_t)(log_big + 1) * 8);
    const uint32_t w_r[1] = {VH_RW}, w_t[1] = {VH_TW}, w_q[VH_CH] = {VH_D, VH_D, VH_D, VH_D};
    for (uint32_t q = 0; q < fp.num_queries; q++) {
        const size_t index = ch.sample_bits(log_big);
        uint32_t rrow[VH_RW], trow[VH_TW], qrow[VH_CH * VH_D];
        int code = 0;
        if (rd.u32() != 3) return reject(12, "query shape");
        if (!read_check_hiding_opening(rd, hash, root_r, index, log_big, 1, w_r, rrow, path, &code)) return reject(code, "randomization opening");
        if (!read_check_hiding_opening(rd, hash, root_t, index, log_big, 1, w_t, trow, path, &code)) return reject(code, "trace opening");
        if (!read_check_hiding_opening(rd, hash, root_q, index, log_big, VH_CH, w_q, qrow, path, &code)) return reject(code, "quotient opening");
        const uint32_t xi = bb::mul(gen, bb::pow(bb::two_adic_generator(log_big), rev_bits_host(index, log_big)));
        Ext d0 = bb::inv(bb::sub(zeta, bb::ext_from_base(xi))), d1 = bb::inv(bb::sub(zeta_next, bb::ext_from_base(xi)));
        Ext ro = bb::ext_zero();
        uint32_t k = 0;
        for (uint32_t j = 0; j < VH_RW; j++, k++) ro = bb::add(ro, bb::mul(alp[k], bb::mul(bb::sub(opened[k], bb::ext_from_base(rrow[j])), d0)));
        for (uint32_t j = 0; j < VH_TW; j++, k++) ro = bb::add(ro, bb::mul(alp[k], bb::mul(bb::sub(opened[k], bb::ext_from_base(trow[j])), d0)));
        for (uint32_t j = 0; j < VH_TW; j++, k++) ro = bb::add(ro, bb::mul(alp[k], bb::mul(bb::sub(opened[k], bb::ext_from_base(trow[j])), d1)));
        for (uint32_t j = 0; j < VH_CH * VH_D; j++, k++) ro = bb::add(ro, bb::mul(alp[k], bb::mul(bb::sub(opened[k], bb::ext_from_base(qrow[j])), d0)));
        if (rd.u32() != n_rounds) return reject(12, "query shape");
        Ext folded = ro;
        size_t idx = index;
        for (uint32_t r = 0; r < n_rounds; r++) {
            const uint32_t lfh = log_big - 1 - r;
            Ext sib = rd.ext();
            uint32_t row[8 + VH_SALT];
            if (rd.u32() != VH_SALT) return reject(12, "query shape");
            rd.felts(row + 8, VH_SALT);
            if (rd.u32() != lfh) return reject(12, "query shape");
            rd.digests(hash, path.data(), lfh);
            Ext ev[2];
            ev[idx & 1] = folded; ev[(idx & 1) ^ 1] = sib;
            const size_t pair = idx >> 1;
            memcpy(row, ev[0].c, 16); memcpy(row + 4, ev[1].c, 16);
            if (!verify_opening(hash, &froots[(size_t)r * 8], pair, row, 8 + VH_SALT, path.data(), lfh)) return reject(14, "FRI layer opening");
            const uint32_t s = bb::pow(bb::two_adic_generator(lfh + 1), rev_bits_host(pair, lfh));
            Ext num = bb::mul(bb::sub(betas[r], bb::ext_from_base(s)), bb::sub(ev[1], ev[0]));
            folded = bb::add(ev[0], bb::scale(num, bb::inv(bb::sub(bb::neg(s), s))));
            idx = pair;
        }
        const uint32_t lfinal = fp.log_blowup + fp.log_final_poly_len;
        const uint32_t xf = bb::pow(bb::two_adic_generator(lfinal), rev_bits_host(idx, lfinal));
        Ext evf = bb::ext_zero();
        for (uint32_t i = fpl; i-- > 0;) evf = bb::add(bb::scale(evf, xf), fpoly[i]);
        if (rd.bad) return reject(9, "truncated proof");
        if (!bb::eq(evf, folded)) return reject(15, "FinalPolyMismatch");
    }
    if (why) why->clear();
    return 0;
}

}  // namespace p3
