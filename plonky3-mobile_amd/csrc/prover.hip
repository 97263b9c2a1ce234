// fib_air STARK prover on gfx950: everything O(N) stays in HBM; the host only runs the transcript.
//
// Stands in for the call chain the reference drives at native/src/fib_air.rs:60-70
//   p3_uni_stark::prove -> TwoAdicFriPcs::{commit, open} -> p3_fri::prove (commit phase, grind, queries)
// instantiated as north_star asks (BabyBear, Poseidon2 MMCS + DuplexChallenger<_, Perm16, 16, 8>, non-hiding), or —
// hash = HASH_KECCAK — with the reference's own hashes (fib_air.rs:28-53: Keccak MMCS + SerializingChallenger32 over
// a Keccak-256 HashChallenger; still non-hiding).
// Device: trace (fib_air.hip), coset LDEs (ntt.hip), Merkle trees (mmcs.hip) and the kernels below:
//   selectors table, quotient values, inverse denominators 1/(z - x), barycentric openings,
//   reduced (DEEP) openings, FRI folds, proof-of-work search, batched query gathers.
// Host (this file): DuplexChallenger (a handful of permutations per proof), parameter bookkeeping,
//   proof serialisation.  Wire format: u32 LE Montgomery words, u32 vector counts, Plonky3 struct order.
// All of these passes are HBM- or integer-VALU-bound element-wise / reduction kernels: one lane per row,
// 16-byte accesses, no MFMA.
#include <chrono>
#include <deque>
#include <memory>
#include <cstring>

#include "bb31.hip.h"
#include "challenger.h"
#include "common.h"
#include "keccak.hip.h"
#include "mmcs.h"
#include "poseidon2.hip.h"
#include "poseidon2_f64.hip.h"
#include "prover.h"
#include "rng.h"
#include "rng_dev.hip.h"
#include "transcript.hip.h"

namespace p3 {

using bb::Ext;

// ------------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tl(const TwoLevelTable& t, uint32_t e) {
    return bb::mul(t.lo[e & ((1u << t.T) - 1u)], t.hi[e >> t.T]);
}
__device__ __forceinline__ uint32_t brev(uint32_t v, uint32_t bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

__device__ __forceinline__ Ext ld_ext(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Ext{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void st_ext(uint32_t* p, const Ext& e) {
    *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

// selectors_on_coset for the trace domain <g_n> evaluated on GENERATOR*<g_n> (quotient degree 1):
// sel[i] = (Z_H/(x_i - 1), Z_H/(x_i - g^-1)), x_i = 31 g^i.  Depends only on log_n: cached per context.
// One lane inverts 2*SEL_CHUNK values with Montgomery's trick.
constexpr int SEL_CHUNK = 8;
__global__ void selectors_kernel(TwoLevelTable roots, uint32_t n, uint32_t gen, uint32_t ginv, uint32_t zh, uint2* sel) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i0 = t * SEL_CHUNK;
    if (i0 >= n) return;
    uint32_t v[2 * SEL_CHUNK], pre[2 * SEL_CHUNK];
    uint32_t acc = bb::ONE;
#pragma unroll
    for (int k = 0; k < SEL_CHUNK; k++) {
        uint32_t x = bb::mul(gen, tl(roots, (i0 + k) & (n - 1)));
        v[2 * k] = bb::sub(x, bb::ONE);
        v[2 * k + 1] = bb::sub(x, ginv);
    }
#pragma unroll
    for (int k = 0; k < 2 * SEL_CHUNK; k++) { pre[k] = acc; acc = bb::mul(acc, v[k]); }
    uint32_t inv = bb::mul(bb::inv(acc), zh);
#pragma unroll
    for (int k = 2 * SEL_CHUNK - 1; k >= 0; k--) {
        uint32_t r = bb::mul(inv, pre[k]);
        inv = bb::mul(inv, v[k]);
        v[k] = r;
    }
#pragma unroll
    for (int k = 0; k < SEL_CHUNK; k++)
        if (i0 + k < n) sel[i0 + k] = make_uint2(v[2 * k], v[2 * k + 1]);
}

// quotient_values for FibonacciAir (fib_air.rs:232-264), natural order out (n x 4 base words).  The public values and
// the powers of alpha come from the device transcript.
struct QuotArgs {
    const uint2* lde;   // committed trace LDE, bit-reversed rows; first n rows = the quotient domain
    const uint2* sel;
    uint32_t* out;
    const DevState* ds;
    TwoLevelTable roots;  // w_n^i
    uint32_t n, log_n, gen, ginv, zh_inv;
};
__global__ void __launch_bounds__(256) fib_quotient_kernel(QuotArgs a) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const DevState* __restrict__ ds = a.ds;
    uint2 loc = a.lde[brev(i, a.log_n)];
    uint2 nxt = a.lde[brev((i + 1) & (a.n - 1), a.log_n)];
    uint2 s = a.sel[i];
    uint32_t x = bb::mul(a.gen, tl(a.roots, i));
    uint32_t trans = bb::sub(x, a.ginv);
    uint32_t c0 = bb::mul(s.x, bb::sub(loc.x, ds->pis[0]));
    uint32_t c1 = bb::mul(s.x, bb::sub(loc.y, ds->pis[1]));
    uint32_t c2 = bb::mul(trans, bb::sub(loc.y, nxt.x));
    uint32_t c3 = bb::mul(trans, bb::sub(bb::add(loc.x, loc.y), nxt.y));
    uint32_t c4 = bb::mul(s.y, bb::sub(loc.y, ds->pis[2]));
    Ext acc;
#pragma unroll
    for (int k = 0; k < 4; k++)  // five products per coefficient: two dot2 and one product
        acc.c[k] = bb::add(bb::add(bb::dot2(ds->apow[4].c[k], c0, ds->apow[3].c[k], c1), bb::dot2(ds->apow[2].c[k], c2, ds->apow[1].c[k], c3)),
                           bb::mul(ds->apow[0].c[k], c4));
    st_ext(a.out + 4 * (size_t)i, bb::scale(acc, a.zh_inv));
}

// compute_inverse_denominators: d0[j] = 1/(z0 - x_j), d1[j] = 1/(z1 - x_j), x_j = 31 g_big^bitrev(j),
// over the whole LDE domain in committed (bit-reversed) order.
// Only the constant coefficient of z - x varies, so the quartic inverse is taken through the tower
// F_p[Y]/(Y^2 - 11), X^2 = Y:  a = A + B X with A = a0 + z2 Y, B = z1 + z3 Y,  1/a = (A - B X) / (A^2 - Y B^2).
// Y B^2 is a constant of the point z; A^2 - Y B^2 = (a0^2 + k0) + (k1 a0 - c1) Y =: d0 + d1 Y; its inverse is
// (d0 - d1 Y) / (d0^2 - 11 d1^2), and those BASE-FIELD norms are batch-inverted per lane (Montgomery's trick).
// 24 base products per inverse instead of the 57 of a batched quartic-extension inversion; same field elements.
constexpr int DEN_CHUNK = 4;
__global__ void __launch_bounds__(256) inv_denoms_kernel(TwoLevelTable roots, uint32_t big, uint32_t log_big, uint32_t gen,
                                                         const DevState* __restrict__ ds, uint32_t* d0, uint32_t* d1) {
    const DenConsts k0 = ds->k0, k1 = ds->k1;  // constants of zeta and zeta * g from the device transcript
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t j0 = t * DEN_CHUNK;
    if (j0 >= big) return;
    uint32_t a0[2 * DEN_CHUNK], e0[2 * DEN_CHUNK], e1[2 * DEN_CHUNK], nrm[2 * DEN_CHUNK], pre[2 * DEN_CHUNK];
#pragma unroll
    for (int k = 0; k < DEN_CHUNK; k++) {
        uint32_t x = bb::mul(gen, tl(roots, brev((j0 + k) & (big - 1), log_big)));
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const DenConsts& c = z ? k1 : k0;
            const uint32_t a = bb::sub(c.z0, x);
            const uint32_t dd0 = bb::add(bb::sqr(a), c.k0), dd1 = bb::sub(bb::mul(c.k1, a), c.c1);
            a0[2 * k + z] = a; e0[2 * k + z] = dd0; e1[2 * k + z] = dd1;
            nrm[2 * k + z] = bb::sub(bb::sqr(dd0), bb::mul(bb::W_MONTY, bb::sqr(dd1)));
        }
    }
    uint32_t acc = bb::ONE;
#pragma unroll
    for (int k = 0; k < 2 * DEN_CHUNK; k++) { pre[k] = acc; acc = bb::mul(acc, nrm[k]); }
    uint32_t inv = bb::inv(acc);
#pragma unroll
    for (int k = 2 * DEN_CHUNK - 1; k >= 0; k--) {
        const uint32_t r = bb::mul(inv, pre[k]);
        inv = bb::mul(inv, nrm[k]);
        nrm[k] = r;  // 1 / norm
    }
#pragma unroll
    for (int k = 0; k < DEN_CHUNK; k++) {
        if (j0 + k >= big) break;
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const DenConsts& c = z ? k1 : k0;
            const uint32_t i = 2 * k + z;
            const uint32_t f0 = bb::mul(e0[i], nrm[i]), f1 = bb::neg(bb::mul(e1[i], nrm[i]));  // 1/D = f0 + f1 Y
            // (A - B X)(f0 + f1 Y): 1 and X^2 from A (a0 + z2 Y), X and X^3 from -B (z1 + z3 Y)
            Ext r;
            r.c[0] = bb::add(bb::mul(a0[i], f0), bb::mul(c.z2w, f1));
            r.c[2] = bb::add(bb::mul(a0[i], f1), bb::mul(c.z2, f0));
            r.c[1] = bb::neg(bb::add(bb::mul(c.z1, f0), bb::mul(c.z3w, f1)));
            r.c[3] = bb::neg(bb::add(bb::mul(c.z1, f1), bb::mul(c.z3, f0)));
            st_ext((z ? d1 : d0) + 4 * (size_t)(j0 + k), r);
        }
    }
}

// interpolate_coset (barycentric) for all three openings at once: partial sums over the low coset
//   acc[0..1] += x d0 trace[j][c],  acc[2..3] += x d1 trace[j][c],  acc[4..7] += x d0 quot[j][c]
// written per workgroup; the host adds the partials and applies (z^n - s^n)/(n s^n).
constexpr int BARY_BLOCK = 256;
__global__ void __launch_bounds__(BARY_BLOCK) barycentric_kernel(TwoLevelTable roots, uint32_t n, uint32_t log_big,
                                                                 uint32_t gen, const uint2* lde_t, const uint4* lde_q,
                                                                 const uint32_t* d0, const uint32_t* d1,
                                                                 uint32_t* partials) {
    Ext acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = bb::ext_zero();
    // two rows per step: acc += e_j v_j + e_j' v_j' under ONE Montgomery reduction per coefficient (bb::dot2); a lane whose second
    // row lies past the end pairs its row with a zero weight
    const uint32_t step = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += 2 * step) {
        const bool two = j + step < n;
        const uint32_t j2 = two ? j + step : j;
        const uint32_t xa = bb::mul(gen, tl(roots, brev(j, log_big))), xb = two ? bb::mul(gen, tl(roots, brev(j2, log_big))) : 0u;
        const Ext e0a = bb::scale(ld_ext(d0 + 4 * (size_t)j), xa), e1a = bb::scale(ld_ext(d1 + 4 * (size_t)j), xa);
        const Ext e0b = bb::scale(ld_ext(d0 + 4 * (size_t)j2), xb), e1b = bb::scale(ld_ext(d1 + 4 * (size_t)j2), xb);
        const uint2 ta = lde_t[j], tb = lde_t[j2];
        const uint4 qa = lde_q[j], qb = lde_q[j2];
        auto fma2 = [](Ext& s, const Ext& ea, uint32_t va, const Ext& eb, uint32_t vb) {
#pragma unroll
            for (int c = 0; c < 4; c++) s.c[c] = bb::add(s.c[c], bb::dot2(ea.c[c], va, eb.c[c], vb));
        };
        fma2(acc[0], e0a, ta.x, e0b, tb.x);
        fma2(acc[1], e0a, ta.y, e0b, tb.y);
        fma2(acc[2], e1a, ta.x, e1b, tb.x);
        fma2(acc[3], e1a, ta.y, e1b, tb.y);
        fma2(acc[4], e0a, qa.x, e0b, qb.x);
        fma2(acc[5], e0a, qa.y, e0b, qb.y);
        fma2(acc[6], e0a, qa.z, e0b, qb.z);
        fma2(acc[7], e0a, qa.w, e0b, qb.w);
    }
    __shared__ uint32_t red[BARY_BLOCK / 64][32];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint32_t v = acc[k].c[c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = bb::add(v, (uint32_t)__shfl_down((int)v, off, 64));
            if (lane == 0) red[wave][4 * k + c] = v;
        }
    __syncthreads();
    if (threadIdx.x < 32) {
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < BARY_BLOCK / 64; w++) v = bb::add(v, red[w][threadIdx.x]);
        partials[blockIdx.x * 32 + threadIdx.x] = v;
    }
}

// reduced openings over the LDE domain (TwoAdicFriPcs::open):
//   ro[j] = (ry0 - rt) d0 + alpha^2 (ry1 - rt) d1 + alpha^4 (ry2 - rq) d0,
//   rt = sum_c alpha^c trace[j][c], rq = sum_c alpha^c quot[j][c],
// evaluated as  d0 * [(ry0 + alpha^4 ry2) - rt - sum_c alpha^(4+c) quot_c] + d1 * [alpha^2 ry1 - sum_c alpha^(2+c) trace_c]:
// two extension products and 32 extension-by-base products instead of five and 24 (64 vs 104 base products).
struct ReducedArgs {
    const uint2* lde_t;
    const uint4* lde_q;
    const uint32_t* d0;
    const uint32_t* d1;
    uint32_t* ro;
    const DevState* ds;  // alp[0..8), y02 = ry0 + alpha^4 ry2, y1 = alpha^2 ry1
    uint32_t big;
};
__global__ void __launch_bounds__(256) reduced_openings_kernel(ReducedArgs a) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.big) return;
    const DevState* __restrict__ ds = a.ds;
    uint2 t = a.lde_t[j];
    uint4 q = a.lde_q[j];
    // u = t.x + alp1 t.y + alp4 q.x + alp5 q.y + alp6 q.z + alp7 q.w, w = alp2 t.x + alp3 t.y: extension-by-base products summed two per
    // Montgomery reduction (bb::dot2: 6 instructions for two terms instead of 2 x 5 + 3)
    Ext u, w;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        u.c[c] = bb::add(bb::add(bb::dot2(ds->alp[4].c[c], q.x, ds->alp[5].c[c], q.y), bb::dot2(ds->alp[6].c[c], q.z, ds->alp[7].c[c], q.w)),
                         bb::mul(ds->alp[1].c[c], t.y));
        w.c[c] = bb::dot2(ds->alp[2].c[c], t.x, ds->alp[3].c[c], t.y);
    }
    u.c[0] = bb::add(u.c[0], t.x);
    Ext e0 = ld_ext(a.d0 + 4 * (size_t)j), e1 = ld_ext(a.d1 + 4 * (size_t)j);
    Ext r = bb::add(bb::mul(bb::sub(ds->y02, u), e0), bb::mul(bb::sub(ds->y1, w), e1));
    st_ext(a.ro + 4 * (size_t)j, r);
}

// TwoAdicFriFolding::fold_matrix: out[i] = (lo + hi)/2 + (beta/2) g^-bitrev(i) (lo - hi); beta/2 from the transcript
__global__ void __launch_bounds__(256) fri_fold_kernel(TwoLevelTable inv_roots /* w_len^-e */, const uint32_t* in,
                                                       uint32_t* out, uint32_t half, uint32_t log_half,
                                                       const DevState* __restrict__ ds, uint32_t round, uint32_t one_half) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Ext half_beta = ds->half_beta[round];
    Ext lo = ld_ext(in + 8 * (size_t)i), hi = ld_ext(in + 8 * (size_t)i + 4);
    uint32_t p = tl(inv_roots, brev(i, log_half));
    Ext s = bb::scale(bb::add(lo, hi), one_half);
    Ext d = bb::mul(bb::scale(half_beta, p), bb::sub(lo, hi));
    st_ext(out + 4 * (size_t)i, bb::add(s, d));
}

// GrindingChallenger::grind: smallest canonical w with sample_bits(bits) == 0 after observe(w).  The sponge state
// with the pending inputs comes from the device transcript; blocks whose whole range lies above a witness already
// found return at once (the smallest one wins through atomicMin, as in a serial search).
__global__ void __launch_bounds__(256) grind_kernel(DevState* ds, uint32_t mask, uint32_t base) {
    uint32_t w = base + blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= bb::P) return;
    if (base + blockIdx.x * blockDim.x > *(volatile uint32_t*)&ds->grind_result) return;
    // fp64 form of the permutation (poseidon2_f64.hip.h: canonical values in doubles), as in the tree kernels
    const uint32_t pos = ds->n_in;  // the witness is the next observed element
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = p2f::load_elem((i < 8 && (uint32_t)i < pos) ? ds->inb[i & 7] : ds->st[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = (i == (int)pos) ? (double)w : s[i];
    p2f::permute(s);
    if ((bb::from_monty(p2f::store_elem(s[7])) & mask) == 0) atomicMin(&ds->grind_result, w);
}

// The same search for the Keccak-256 HashChallenger: candidate w is observed as the 4 little-endian bytes of its
// Montgomery word after the pending input bytes; the transcript has absorbed the complete 136-byte blocks already and
// hands over the state plus up to two template blocks (DevState::kgrind).  sample_bits pops digest bytes from the
// back, four per try, masks to 31 bits and retries while >= P.
__global__ void __launch_bounds__(256) grind_keccak_kernel(DevState* ds, uint32_t mask, uint32_t base) {
    uint32_t w = base + blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= bb::P) return;
    if (base + blockIdx.x * blockDim.x > *(volatile uint32_t*)&ds->grind_result) return;
    const KeccakGrindArgs& a = ds->kgrind;
    const uint32_t wm = bb::to_monty(w);
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = a.state[i];
    for (uint32_t blk = 0; blk < a.n_blocks; blk++) {
#pragma unroll
        for (int i = 0; i < 17; i++) {
            uint64_t lane = a.block[blk][i];
            // witness bytes that fall into this lane: byte b of the word sits at template offset wpos + b
            const int32_t rel = (int32_t)(a.wpos) - (int32_t)(blk * 136 + i * 8);  // offset of byte 0 within the lane
            if (rel > -4 && rel < 8) {
                const uint64_t v = (uint64_t)wm;
                lane ^= rel >= 0 ? (v << (8 * rel)) : (v >> (8 * -rel));
            }
            st[i] ^= lane;
        }
        if (blk + 1 == a.n_blocks) kk::permute_digest(st);  // only the digest words are read below
        else kk::permute(st);
    }
    // digest bytes d[0..31] = lanes 0..3 little endian; pops come from d[31] downwards
    bool ok = false;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint64_t lane = st[3 - t / 2];
        const uint32_t hi_or_lo = (t & 1) ? (uint32_t)lane : (uint32_t)(lane >> 32);  // bytes 4k+3 .. 4k, k = 7 - t
        const uint32_t v = __builtin_bswap32(hi_or_lo) & 0x7fffffffu;                // first pop = low byte
        if (v < bb::P) { ok = (v & mask) == 0; break; }
    }
    if (ok) atomicMin(&ds->grind_result, w);
}

// Batched Mmcs::open_batch for the query phase: block (q, t) copies tree t's opened row and sibling path
// for query q into its fixed slot of the staging buffer.
struct QTree {
    const uint32_t* mat;
    const uint32_t* layers;
    uint32_t width, log_height, shift, slot_off;  // index used = query_index >> shift; slot_off in words
};
__global__ void query_gather_kernel(const QTree* trees, uint32_t n_trees, const uint32_t* indices, uint32_t slot_words,
                                    uint32_t* out) {
    P3_LATENCY_BOUND_KERNEL();
    const QTree t = trees[blockIdx.y];
    // masked: whatever the index buffer holds, the gather stays inside the tree
    uint64_t index = (indices[blockIdx.x] >> t.shift) & ((1ull << t.log_height) - 1ull);
    uint32_t* dst = out + (size_t)blockIdx.x * slot_words + t.slot_off;
    for (uint32_t c = threadIdx.x; c < t.width; c += blockDim.x) dst[c] = t.mat[index * t.width + c];
    uint64_t base = 0, len = 1ull << t.log_height;
    for (uint32_t i = 0; i < t.log_height; i++) {
        uint64_t sib = (index >> i) ^ 1;
        if (threadIdx.x < 8) dst[t.width + i * 8 + threadIdx.x] = t.layers[base + sib * 8 + threadIdx.x];
        base += len * 8;
        len >>= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// transcript kernels: one wavefront each, between the bulk kernels, on the same stream (transcript.hip.h)
// ------------------------------------------------------------------------------------------------
// Proof staging buffer (device words; copied to the host once per proof):
//   root_t[8] root_q[8] opened[8][4] froots[n_rounds][8] fpoly[fpl][4] witness status qidx[nq] (pad) slots[nq][slot_words]
struct StageLayout {
    uint32_t root_t = 0, root_q = 8, opened = 16, froots = 48, fpoly = 0, witness = 0, status = 0, qidx = 0, slots = 0, words = 0;
};
struct TsArgs {
    DevState* ds;
    uint32_t* ps;
    int kind;
    StageLayout lay;
};

// observe the instance, sample alpha: p3_uni_stark::prove up to the quotient computation
__global__ void __launch_bounds__(64) ts_begin_kernel(TsArgs a, const uint32_t* trace, uint32_t n, uint32_t log_n) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ KState ks;
    DevChal ch;
    ch.begin(a.kind, a.ds, &ks, true);
    const uint32_t pis[3] = {trace[0], trace[1], trace[2 * (size_t)(n - 1) + 1]};  // first row and last right value
    ch.observe(bb::to_monty(log_n));  // log_ext_degree
    ch.observe(bb::to_monty(log_n));  // log_degree
    ch.observe_n(a.ps + a.lay.root_t, 8);
    ch.observe_n(pis, 3);
    const Ext alpha = ch.sample_ext();
    Ext ap[5];
    ap[0] = bb::ext_one();
#pragma unroll
    for (int k = 1; k < 5; k++) ap[k] = bb::mul(ap[k - 1], alpha);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; k++) a.ds->pis[k] = pis[k];
        for (int k = 0; k < 5; k++) a.ds->apow[k] = ap[k];
        a.ds->status = 0;
        a.ps[a.lay.status] = 0;
    }
    ch.end();
}
// observe the quotient commitment, sample zeta
__global__ void __launch_bounds__(64) ts_zeta_kernel(TsArgs a, uint32_t g_n) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ KState ks;
    DevChal ch;
    ch.begin(a.kind, a.ds, &ks, false);
    ch.observe_n(a.ps + a.lay.root_q, 8);
    const Ext zeta = ch.sample_ext();
    const Ext zeta_next = bb::scale(zeta, g_n);
    if (threadIdx.x == 0) {
        a.ds->zeta = zeta;
        a.ds->zeta_next = zeta_next;
        a.ds->k0 = den_consts(zeta);
        a.ds->k1 = den_consts(zeta_next);
    }
    ch.end();
}
// finish the barycentric sums into the opened values, observe them, sample the batching challenge
constexpr int TS_OPEN_THREADS = 256;
__global__ void __launch_bounds__(TS_OPEN_THREADS) ts_open_kernel(TsArgs a, const uint32_t* partials, uint32_t n_blocks,
                                                                  uint32_t log_n, uint32_t sn, uint32_t denom) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ KState ks;
    __shared__ uint32_t red[TS_OPEN_THREADS / 32][32];
    __shared__ uint32_t tot[32];
    {
        const uint32_t col = threadIdx.x & 31u, part = threadIdx.x >> 5;
        uint32_t v = 0;
        constexpr uint32_t STEP = TS_OPEN_THREADS / 32;
        uint32_t b = part;
        for (; b + 3 * STEP < n_blocks; b += 4 * STEP) {  // four independent loads in flight per lane
            const uint32_t p0 = partials[b * 32 + col], p1 = partials[(b + STEP) * 32 + col],
                           p2 = partials[(b + 2 * STEP) * 32 + col], p3 = partials[(b + 3 * STEP) * 32 + col];
            v = bb::add(bb::add(v, bb::add(p0, p1)), bb::add(p2, p3));
        }
        for (; b < n_blocks; b += STEP) v = bb::add(v, partials[b * 32 + col]);
        red[part][col] = v;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        uint32_t v = 0;
#pragma unroll
        for (int p = 0; p < TS_OPEN_THREADS / 32; p++) v = bb::add(v, red[p][threadIdx.x]);
        tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    DevChal ch;
    ch.begin(a.kind, a.ds, &ks, false);
    Ext opened[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int c = 0; c < 4; c++) opened[k].c[c] = tot[4 * k + c];
    {   // interpolate_coset's factor (z^n - s^n) / (n s^n) for z = zeta and zeta * g
        Ext z0 = a.ds->zeta, z1 = a.ds->zeta_next;
        for (uint32_t i = 0; i < log_n; i++) { z0 = bb::sqr(z0); z1 = bb::sqr(z1); }
        const Ext f0 = bb::scale(bb::sub(z0, bb::ext_from_base(sn)), denom);
        const Ext f1 = bb::scale(bb::sub(z1, bb::ext_from_base(sn)), denom);
#pragma unroll
        for (int k = 0; k < 8; k++) opened[k] = bb::mul(opened[k], (k == 2 || k == 3) ? f1 : f0);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) ch.observe_ext(opened[k]);
    const Ext al = ch.sample_ext();
    Ext alp[8];
    alp[0] = bb::ext_one();
#pragma unroll
    for (int k = 1; k < 8; k++) alp[k] = bb::mul(alp[k - 1], al);
    const Ext ry0 = bb::add(opened[0], bb::mul(alp[1], opened[1]));
    const Ext ry1 = bb::add(opened[2], bb::mul(alp[1], opened[3]));
    const Ext ry2 = bb::add(bb::add(opened[4], bb::mul(alp[1], opened[5])),
                            bb::add(bb::mul(alp[2], opened[6]), bb::mul(alp[3], opened[7])));
    if (threadIdx.x == 0) {
        for (int k = 0; k < 8; k++) {
            a.ds->alp[k] = alp[k];
            for (int c = 0; c < 4; c++) a.ps[a.lay.opened + 4 * k + c] = opened[k].c[c];
        }
        a.ds->y02 = bb::add(ry0, bb::mul(alp[4], ry2));
        a.ds->y1 = bb::mul(alp[2], ry1);
    }
    ch.end();
}
// FRI commit phase, round r: observe the layer's commitment, sample beta
__global__ void __launch_bounds__(64) ts_fri_round_kernel(TsArgs a, uint32_t round, uint32_t one_half) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ KState ks;
    DevChal ch;
    ch.begin(a.kind, a.ds, &ks, false);
    ch.observe_n(a.ps + a.lay.froots + 8 * round, 8);
    const Ext beta = ch.sample_ext();
    if (threadIdx.x == 0) a.ds->half_beta[round] = bb::scale(beta, one_half);
    ch.end();
}
// The TAIL of the FRI commit phase in ONE workgroup and one launch (Poseidon2 hashes): every round whose layer has at most 2^7 rows
// (2^8 extension elements) — commit the layer (16-lane cooperative sponge per row, cooperative tree levels), observe the root, sample
// beta, fold — with the vector and the shrinking digest layer in LDS; every layer, root, beta / 2 and folded vector also goes to HBM
// where the separate launches put them (openings, the final polynomial and the queries read them there).  Replaces, per round, a leaf
// launch, one or two tree launches, a transcript launch and a fold launch: about thirty launches of a 2^20 proof, all of a 2^7 one.
constexpr uint32_t FRI_TAIL_MAX_LOG = 8;  // elements of the first tail layer: 2^8 (128 rows of two)
struct FriTailArgs {
    TsArgs ts;
    uint32_t* vec;      // fri_vec
    uint32_t* layers;   // fri_layers
    uint32_t vec_off[FRI_TAIL_MAX_LOG + 1], layer_off[FRI_TAIL_MAX_LOG];  // of rounds r0 + k
    TwoLevelTable inv_roots[FRI_TAIL_MAX_LOG];                           // w_len^-e of round r0 + k
    uint32_t r0, n_tail, log_len0, one_half;
};
__global__ void __launch_bounds__(1024) fri_tail_kernel(FriTailArgs a) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ __attribute__((aligned(16))) uint32_t vbuf[2][4 << FRI_TAIL_MAX_LOG];  // the layer's vector, ping-pong across rounds
    __shared__ uint32_t dig[2][8 << (FRI_TAIL_MAX_LOG - 1)];                          // digest layer, ping-pong across levels
    __shared__ KState ks;
    __shared__ Ext hb;
    const uint32_t tid = threadIdx.x, lane16 = tid & 15u, grp = tid >> 4, wave_first = (tid >> 6) << 2;  // 64 lane-rows of 16
    const p2c::LaneConst lc = p2c::lane_constants(lane16);
    DevChal ch;
    if (tid < 64) ch.begin(HASH_POSEIDON2, a.ts.ds, &ks, false);
    uint32_t len = 1u << a.log_len0;
    for (uint32_t i = tid; i < 4 * len; i += blockDim.x) vbuf[0][i] = a.vec[a.vec_off[0] + i];
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t k = 0; k < a.n_tail; k++, len >>= 1) {
        const uint32_t half = len >> 1, log_half = a.log_len0 - 1 - k;
        uint32_t* lay = a.layers + a.layer_off[k];
        // leaf layer: row i = elements 2i, 2i + 1 = eight words = one sponge block
        for (uint32_t base = 0; base < half; base += 64) {
            const uint32_t row = base + grp;
            uint32_t res = 0;
            if (base + wave_first < half) {  // uniform over the wave: all sixteen lanes of an active row take part
                const uint32_t v = (row < half && lane16 < 8) ? vbuf[cur][row * 8 + lane16] : 0u;
                res = p2c::permute(v, lc);
            }
            if (row < half && lane16 < 8) { dig[0][row * 8 + lane16] = res; lay[row * 8 + lane16] = res; }
        }
        __syncthreads();
        // tree levels
        uint32_t d = 0;
        size_t off = (size_t)half * 8;
        for (uint32_t n = half; n > 1; n >>= 1) {
            const uint32_t m = n >> 1;  // <= 64 compressions
            uint32_t res = 0;
            if (wave_first < m) {
                const uint32_t v = grp < m ? dig[d][grp * 16 + lane16] : 0u;
                res = p2c::permute(v, lc);
            }
            if (grp < m && lane16 < 8) { dig[d ^ 1][grp * 8 + lane16] = res; lay[off + grp * 8 + lane16] = res; }
            __syncthreads();
            d ^= 1;
            off += (size_t)m * 8;
        }
        // root -> staging buffer; transcript on the first wave: observe the root, sample beta
        const uint32_t round = a.r0 + k;
        if (tid < 8) a.ts.ps[a.ts.lay.froots + 8 * round + tid] = dig[d][tid];
        if (tid < 64) {
            for (uint32_t i = 0; i < 8; i++) ch.observe(dig[d][i]);
            const Ext beta = ch.sample_ext();
            if (tid == 0) { hb = bb::scale(beta, a.one_half); a.ts.ds->half_beta[round] = hb; }
        }
        __syncthreads();
        // fold (TwoAdicFriFolding::fold_matrix, as fri_fold_kernel)
        if (tid < half) {
            const Ext half_beta = hb;
            const Ext lo = ld_ext(&vbuf[cur][8 * tid]), hi = ld_ext(&vbuf[cur][8 * tid + 4]);
            const uint32_t p = tl(a.inv_roots[k], brev(tid, log_half));
            const Ext r = bb::add(bb::scale(bb::add(lo, hi), a.one_half), bb::mul(bb::scale(half_beta, p), bb::sub(lo, hi)));
            st_ext(&vbuf[cur ^ 1][4 * tid], r);
            st_ext(a.vec + a.vec_off[k + 1] + 4 * (size_t)tid, r);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (tid < 64) ch.end();
}

// observe the final polynomial; set up the proof-of-work search
__global__ void __launch_bounds__(64) ts_final_kernel(TsArgs a, uint32_t fpl, uint32_t pow_mask) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ KState ks;
    DevChal ch;
    ch.begin(a.kind, a.ds, &ks, false);
    for (uint32_t i = 0; i < 4 * fpl; i++) ch.observe(a.ps[a.lay.fpoly + i]);
    ch.end();
    if (threadIdx.x == 0) {
        a.ds->grind_result = 0xffffffffu;
        if (a.kind == HASH_KECCAK) {
            // the sponge holds the complete blocks; the template carries the pending tail, room for the witness, padding
            KeccakGrindArgs& g = a.ds->kgrind;
            for (int i = 0; i < 25; i++) g.state[i] = ks.st[i];
            uint8_t tmpl[272];
            for (int i = 0; i < 272; i++) tmpl[i] = 0;
            const uint32_t tail = ks.blen;
            for (uint32_t i = 0; i < tail; i++) tmpl[i] = ks.blk[i];
            const uint32_t end = tail + 4;  // message end within the template
            g.n_blocks = end < 136 ? 1 : 2;
            tmpl[end] ^= 0x01;
            tmpl[g.n_blocks * 136 - 1] ^= 0x80;
            for (int b = 0; b < 2; b++)
                for (int i = 0; i < 17; i++) {
                    uint64_t w = 0;
                    for (int j = 0; j < 8; j++) w |= (uint64_t)tmpl[b * 136 + 8 * i + j] << (8 * j);
                    g.block[b][i] = w;
                }
            g.wpos = tail;
            g.mask = pow_mask;
            g.base = 0;
        }
    }
}
// check the witness, observe it, sample the query indices.  If the search range held no witness the transcript is
// left untouched and the host continues the search (status = ST_GRIND_MISS).
__global__ void __launch_bounds__(64) ts_queries_kernel(TsArgs a, uint32_t nq, uint32_t log_big, uint32_t pow_bits, uint32_t* qidx) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ KState ks;
    const uint32_t found = a.ds->grind_result;
    if (found == 0xffffffffu) {
        if (threadIdx.x == 0) { a.ds->status = ST_GRIND_MISS; a.ps[a.lay.status] = ST_GRIND_MISS; }
        for (uint32_t q = threadIdx.x; q < nq; q += 64) qidx[q] = 0;  // the gather that follows reads defined indices
        return;
    }
    DevChal ch;
    ch.begin(a.kind, a.ds, &ks, false);
    const uint32_t witness = bb::to_monty(found);
    ch.observe(witness);
    const uint32_t zero_bits = ch.sample_bits(pow_bits);
    if (threadIdx.x == 0) {
        a.ps[a.lay.witness] = witness;
        a.ps[a.lay.status] = zero_bits == 0 ? 0u : 2u;  // 2: the transcript rejects the witness the search returned
        a.ds->status = 0;
    }
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t idx = ch.sample_bits(log_big);
        if (threadIdx.x == 0) { qidx[q] = idx; a.ps[a.lay.qidx + q] = idx; }
    }
    ch.end();
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
static void put_u32(std::vector<uint8_t>& b, uint32_t v) {
    size_t o = b.size();
    b.resize(o + 4);
    memcpy(b.data() + o, &v, 4);
}
static void put_words(std::vector<uint8_t>& b, const uint32_t* w, size_t n) {
    size_t o = b.size();
    b.resize(o + 4 * n);
    memcpy(b.data() + o, w, 4 * n);
}

constexpr int N_STAGE_EVENTS = 7;

struct FibProver::Impl {
    int hash = HASH_POSEIDON2;
    int profile = PROFILE_LATENCY;
    int device = -1;  // the arena's device: prove() refuses to run with another one current
    uint32_t log_n = 0, log_big = 0;
    FriParams fp{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // arena
    uint32_t *trace = nullptr, *lde_t = nullptr, *qflat = nullptr, *lde_q = nullptr, *d0 = nullptr, *d1 = nullptr;
    uint32_t *layers_t = nullptr, *layers_q = nullptr, *fri_vec = nullptr, *fri_layers = nullptr;
    uint32_t *partials = nullptr, *qidx = nullptr, *pstage = nullptr, *fp_ev = nullptr;
    DevState* ds = nullptr;
    QTree* qtrees = nullptr;
    // pinned: the proof staging buffer lands here, one copy per proof.  Two slots (and two sets of stage events): a second
    // proof may be enqueued behind the first on the same stream before the first is collected (enqueue / finish)
    uint32_t* host_stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    struct Pending { uint64_t a, b; int slot; std::chrono::steady_clock::time_point t_start, t_enq; uint64_t seq; };
    std::deque<Pending> pending;
    uint64_t next_seq = 1, arena_owner = 0;  // arena_owner: the proof whose launch sequence ran last (its state is in the arena)
    StageLayout lay;
    hipEvent_t ev[2][N_STAGE_EVENTS] = {{nullptr}, {nullptr}};
    std::vector<void*> allocs;
    uint32_t n_rounds = 0;
    std::vector<size_t> fri_vec_off, fri_layer_off;  // word offsets per round
    size_t slot_words = 0;
    uint32_t bary_blocks = 0;
    StageTimes times{};
    // diagnostics of the proof-of-work continuation path (p3hip_fib_prover_grind_miss_probe): how often the first search
    // range was empty, and the query-index buffer as the device held it when the host learnt of the last miss
    uint64_t grind_misses = 0;
    std::vector<uint32_t> miss_qidx;
    ~Impl() {
        for (void* p : allocs) (void)hipFree(p);
        for (auto* h : host_stage) if (h) (void)hipHostFree(h);
        for (auto& set : ev) for (auto& e : set) if (e) (void)hipEventDestroy(e);
        for (auto& e : done) if (e) (void)hipEventDestroy(e);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
    int alloc(uint32_t** p, size_t words) {
        P3_HIP(hipMalloc(reinterpret_cast<void**>(p), words * 4 + 64));
        allocs.push_back(*p);
        return OK;
    }
};

FibProver::FibProver() : im(new Impl()) {}
FibProver::~FibProver() { delete im; }

int FibProver::init(uint32_t log_n, const FriParams& fp, hipStream_t stream, bool own_stream, int hash, int profile) {
    Impl& s = *im;
    if (profile != PROFILE_THROUGHPUT && profile != PROFILE_LATENCY) return fail(ERR_BAD_ARG, "fib prover: unknown profile");
    s.profile = profile;
    s.stream = stream; s.own_stream = own_stream;  // first: an owned stream is destroyed with the prover even when init fails
    if (hash != HASH_POSEIDON2 && hash != HASH_KECCAK) return fail(ERR_BAD_ARG, "fib prover: unknown hash configuration");
    im->hash = hash;
    P3_HIP(hipGetDevice(&s.device));
    if (log_n < 1) return fail(ERR_BAD_ARG, "fib prover: log_n must be >= 1");
    if (log_n + fp.log_blowup > MAX_LOG_DOMAIN)
        return fail(ERR_BAD_ARG, "fib prover: LDE domain above 2^" + std::to_string(MAX_LOG_DOMAIN) + " points (log_n + log_blowup)");
    if (fp.log_blowup < 1) return fail(ERR_BAD_ARG, "fib prover: log_blowup must be >= 1");
    // p3_fri::prover::prove: `if log_final_poly_len > 0 { assert!(log_min_height > log_final_poly_len + log_blowup) }`
    if (fp.log_final_poly_len > log_n || (fp.log_final_poly_len > 0 && fp.log_final_poly_len >= log_n))
        return fail(ERR_BAD_ARG, "fib prover: log_final_poly_len must be below the trace's log height");
    if (fp.proof_of_work_bits > 30) return fail(ERR_BAD_ARG, "fib prover: proof_of_work_bits too large");
    s.log_n = log_n; s.fp = fp; s.log_big = log_n + fp.log_blowup;
    const size_t n = (size_t)1 << log_n, big = (size_t)1 << s.log_big;
    int rc;
    if ((rc = s.alloc(&s.trace, n * 2))) return rc;
    if ((rc = s.alloc(&s.lde_t, big * 2))) return rc;
    if ((rc = s.alloc(&s.qflat, n * 4))) return rc;
    if ((rc = s.alloc(&s.lde_q, big * 4))) return rc;
    if ((rc = s.alloc(&s.d0, big * 4))) return rc;
    if ((rc = s.alloc(&s.d1, big * 4))) return rc;
    if ((rc = s.alloc(&s.layers_t, mmcs_layer_words(big)))) return rc;
    if ((rc = s.alloc(&s.layers_q, mmcs_layer_words(big)))) return rc;
    // FRI: vector r has big >> r ext elements; its commitment tree has (big >> (r+1)) leaves
    s.n_rounds = s.log_big - fp.log_blowup - fp.log_final_poly_len;
    if (s.n_rounds > MAX_FRI_ROUNDS) return fail(ERR_BAD_ARG, "fib prover: too many FRI rounds");
    size_t vec_words = 0, layer_words = 0;
    for (uint32_t r = 0; r <= s.n_rounds; r++) { s.fri_vec_off.push_back(vec_words); vec_words += (big >> r) * 4; }
    for (uint32_t r = 0; r < s.n_rounds; r++) { s.fri_layer_off.push_back(layer_words); layer_words += mmcs_layer_words(big >> (r + 1)); }
    if ((rc = s.alloc(&s.fri_vec, vec_words))) return rc;
    if ((rc = s.alloc(&s.fri_layers, layer_words + 8))) return rc;
    // 256 workgroups: each lane then folds 16 rows at 2^20 before the 32 wave reductions (which cost as much as ~3 rows),
    // and the transcript kernel that finishes the sums reads 256 x 32 partial words instead of 1024 x 32
    s.bary_blocks = (uint32_t)std::min<size_t>(256, (n + BARY_BLOCK - 1) / BARY_BLOCK);
    if ((rc = s.alloc(&s.partials, (size_t)s.bary_blocks * 32))) return rc;
    const size_t fpl = (size_t)1 << fp.log_final_poly_len;
    if ((rc = s.alloc(&s.fp_ev, fpl * 4))) return rc;
    {
        uint32_t* p = nullptr;
        if ((rc = s.alloc(&p, (sizeof(DevState) + 3) / 4))) return rc;
        s.ds = reinterpret_cast<DevState*>(p);
        P3_HIP(hipMemset(s.ds, 0, sizeof(DevState)));
    }
    // query staging: per query one slot holding every tree's (row, path)
    size_t slot = (2 + s.log_big * 8) + (4 + s.log_big * 8);
    for (uint32_t r = 0; r < s.n_rounds; r++) slot += 8 + (size_t)(s.log_big - 1 - r) * 8;
    s.slot_words = slot;
    const uint32_t nq = fp.num_queries;
    StageLayout& L = s.lay;
    L.fpoly = L.froots + 8 * s.n_rounds;
    L.witness = L.fpoly + 4 * (uint32_t)fpl;
    L.status = L.witness + 1;
    L.qidx = L.status + 1;
    L.slots = (L.qidx + nq + 3u) & ~3u;
    const size_t stage_words = (size_t)L.slots + slot * nq;
    if (stage_words > 0xffffffffull) return fail(ERR_BAD_ARG, "fib prover: proof staging buffer too large");
    L.words = (uint32_t)stage_words;
    if ((rc = s.alloc(&s.pstage, stage_words))) return rc;
    if ((rc = s.alloc(&s.qidx, std::max<uint32_t>(nq, 1)))) return rc;
    uint32_t* qt = nullptr;
    if ((rc = s.alloc(&qt, (sizeof(QTree) / 4) * (s.n_rounds + 2)))) return rc;
    s.qtrees = reinterpret_cast<QTree*>(qt);
    for (auto& h : s.host_stage) P3_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), stage_words * 4 + 64));
    for (auto& set : s.ev) for (auto& e : set) P3_HIP(hipEventCreate(&e));
    for (auto& e : s.done) P3_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // device descriptors of the trees opened per query (fixed for the prover's lifetime)
    std::vector<QTree> qd;
    uint32_t off = 0;
    qd.push_back(QTree{s.lde_t, s.layers_t, 2, s.log_big, 0, off}); off += 2 + s.log_big * 8;
    qd.push_back(QTree{s.lde_q, s.layers_q, 4, s.log_big, 0, off}); off += 4 + s.log_big * 8;
    for (uint32_t r = 0; r < s.n_rounds; r++) {
        uint32_t lh = s.log_big - 1 - r;
        qd.push_back(QTree{s.fri_vec + s.fri_vec_off[r], s.fri_layers + s.fri_layer_off[r], 8, lh, r + 1, off});
        off += 8 + lh * 8;
    }
    P3_HIP(hipMemcpy(s.qtrees, qd.data(), qd.size() * sizeof(QTree), hipMemcpyHostToDevice));
    return OK;
}

// Cached per context: selector table for log_n (built by a kernel on `stream`; other streams wait for its event).
static int get_selectors(Context& cx, hipStream_t stream, uint32_t log_n, const uint2** out) {
    auto it = cx.selector_tables.find(log_n);
    if (it == cx.selector_tables.end()) {
        const uint32_t n = 1u << log_n;
        CachedTable e;
        P3_HIP(hipMalloc(reinterpret_cast<void**>(&e.t.lo), (size_t)n * 8));
        TwoLevelTable roots;
        int rc = cx.get_root_table(stream, log_n, false, &roots);
        if (rc) { (void)hipFree(e.t.lo); return rc; }
        uint32_t gen = bb::to_monty(bb::GEN);
        uint32_t g = bb::two_adic_generator(log_n), ginv = bb::inv(g);
        uint32_t zh = bb::sub(bb::pow(gen, n), bb::ONE);
        uint32_t threads = (n + SEL_CHUNK - 1) / SEL_CHUNK;
        hipLaunchKernelGGL(selectors_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, roots, n, gen, ginv, zh,
                           reinterpret_cast<uint2*>(e.t.lo));
        P3_HIP(hipGetLastError());
        if ((rc = cx.mark_built(stream, e))) return rc;
        it = cx.selector_tables.emplace(log_n, e).first;
    }
    int rc = cx.wait_ready(stream, it->second);
    if (rc) return rc;
    *out = reinterpret_cast<const uint2*>(it->second.t.lo);
    return OK;
}

// Everything is ENQUEUED: the transcript runs on the device between the bulk kernels, so the host synchronises once
// per proof, when the staging buffer (commitments, opened values, final polynomial, witness, query openings) has
// landed in pinned memory.  The only other synchronisations are the rare continuation of a proof-of-work search whose
// first range (16x the expected number of candidates) held no witness.
int FibProver::prove(uint64_t a, uint64_t b, std::vector<uint8_t>* proof) {
    if (!im->pending.empty()) return fail(ERR_BAD_ARG, "fib prover: finish the enqueued proofs before a synchronous prove");
    int rc = enqueue(a, b);
    if (rc) return rc;
    return finish(proof);
}
// enqueue: everything up to the copy of the staging buffer into pinned memory, plus an event behind it; returns at once.
// At most two proofs may be in flight (the second one's kernels simply queue behind the first on the prover's stream: the
// arena is reused in stream order, the two results land in different pinned buffers).  finish: waits for the OLDEST one,
// continues an empty proof-of-work search if need be, serialises.
int FibProver::enqueue(uint64_t a, uint64_t b) {
    Impl& s = *im;
    if (s.pending.size() >= 2) return fail(ERR_BAD_ARG, "fib prover: two proofs already in flight (finish one first)");
    const int slot = s.pending.empty() ? 0 : 1 - s.pending.back().slot;
    Impl::Pending p{a, b, slot, std::chrono::steady_clock::now(), {}, s.next_seq++};
    int rc = run(a, b, slot, 0, nullptr, &p);
    if (rc) return rc;
    s.pending.push_back(p);
    return OK;
}
int FibProver::finish(std::vector<uint8_t>* proof) {
    Impl& s = *im;
    if (s.pending.empty()) return fail(ERR_BAD_ARG, "fib prover: no proof in flight");
    Impl::Pending p = s.pending.front();
    int rc = run(p.a, p.b, p.slot, 1, proof, &p);
    s.pending.pop_front();
    return rc;
}
// phase 0: enqueue only; phase 1: collect only
int FibProver::run(uint64_t a, uint64_t b, int slot, int phase, std::vector<uint8_t>* proof, void* pending_rec) {
    Impl& s = *im;
    Impl::Pending& pend = *static_cast<Impl::Pending*>(pending_rec);
    hipEvent_t* const ev = s.ev[slot];
    Context* cxp;
    int rc = get_context(&cxp);
    if (rc) return rc;
    Context& cx = *cxp;
    if (cx.device != s.device)
        return fail(ERR_BAD_ARG, "fib prover: created on device " + std::to_string(s.device) + ", current device is " + std::to_string(cx.device));
    hipStream_t st = s.stream;
    const uint32_t log_n = s.log_n, log_big = s.log_big;
    const uint32_t n = 1u << log_n, big = 1u << log_big;
    const uint32_t gen = bb::to_monty(bb::GEN);
    const uint32_t g_n = bb::two_adic_generator(log_n), g_n_inv = bb::inv(g_n);
    const StageLayout& L = s.lay;
    const TsArgs ts{s.ds, s.pstage, s.hash, L};
    auto commit = [&](const uint32_t* mat, size_t h, size_t w, uint32_t* layers, uint32_t root_slot) -> int {
        const uint32_t* mp[1] = {mat};
        size_t hh[1] = {h}, ww[1] = {w};
        Tree* tp = nullptr;
        int r = mmcs_commit(st, mp, hh, ww, 1, &tp, layers, s.pstage + root_slot, s.hash, nullptr, s.profile);
        if (r) return r;
        std::unique_ptr<Tree> t(tp);  // the layers live in the arena; the descriptor is not needed again
        if (!t->root_copied)
            P3_HIP(hipMemcpyAsync(s.pstage + root_slot, t->layers + t->layer_off.back(), 32, hipMemcpyDeviceToDevice, st));
        return OK;
    };
    const uint32_t one_half = bb::inv(bb::to_monty(2));
    const uint32_t fpl = 1u << s.fp.log_final_poly_len;
    const uint32_t pow_mask = (1u << s.fp.proof_of_work_bits) - 1u;
    const uint32_t nq = s.fp.num_queries;
    auto grind = [&](uint64_t base, uint32_t count) -> int {
        if (s.hash == HASH_KECCAK) hipLaunchKernelGGL(grind_keccak_kernel, dim3(count / 256), dim3(256), 0, st, s.ds, pow_mask, (uint32_t)base);
        else hipLaunchKernelGGL(grind_kernel, dim3(count / 256), dim3(256), 0, st, s.ds, pow_mask, (uint32_t)base);
        P3_HIP(hipGetLastError());
        return OK;
    };
    // P3HIP_GRIND_FIRST_LOG (tests): log2 of the whole first search range, to exercise the continuation path
    const uint32_t first_log = [] { const char* e = getenv("P3HIP_GRIND_FIRST_LOG"); return e ? (uint32_t)atoi(e) : 0u; }();
    uint32_t batch = 1u << std::min<uint32_t>(std::max<uint32_t>(first_log ? first_log : s.fp.proof_of_work_bits + 4, 8), 24);
    auto queries = [&]() -> int {
        hipLaunchKernelGGL(ts_queries_kernel, dim3(1), dim3(64), 0, st, ts, nq, log_big, s.fp.proof_of_work_bits, s.qidx);
        P3_HIP(hipGetLastError());
        if (nq) {
            hipLaunchKernelGGL(query_gather_kernel, dim3(nq, s.n_rounds + 2), dim3(64), 0, st, s.qtrees, s.n_rounds + 2, s.qidx,
                               (uint32_t)s.slot_words, s.pstage + L.slots);
            P3_HIP(hipGetLastError());
        }
        P3_HIP(hipMemcpyAsync(s.host_stage[slot], s.pstage, (size_t)L.words * 4, hipMemcpyDeviceToHost, st));
        return OK;
    };
    // the whole launch sequence of one proof (a lambda: finish() runs it again when a proof-of-work search that came up
    // empty has to be continued after a newer proof has already reused the arena)
    auto body = [&]() -> int {
        s.arena_owner = pend.seq;
        P3_HIP(hipEventRecord(ev[0], st));

        // ---- trace + commit (pcs.commit: bit-reversed coset LDE, shift GENERATOR) ----
        if ((rc = fib_trace(st, a, b, n, s.trace))) return rc;
        if ((rc = ntt_coset_lde(cx, st, s.trace, s.lde_t, n, 2, s.fp.log_blowup, gen, true))) return rc;
        if ((rc = commit(s.lde_t, big, 2, s.layers_t, L.root_t))) return rc;
        hipLaunchKernelGGL(ts_begin_kernel, dim3(1), dim3(64), 0, st, ts, s.trace, n, log_n);
        P3_HIP(hipGetLastError());
        P3_HIP(hipEventRecord(ev[1], st));

        // ---- quotient values + commit (shift GENERATOR/GENERATOR = 1) ----
        const uint2* sel = nullptr;
        if ((rc = get_selectors(cx, st, log_n, &sel))) return rc;
        {
            QuotArgs qa{};
            qa.lde = reinterpret_cast<const uint2*>(s.lde_t);
            qa.sel = sel;
            qa.out = s.qflat;
            qa.ds = s.ds;
            if ((rc = cx.get_root_table(st, log_n, false, &qa.roots))) return rc;
            qa.n = n; qa.log_n = log_n; qa.gen = gen; qa.ginv = g_n_inv;
            qa.zh_inv = bb::inv(bb::sub(bb::pow(gen, n), bb::ONE));
            hipLaunchKernelGGL(fib_quotient_kernel, dim3((n + 255) / 256), dim3(256), 0, st, qa);
            P3_HIP(hipGetLastError());
        }
        if ((rc = ntt_coset_lde(cx, st, s.qflat, s.lde_q, n, 4, s.fp.log_blowup, bb::ONE, true))) return rc;
        if ((rc = commit(s.lde_q, big, 4, s.layers_q, L.root_q))) return rc;
        hipLaunchKernelGGL(ts_zeta_kernel, dim3(1), dim3(64), 0, st, ts, g_n);
        P3_HIP(hipGetLastError());
        P3_HIP(hipEventRecord(ev[2], st));

        // ---- pcs.open: opened values ----
        TwoLevelTable roots_big;
        if ((rc = cx.get_root_table(st, log_big, false, &roots_big))) return rc;
        {
            uint32_t threads = (big + DEN_CHUNK - 1) / DEN_CHUNK;
            hipLaunchKernelGGL(inv_denoms_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, roots_big, big, log_big, gen,
                               s.ds, s.d0, s.d1);
            P3_HIP(hipGetLastError());
            hipLaunchKernelGGL(barycentric_kernel, dim3(s.bary_blocks), dim3(BARY_BLOCK), 0, st, roots_big, n, log_big, gen,
                               reinterpret_cast<const uint2*>(s.lde_t), reinterpret_cast<const uint4*>(s.lde_q), s.d0, s.d1,
                               s.partials);
            P3_HIP(hipGetLastError());
            const uint32_t sn = bb::pow(gen, n);
            const uint32_t denom = bb::inv(bb::mul(bb::to_monty(n), sn));
            hipLaunchKernelGGL(ts_open_kernel, dim3(1), dim3(TS_OPEN_THREADS), 0, st, ts, s.partials, s.bary_blocks, log_n, sn, denom);
            P3_HIP(hipGetLastError());
        }
        P3_HIP(hipEventRecord(ev[3], st));

        // ---- reduced openings -> FRI input ----
        {
            ReducedArgs ra{};
            ra.lde_t = reinterpret_cast<const uint2*>(s.lde_t);
            ra.lde_q = reinterpret_cast<const uint4*>(s.lde_q);
            ra.d0 = s.d0; ra.d1 = s.d1; ra.ro = s.fri_vec + s.fri_vec_off[0]; ra.big = big;
            ra.ds = s.ds;
            hipLaunchKernelGGL(reduced_openings_kernel, dim3((big + 255) / 256), dim3(256), 0, st, ra);
            P3_HIP(hipGetLastError());
        }

        // ---- FRI commit phase ----
        // LATENCY profile: rounds whose layer has at most 2^7 rows in ONE launch (fri_tail_kernel; Poseidon2 hashes).  Worth little:
        // a 2^20 proof 3.51 against 3.52 ms (the rounds are a chain of permutation
        // latencies either way, the ~30 launches they save cost little on a stream that is waiting anyway), a 2^10 proof 0.705 ->
        // 0.678 ms, four provers 584 -> 580 proofs/s (profiles/r04_latency_ab.txt)
        uint32_t r_tail = s.n_rounds;
        if (s.profile == PROFILE_LATENCY && s.hash == HASH_POSEIDON2)
            while (r_tail > 0 && (big >> (r_tail - 1)) <= (1u << FRI_TAIL_MAX_LOG)) r_tail--;
        for (uint32_t r = 0; r < r_tail; r++) {
            uint32_t len = big >> r, half = len >> 1;
            // ExtensionMmcs: rows of two ext elements, flattened
            if ((rc = commit(s.fri_vec + s.fri_vec_off[r], half, 8, s.fri_layers + s.fri_layer_off[r], L.froots + 8 * r))) return rc;
            hipLaunchKernelGGL(ts_fri_round_kernel, dim3(1), dim3(64), 0, st, ts, r, one_half);
            P3_HIP(hipGetLastError());
            TwoLevelTable inv_roots;
            uint32_t log_half = log_big - 1 - r;
            if ((rc = cx.get_root_table(st, log_half + 1, true, &inv_roots))) return rc;
            hipLaunchKernelGGL(fri_fold_kernel, dim3((half + 255) / 256), dim3(256), 0, st, inv_roots,
                               s.fri_vec + s.fri_vec_off[r], s.fri_vec + s.fri_vec_off[r + 1], half, log_half, s.ds, r, one_half);
            P3_HIP(hipGetLastError());
        }
        if (r_tail < s.n_rounds) {
            FriTailArgs ta{};
            ta.ts = ts; ta.vec = s.fri_vec; ta.layers = s.fri_layers;
            ta.r0 = r_tail; ta.n_tail = s.n_rounds - r_tail; ta.log_len0 = log_big - r_tail; ta.one_half = one_half;
            for (uint32_t k = 0; k <= ta.n_tail; k++) {
                if (s.fri_vec_off[r_tail + k] > 0xffffffffull) return fail(ERR_INTERNAL, "fri tail: vector offset out of range");
                ta.vec_off[k] = (uint32_t)s.fri_vec_off[r_tail + k];
            }
            for (uint32_t k = 0; k < ta.n_tail; k++) {
                if (s.fri_layer_off[r_tail + k] > 0xffffffffull) return fail(ERR_INTERNAL, "fri tail: layer offset out of range");
                ta.layer_off[k] = (uint32_t)s.fri_layer_off[r_tail + k];
                if ((rc = cx.get_root_table(st, ta.log_len0 - k, true, &ta.inv_roots[k]))) return rc;
            }
            hipLaunchKernelGGL(fri_tail_kernel, dim3(1), dim3(1024), 0, st, ta);
            P3_HIP(hipGetLastError());
        }
        // final polynomial: first 2^lfp entries (bit-reversed order) -> natural order -> inverse DFT (of the four base
        // coordinates: the transform is linear over the base field) straight into the staging buffer
        if ((rc = bit_reverse_rows(st, s.fri_vec + s.fri_vec_off[s.n_rounds], s.fp_ev, fpl, 4))) return rc;
        if ((rc = ntt_dft(cx, st, s.fp_ev, s.pstage + L.fpoly, fpl, 4, true))) return rc;
        hipLaunchKernelGGL(ts_final_kernel, dim3(1), dim3(64), 0, st, ts, fpl, pow_mask);
        P3_HIP(hipGetLastError());
        P3_HIP(hipEventRecord(ev[4], st));

        // ---- proof of work: two launches, no synchronisation.  The first covers 2x the expected number of candidates
        // (every block of a launch is resident before the first one finishes, so a wider first launch would simply do
        // all of its work); the second covers up to 16x and its blocks return at once when the first found a witness
        // (P[first misses] = e^-2, P[both miss] = e^-16: then the host continues the search after the proof's sync).
        {
            const uint32_t head = first_log ? batch : std::min<uint32_t>(batch, 1u << std::max<uint32_t>(s.fp.proof_of_work_bits + 1, 8));
            if ((rc = grind(0, head))) return rc;
            if (batch > head && (rc = grind(head, batch - head))) return rc;
        }
        P3_HIP(hipEventRecord(ev[5], st));

        // ---- query phase ----
        if ((rc = queries())) return rc;
        P3_HIP(hipEventRecord(ev[6], st));
        return OK;
    };
    if (phase == 0) {
        if ((rc = body())) return rc;
        P3_HIP(hipEventRecord(s.done[slot], st));
        pend.t_enq = std::chrono::steady_clock::now();
        return OK;
    }
    const auto t_start = pend.t_start, t_enq = pend.t_enq;
    P3_HIP(hipEventSynchronize(s.done[slot]));  // the one synchronisation of a proof
    const uint32_t* hp = s.host_stage[slot];
    static const bool trace = getenv("P3HIP_TRACE") != nullptr;
#define TR(...) do { if (trace) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)
    TR("prove: enqueued in %.0f us of host time, synced after another %.0f us, status %u\n",
       std::chrono::duration<double, std::micro>(t_enq - t_start).count(),
       std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq).count(), hp[L.status]);
    if (hp[L.status] == ST_GRIND_MISS) {
        if (s.arena_owner != pend.seq) {
            // a newer proof was queued behind this one and has reused the arena: let the stream drain (that proof's result
            // waits in the other pinned buffer), then run this proof again from the start — it reaches the same empty range
            P3_HIP(hipStreamSynchronize(st));
            if ((rc = body())) return rc;
            P3_HIP(hipStreamSynchronize(st));
        }
        // What the query phase left behind on the miss: ts_queries_kernel zeroes the index buffer before it returns, because
        // query_gather_kernel is already queued behind it and reads that buffer (the round-2 fault: DESIGN.md section 5)
        s.grind_misses++;
        s.miss_qidx.assign(nq, 0xdeadbeefu);
        if (nq) P3_HIP(hipMemcpyAsync(s.miss_qidx.data(), s.qidx, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        P3_HIP(hipStreamSynchronize(st));
        // continue the search range by range (each 4x the previous one), then redo the query phase
        uint32_t found = 0xffffffffu;
        for (uint64_t base = batch; base < bb::P && found == 0xffffffffu; base += batch) {
            batch = std::min<uint32_t>(batch * 4, 1u << 24);
            TR("prove: continue search base %llu batch %u\n", (unsigned long long)base, batch);
            if ((rc = grind(base, batch))) return rc;
            P3_HIP(hipMemcpyAsync(&found, &s.ds->grind_result, 4, hipMemcpyDeviceToHost, st));
            P3_HIP(hipStreamSynchronize(st));
            TR("prove: found %u\n", found);
        }
        if (found == 0xffffffffu) return fail(ERR_INTERNAL, "grind: no proof-of-work witness found");
        if ((rc = queries())) return rc;
        P3_HIP(hipStreamSynchronize(st));
        TR("prove: queries redone, status %u\n", hp[L.status]);
    }
    if (hp[L.status] != 0) return fail(ERR_INTERNAL, "grind: witness rejected by the device transcript");

    // ---- serialise (same layout as the test oracle's restatement) ----
    std::vector<uint8_t>& pf = *proof;
    pf.clear();
    pf.reserve(64 + (size_t)nq * s.slot_words * 4 + 4096);
    put_u32(pf, 0x42463350u); put_u32(pf, 1); put_u32(pf, log_n);
    put_words(pf, hp + L.root_t, 8); put_words(pf, hp + L.root_q, 8);
    const uint32_t* op = hp + L.opened;
    put_u32(pf, 2); put_words(pf, op, 8);
    put_u32(pf, 2); put_words(pf, op + 8, 8);
    put_u32(pf, 1); put_u32(pf, 4); put_words(pf, op + 16, 16);
    put_u32(pf, s.n_rounds); put_words(pf, hp + L.froots, (size_t)s.n_rounds * 8);
    put_u32(pf, nq);
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t* slot = hp + L.slots + (size_t)q * s.slot_words;
        put_u32(pf, 2);
        put_u32(pf, 1); put_u32(pf, 2); put_words(pf, slot, 2); put_u32(pf, log_big); put_words(pf, slot + 2, (size_t)log_big * 8);
        slot += 2 + log_big * 8;
        put_u32(pf, 1); put_u32(pf, 4); put_words(pf, slot, 4); put_u32(pf, log_big); put_words(pf, slot + 4, (size_t)log_big * 8);
        slot += 4 + log_big * 8;
        put_u32(pf, s.n_rounds);
        for (uint32_t r = 0; r < s.n_rounds; r++) {
            uint32_t lh = log_big - 1 - r;
            uint32_t idx = hp[L.qidx + q] >> r;
            put_words(pf, slot + 4 * ((idx ^ 1) & 1), 4);  // sibling_value
            put_u32(pf, lh); put_words(pf, slot + 8, (size_t)lh * 8);
            slot += 8 + lh * 8;
        }
    }
    put_u32(pf, fpl);
    put_words(pf, hp + L.fpoly, (size_t)fpl * 4);
    put_u32(pf, hp[L.witness]);
    TR("prove: serialised %zu bytes\n", pf.size());
    // stage times on the device timeline (events between the stages of the stream)
    float ms[N_STAGE_EVENTS - 1] = {0};
    for (int k = 0; k + 1 < N_STAGE_EVENTS; k++) (void)hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]);
    s.times.trace_commit_ms += ms[0]; s.times.quotient_commit_ms += ms[1]; s.times.open_ms += ms[2];
    s.times.fri_commit_ms += ms[3]; s.times.grind_ms += ms[4]; s.times.query_ms += ms[5]; s.times.proofs += 1;
    return OK;
}

const StageTimes& FibProver::times() const { return im->times; }
void FibProver::reset_times() { im->times = StageTimes{}; }
uint64_t FibProver::grind_misses(std::vector<uint32_t>* last_indices) const {
    if (last_indices) *last_indices = im->miss_qidx;
    return im->grind_misses;
}

}  // namespace p3

#include "prover_hiding.hip.inc"
