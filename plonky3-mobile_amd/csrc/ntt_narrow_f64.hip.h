// The three-launch narrow coset LDE of ntt_narrow.hip.h with its butterflies in DOUBLE PRECISION integer arithmetic.
//
// Why it was built: the integer kernels spent 12 VALU instructions per butterfly (add/sub/min, sub/add, and the then seven-op Montgomery
// product; ten with the five-instruction product of round 4, bb31.hip.h).  BabyBear has one spare bit in a 32-bit word (4P > 2^32), so neither Harvey's lazy butterflies nor a signed lazy
// representation fit; in fp64 they do: a lane keeps integers |v| < 2^44 congruent to the Montgomery WORDS (the transform is linear, so
// the words themselves are transformed, with CANONICAL twiddles), a modular add / sub is ONE v_add_f64 with no reduction for a whole
// 12-stage digit, and the product is the four-op magic-number form of poseidon2_f64.hip.h:
//     (x, y) -> (x + y, (x - y) * w)        6 instructions instead of 12.
// Conversions: one v_cvt_f64_u32 per loaded word; 3 instructions per stored word when the value comes out of a product
// (|r| <= P/2 + 1: convert, add P, min), 4 when it does not (floor-quotient reduction, below).  Between the register rounds the tile
// is exchanged through LDS as doubles (one 8-byte plane per column of the lane's vector, the same padded layout and bank analysis
// as the uint2 tiles of the integer kernels); the last hand-over of K1 / K3 is on words.  (A form with EVERY hand-over on words — five
// more instructions per element and hand-over, the integer kernels' LDS footprint — lost by more at every size and was retired in
// round 5: profiles/r04_lde_f64_vs_int.txt, 1461 against 1259 against 837 us at cfg3's shape.)
//
// What it bought (profiles/r03_pmc_lde_valu.json, r03_lde_f64_vs_int.txt): a third fewer VALU instructions
// per wave and 2-8 % of the unit's time up to 2^20 rows — the unit is a chain of unoverlapped phases, and the hand-overs of doubles
// cost what the rounds save.  From 2^21 rows on these kernels LOSE: 64 VGPRs for a lane's 16 x 2 points halve the waves per SIMD of
// kernels that live on them.  lde_narrow therefore uses them up to 2^19 rows and at 2^20 x 2 only.
//
// Results are bit-identical to the integer kernels (same data layouts in HBM, same intermediates T and dst).
// Stage semantics: backend_vulkan.rs:881-942 (DIF form), twiddle layout :977-996.
#pragma once
#include "poseidon2_f64.hip.h"  // MAGIC, MAGIC_P

namespace p3 {
namespace narrow64 {
using namespace narrow;

constexpr double PD = 2013265921.0;

struct Magic { double m, mp; };  // 1.5 * 2^52 and (1.5 * 2^52) P in VGPR pairs the compiler cannot rematerialise (poseidon2_f64.hip.h)
__device__ __forceinline__ Magic pin_magic() {
    Magic r{p2f::MAGIC, p2f::MAGIC_P};
    asm volatile("" : "+v"(r.m), "+v"(r.mp));
    return r;
}
// uniform constants, read from the kernel arguments (SGPR pairs: a VOP3 fp64 op takes one scalar operand)
struct Uni { double npm1, pinv, fbias; };

// a * b mod P for integers |b| < 2^51 / |a/P|, aP = a / P: |result| <= (1/2 + 2^-9) P
__device__ __forceinline__ double mulm(double a, double aP, double b, const Magic& k, double npm1) {
    const double qb = __fma_rn(b, aP, k.m);     // M + rint(ab / P)
    const double c = __fma_rn(qb, npm1, k.mp);  // M - q (P - 1), exact
    const double t = __fma_rn(a, b, c);         // M + (ab - qP) + q, exact
    return t - qb;
}
// Montgomery word of a table entry -> canonical value as a double (tables of the inter-digit twiddles and coset
// scales stay the integer ones of the context; two or three conversions per lane and kernel)
__device__ __forceinline__ double canon(uint32_t monty) { return (double)bb::from_monty(monty); }

// word of a value that came out of a product: |r| <= P/2 + 2^22
__device__ __forceinline__ uint32_t word_centered(double r) {
    const uint32_t w = (uint32_t)(int32_t)r;
    return min(w, w + bb::P);
}
// word of any integer |x| < 2^43: q = rint(x / P - 1/2 + 2^-33) = floor(x / P) exactly (x / P is at least 2^-31 away
// from the next integer unless it is one; the product and the fma round off by less than 2^-40), r = x - qP in [0, P)
__device__ __forceinline__ uint32_t word_any(double x, const Uni& u) {
    const double q = __builtin_rint(__fma_rn(x, u.pinv, u.fbias));
    return (uint32_t)__fma_rn(q, -PD, x);
}

template <int VW> struct DV { double c[VW]; };

template <int VW>
__device__ __forceinline__ DV<VW> ld_words(const void* base, uint32_t off) {
    DV<VW> r;
    if constexpr (VW == 2) { const uint2 x = ldv<uint2>(base, off); r.c[0] = (double)x.x; r.c[1] = (double)x.y; }
    else r.c[0] = (double)ldv<uint32_t>(base, off);
    return r;
}
template <int VW, bool CENTERED>
__device__ __forceinline__ void st_words(void* base, uint32_t off, const DV<VW>& v, const Uni& u) {
    if constexpr (VW == 2) {
        if constexpr (CENTERED) stv<uint2>(base, off, make_uint2(word_centered(v.c[0]), word_centered(v.c[1])));
        else stv<uint2>(base, off, make_uint2(word_any(v.c[0], u), word_any(v.c[1], u)));
    } else {
        stv<uint32_t>(base, off, CENTERED ? word_centered(v.c[0]) : word_any(v.c[0], u));
    }
}

// DIF stages UHI-1 .. ULO on the registers (window at bit A), twiddles {w, w / P} from the LDS copy of the stage table
template <int A, int UHI, int ULO, int VW>
__device__ __forceinline__ void stage_block(DV<VW> (&v)[16], const double2* __restrict__ tw, uint32_t t, const Magic& k, double npm1) {
    const uint32_t tlo = t & ((1u << A) - 1u);
#pragma unroll
    for (int u = UHI - 1; u >= ULO; --u) {
        const int d = u - A;
        double2 w[8];
        if (u > 0) {
#pragma unroll
            for (int jl = 0; jl < 8; jl++)
                if (jl < (1 << d)) w[jl] = tw[(1u << u) - 1u + (tlo | ((uint32_t)jl << A))];
        }
#pragma unroll
        for (int j0 = 0; j0 < 16; j0++) {
            if ((j0 >> d) & 1) continue;
            const int j1 = j0 | (1 << d);
#pragma unroll
            for (int cc = 0; cc < VW; cc++) {
                const double x = v[j0].c[cc], y = v[j1].c[cc];
                v[j0].c[cc] = x + y;
                if (u == 0) v[j1].c[cc] = x - y;
                else v[j1].c[cc] = mulm(w[j0 & ((1 << d) - 1)].x, w[j0 & ((1 << d) - 1)].y, x - y, k, npm1);
            }
        }
    }
}
// Round 1 (window at bit B-4): the lane's 15 twiddles are fetched from the global table at kernel entry, beside the data
template <int B>
__device__ __forceinline__ void load_round1_twiddles(const double2* __restrict__ tw, uint32_t t, double (&w1)[15]) {
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int jl = 0; jl < (1 << d); jl++) w1[(1 << d) - 1 + jl] = tw[(1u << (B - 4 + d)) - 1u + (t | ((uint32_t)jl << (B - 4)))].x;
}
template <int VW>
__device__ __forceinline__ void stage_block_round1(DV<VW> (&v)[16], const double (&w1)[15], const Magic& k, const Uni& un) {
#pragma unroll
    for (int d = 3; d >= 0; --d) {
#pragma unroll
        for (int j0 = 0; j0 < 16; j0++) {
            if ((j0 >> d) & 1) continue;
            const int j1 = j0 | (1 << d);
            const double w = w1[(1 << d) - 1 + (j0 & ((1 << d) - 1))], wP = w * un.pinv;
#pragma unroll
            for (int cc = 0; cc < VW; cc++) {
                const double x = v[j0].c[cc], y = v[j1].c[cc];
                v[j0].c[cc] = x + y;
                v[j1].c[cc] = mulm(w, wP, x - y, k, un.npm1);
            }
        }
    }
}

// registers (window AF) -> LDS -> registers (window AT); PL = doubles per column plane
template <int LQ, int AF, int AT, uint32_t PL, int VW, class TL>
__device__ __forceinline__ void exchange(TL& tiles, DV<VW> (&v)[16], uint32_t t, uint32_t q) {
    double* tile = tiles.next();
    double* wp = tile + lds_base<LQ, AF>(t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++)
#pragma unroll
        for (int cc = 0; cc < VW; cc++) wp[cc * PL + lds_joff<LQ>(j << AF)] = v[j].c[cc];
    __syncthreads();
    const double* rp = tile + lds_base<LQ, AT>(t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++)
#pragma unroll
        for (int cc = 0; cc < VW; cc++) v[j].c[cc] = rp[cc * PL + lds_joff<LQ>(j << AT)];
}
template <int B, int LQ, int VW, class TL>
__device__ __forceinline__ void dif_rounds(DV<VW> (&v)[16], TL& tile, const double (&w1)[15], const double2* twl, uint32_t t, uint32_t q,
                                           const Magic& k, const Uni& un) {
    constexpr int A1 = B - 4, A2 = B > 8 ? B - 8 : 0;
    constexpr uint32_t PL = lds_rows(B) << LQ;
    stage_block_round1(v, w1, k, un);
    exchange<LQ, A1, A2, PL>(tile, v, t, q);
    stage_block<A2, A1, A2>(v, twl, t, k, un.npm1);
    if constexpr (B > 8) {
        exchange<LQ, A2, 0, PL>(tile, v, t, q);
        stage_block<0, A2, 0>(v, twl, t, k, un.npm1);
    }
}
// final layout (position pt_of<0>) -> LDS row = frequency rev_B(position) -> first layout (row pt_of<B-4>)
template <int B, int LQ, int VW, class TL>
__device__ __forceinline__ void to_natural(TL& tiles, DV<VW> (&v)[16], uint32_t t, uint32_t q) {
    constexpr uint32_t PL = lds_rows(B) << LQ;
    double* tile = tiles.next();
    const uint32_t rt = rev_bits(t, B - 4);
    double* wp = tile + (((rt + (rt >> 4)) << LQ) + q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++)
#pragma unroll
        for (int cc = 0; cc < VW; cc++) wp[cc * PL + lds_joff<LQ>(crev(j, 4) << (B - 4))] = v[j].c[cc];
    __syncthreads();
    const double* rp = tile + lds_base<LQ, B - 4>(t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++)
#pragma unroll
        for (int cc = 0; cc < VW; cc++) v[j].c[cc] = rp[cc * PL + lds_joff<LQ>(j << (B - 4))];
}

// The LAST hand-over of K1 (to natural frequency order) and K3 (to row order) on WORDS: the values are reduced for the store
// anyway, and 4-byte words halve the LDS traffic of that exchange (the tile written is the one next() hands out: the words
// occupy the front of it, so the tiles' barrier protocol is unchanged).
template <int B, int LQ, int VW, bool NATURAL, class TL>
__device__ __forceinline__ void exchange_words(TL& tiles, typename Vec<VW>::T (&w)[16], uint32_t t, uint32_t q) {
    using V = typename Vec<VW>::T;
    V* tile = reinterpret_cast<V*>(tiles.next());
    if constexpr (NATURAL) {
        const uint32_t rt = rev_bits(t, B - 4);
        V* wp = tile + (((rt + (rt >> 4)) << LQ) + q);
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) wp[lds_joff<LQ>(crev(j, 4) << (B - 4))] = w[j];
    } else {
        V* wp = tile + lds_base<LQ, 0>(t, q);
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) wp[lds_joff<LQ>(j)] = w[j];
    }
    __syncthreads();
    const V* rp = tile + lds_base<LQ, B - 4>(t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) w[j] = rp[lds_joff<LQ>(j << (B - 4))];
}
template <int VW, bool CENTERED>
__device__ __forceinline__ typename Vec<VW>::T to_words(const DV<VW>& v, const Uni& u) {
    if constexpr (VW == 2) {
        if constexpr (CENTERED) return make_uint2(word_centered(v.c[0]), word_centered(v.c[1]));
        else return make_uint2(word_any(v.c[0], u), word_any(v.c[1], u));
    } else {
        if constexpr (CENTERED) return word_centered(v.c[0]);
        else return word_any(v.c[0], u);
    }
}

// pw[r] = c * phi^r for r < 16, by doubling
__device__ __forceinline__ void power_ladder16(double c, double phi, double (&pw)[16], const Magic& k, const Uni& un) {
    pw[0] = c;
    double step = phi;
#pragma unroll
    for (int len = 1; len < 16; len <<= 1) {
        const double sP = step * un.pinv;
#pragma unroll
        for (int i = 0; i < len; i++) pw[len + i] = mulm(step, sP, pw[i], k, un.npm1);
        if (len * 2 < 16) step = mulm(step, sP, step, k, un.npm1);
    }
}
// v[j] *= pw[REV ? rev4(j) : j]
template <bool REV, int VW>
__device__ __forceinline__ void scale_by(DV<VW> (&v)[16], const double (&pw)[16], const Magic& k, const Uni& un) {
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        const double w = pw[REV ? crev(j, 4) : j], wP = w * un.pinv;
#pragma unroll
        for (int cc = 0; cc < VW; cc++) v[j].c[cc] = mulm(w, wP, v[j].c[cc], k, un.npm1);
    }
}

template <int B, int LQ, int VW, int NT>
constexpr size_t lds_bytes(int tables) {
    return ((size_t)8 * VW * NT * lds_rows(B) << LQ) + (size_t)tables * ((size_t)16 << (B - 4));
}

}  // namespace narrow64

// K1: first inverse digit (the high n1 bits of the row index), inter-digit twiddle, transposed store.
template <int B, int LQ, int VW, int NT>
__global__ void __launch_bounds__(1 << (B - 4 + LQ)) narrow64_inv1_kernel(NarrowArgs a) {
    using namespace narrow64;
    constexpr uint32_t NQ = 1u << LQ, NTH = 1u << (B - 4 + LQ), PL = lds_rows(B) << LQ;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    double* t0 = reinterpret_cast<double*>(smem);
    constexpr uint32_t TD = VW * PL;  // doubles per tile
    Tiles<double, (NT > 1)> tile{t0, t0 + (NT - 1) * TD};
    double2* twl = reinterpret_cast<double2*>(t0 + NT * TD);
    const Magic mk = pin_magic();
    const Uni un{a.neg_pm1, a.pinv, a.fbias};
    const uint32_t q = threadIdx.x & (NQ - 1), t = threadIdx.x >> LQ;
    const uint32_t s = tile_of_block<LQ, VW>(blockIdx.x, a.xcd_remap) * NQ + q;
    const uint32_t lo = slot_row(a, s), cp = slot_col(a, s, lo);
    const uint32_t rowstride = a.W << a.n2;
    const uint32_t ld_off = (VW * s + t * rowstride) * 4u;
    DV<VW> v[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) v[j] = ld_words<VW>(a.src + ((uint64_t)j << (B - 4)) * rowstride, ld_off);
    double w1[15];
    load_round1_twiddles<B>(a.stage_twd, t, w1);
    for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl[i] = a.stage_twd[i];
    const double c = canon(two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo * rev_bits(t, B - 4)));
    const double phi = canon(two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo << (B - 4)));
    dif_rounds<B, LQ>(v, tile, w1, twl, t, q, mk, un);
    {
        double pw[16];
        power_ladder16(c, phi, pw, mk, un);
        scale_by<true>(v, pw, mk, un);
    }
    using WV = typename Vec<VW>::T;
    WV w[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) w[j] = to_words<VW, true>(v[j], un);
    exchange_words<B, LQ, VW, true>(tile, w, t, q);
    if (a.blocked) {
        const uint32_t blk_off = ((((lo >> 2) << (B - 2)) + (t >> 2)) * 16u + (lo & 3u) * 4u + (t & 3u)) * 8u + cp * 4u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) stv<WV>(a.dst + ((uint64_t)j << (B - 6)) * 32u, blk_off, w[j]);
        return;
    }
    const uint32_t st_off = (((lo << B) + t) * a.W + VW * cp) * 4u;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) stv<WV>(a.dst + ((uint64_t)j << (B - 4)) * a.W, st_off, w[j]);
}

// K2: second inverse digit; per coset: scale by (shift g^j)^k / N, first forward digit, twiddle, strided store.
template <int B, int LQ, int VW, int NT, bool LEAN>
__global__ void __launch_bounds__(1 << (B - 4 + LQ)) narrow64_mid_kernel(NarrowArgs a) {
    using namespace narrow64;
    constexpr uint32_t NQ = 1u << LQ, NTH = 1u << (B - 4 + LQ), PL = lds_rows(B) << LQ;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    double* t0 = reinterpret_cast<double*>(smem);
    constexpr uint32_t TD = VW * PL;
    Tiles<double, (NT > 1)> tile{t0, t0 + (NT - 1) * TD};
    double2* twl_i = reinterpret_cast<double2*>(t0 + NT * TD);
    double2* twl_f = twl_i + (1u << (B - 4));
    const Magic mk = pin_magic();
    const Uni un{a.neg_pm1, a.pinv, a.fbias};
    const uint32_t q = threadIdx.x & (NQ - 1), t = threadIdx.x >> LQ;
    const uint32_t s = tile_of_block<LQ, VW>(blockIdx.x, a.xcd_remap) * NQ + q;
    const uint32_t k1 = slot_row(a, s);
    const uint32_t rowstride = a.W << a.n1;
    const uint32_t ld_off = (VW * s + t * rowstride) * 4u, st_off = (VW * s + (t << 4) * rowstride) * 4u;
    const bool blocked = a.blocked;
    const uint32_t blk_off = ((((t >> 2) << (a.n1 - 2)) + (k1 >> 2)) * 16u + (t & 3u) * 4u + (k1 & 3u)) * 8u +
                             (VW == 1 ? (s & 1u) * 4u : 0u);
    const uint64_t blk_jstride = ((uint64_t)32u << (B - 6)) << (a.n1 - 2);
    DV<VW> c[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++)
        c[j] = blocked ? ld_words<VW>(a.src + (uint64_t)j * blk_jstride, blk_off) : ld_words<VW>(a.src + ((uint64_t)j << (B - 4)) * rowstride, ld_off);
    {
        double w1[15];
        load_round1_twiddles<B>(a.stage_twd, t, w1);
        for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl_i[i] = a.stage_twd[i];
        for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl_f[i] = a.stage_twd_fwd[i];
        if (!a.from_coeffs) dif_rounds<B, LQ>(c, tile, w1, twl_i, t, q, mk, un);
    }
    const double c0 = canon(two_level(a.twf_lo, a.twf_hi, a.twf_T, (uint64_t)k1 * rev_bits(t, B - 4)));
    const double phi0 = canon(two_level(a.twf_lo, a.twf_hi, a.twf_T, (uint64_t)k1 << (B - 4)));
    const uint64_t kbase = (uint64_t)k1 + ((uint64_t)t << a.n1);
    const uint32_t cos0 = blockIdx.y * a.cos_per_block, ncos = cos0 + a.cos_per_block;
    uint32_t sc_next = two_level(a.sc_lo[cos0], a.sc_hi[cos0], a.sc_T, kbase);
    if (!a.from_coeffs) {  // c[j] = coefficient k = k1 + N1 * k2, k2 = (j << (B-4)) | t
        to_natural<B, LQ>(tile, c, t, q);
    }
    double pw2[16];
    if constexpr (!LEAN) power_ladder16(c0, phi0, pw2, mk, un);
    for (uint32_t jc = cos0; jc < ncos; jc++) {
        const double sc = canon(sc_next);
        if (jc + 1 < ncos) sc_next = two_level(a.sc_lo[jc + 1], a.sc_hi[jc + 1], a.sc_T, kbase);
        double w1[15];  // in flight while the scale ladder runs
        load_round1_twiddles<B>(a.stage_twd_fwd, t, w1);
        DV<VW> v[16];
        {
            double pw[16];
            power_ladder16(sc, canon(a.sc_phi[jc]), pw, mk, un);
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) {
                const double w = pw[j], wP = w * un.pinv;
#pragma unroll
                for (int cc = 0; cc < VW; cc++) v[j].c[cc] = mulm(w, wP, c[j].c[cc], mk, un.npm1);
            }
        }
        dif_rounds<B, LQ>(v, tile, w1, twl_f, t, q, mk, un);
        if constexpr (LEAN) power_ladder16(c0, phi0, pw2, mk, un);
        scale_by<true>(v, pw2, mk, un);
        uint32_t* o = a.dst + ((uint64_t)rev_bits(jc, a.added) << a.n) * a.W;  // position (t << 4) | j of the coset's block
        if (blocked) {
            // blocked positions straight from the final layout: position (t << 4) | j -> block ((t << 2) | (j >> 2), k1 group)
            const uint32_t fin_off = (((t << 2) << (a.n1 - 2)) + (k1 >> 2)) * 128u + (k1 & 3u) * 8u + (VW == 1 ? (s & 1u) * 4u : 0u);
#pragma unroll
            for (uint32_t j = 0; j < 16; j++)
                st_words<VW, true>(o + ((uint64_t)(j >> 2) << (a.n1 - 2)) * 32u, fin_off + (j & 3u) * 32u, v[j], un);
            continue;
        }
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) st_words<VW, true>(o + (uint64_t)j * rowstride, st_off, v[j], un);
    }
}

// K3: last forward digit on contiguous blocks of 2^B rows, in place.
template <int B, int LQ, int VW, int NT>
__global__ void __launch_bounds__(1 << (B - 4 + LQ)) narrow64_fwd2_kernel(NarrowArgs a) {
    using namespace narrow64;
    constexpr uint32_t NQ = 1u << LQ, NTH = 1u << (B - 4 + LQ), PL = lds_rows(B) << LQ;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    double* t0 = reinterpret_cast<double*>(smem);
    constexpr uint32_t TD = VW * PL;  // doubles per tile
    Tiles<double, (NT > 1)> tile{t0, t0 + (NT - 1) * TD};
    double2* twl = reinterpret_cast<double2*>(t0 + NT * TD);
    const Magic mk = pin_magic();
    const Uni un{a.neg_pm1, a.pinv, a.fbias};
    const uint32_t q = threadIdx.x & (NQ - 1), t = threadIdx.x >> LQ;
    const uint32_t tile0 = k3_tile_of_block(blockIdx.x, a.k3_pairs);
    const uint32_t s = tile0 * NQ + q;
    const uint32_t blk0 = slot_row(a, tile0 * NQ), blk = slot_row(a, s), cp = slot_col(a, s, blk);
    uint32_t* p = a.dst + ((uint64_t)blk0 << B) * a.W;
    const uint32_t off = ((((blk - blk0) << B) + t) * a.W + VW * cp) * 4u;
    DV<VW> v[16];
    if (a.blocked) {
        const uint32_t g0 = blk0 & ~3u;  // see narrow_fwd2_kernel: a.src == a.dst unless the tile owns half a group
        const uint32_t* ps = a.src + ((uint64_t)g0 << B) * a.W;
        const uint32_t blk_off = ((t >> 2) * 16u + (blk - g0) * 4u + (t & 3u)) * 8u + cp * 4u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = ld_words<VW>(ps + ((uint64_t)j << (B - 6)) * 32u, blk_off);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = ld_words<VW>(p + ((uint64_t)j << (B - 4)) * a.W, off);
    }
    double w1[15];
    load_round1_twiddles<B>(a.stage_twd, t, w1);
    for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl[i] = a.stage_twd[i];
    dif_rounds<B, LQ>(v, tile, w1, twl, t, q, mk, un);
    using WV = typename Vec<VW>::T;
    WV w[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) w[j] = to_words<VW, false>(v[j], un);
    exchange_words<B, LQ, VW, false>(tile, w, t, q);  // position order is the wanted order: 16 consecutive lanes hold 16 consecutive rows
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) stv<WV>(p + ((uint64_t)j << (B - 4)) * a.W, off, w[j]);
}

}  // namespace p3
