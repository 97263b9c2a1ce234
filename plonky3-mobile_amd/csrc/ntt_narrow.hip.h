// Three-launch coset LDE for NARROW matrices (W = 2, 4, 6, 8 or 16 columns: the fib_air trace, quotient and hiding shapes) and, with
// 128-byte tile rows, for WIDE ones (any width >= 64 at 2^16 rows).
//
// The general plans (ntt.hip / ntt_fast.hip.h) move one 32-bit word per lane and cut 2^n rows into three 6-8 stage
// digits per direction: five launches for 2^20 -> 2^21, every one re-reading and re-writing the matrix.  Here the
// height is cut into TWO digits (n = n1 + n2, 8..12 stages each), a lane moves a PAIR of columns (8-byte
// accesses, one twiddle serves both columns; single columns for the small heights, where twice the waves matter
// more) and carries 16 points through radix-16 register rounds:
//
//   K1  narrow_inv1_kernel<n1>   x[r1*N2 + r2]  --DIF over r1, twiddle w^-(r2*k1)-->  T[r2*N1 + k1]   (transposed)
//   K2  narrow_mid_kernel<n2>    T[r2*N1 + k1]  --DIF over r2--> c[k1 + N1*k2]  (coefficients stay on chip), then for
//                                every coset j < 2^added of the LDE domain: scale by (shift g^j)^k / N, DIF over k2,
//                                twiddle w^(k1*m1), store at  rev(j)*N + rev(m1)*N1 + k1
//   K3  narrow_fwd2_kernel<n1>   contiguous blocks of N1 rows, DIF over k1, in place
//
// The LDE over shift*<g> (g of order 2^added * N) in bit-reversed row order is exactly 2^added size-N coset
// transforms laid side by side (row rev(j)*N + rev_n(m) = evaluation at shift * g^j * w_N^m), so no stage ever
// runs over the zero padding.  Traffic: (1+1) + (1+2^a) + (2^a+2^a) matrix sweeps = 72 MB for the 2^20 x 2 trace
// at blowup 2 (was 127 MB), 25 MB algorithmic.
//
// LDS tile: [2^B points][2^LQ slots] of uint2 (LQ = 2: 32-byte row segments; 1 for the 12-stage middle kernel), one padding row per 16 so that each of the three register layouts is bank-conflict
// free for ds_read/write_b64 and a lane's sixteen rows are immediates off one base address.  The first round's 15 stage twiddles per
// lane are fetched from the global table at kernel entry, beside the data loads; the later rounds read a small
// LDS copy that is complete by the first exchange: no barrier precedes the first butterfly.
// Included by ntt.hip (needs two_level, rev_bits, power_ladder, crev).
#pragma once

// LDS tiles per kernel: two cost one barrier per hand-over instead of two (the tile written now was last read two hand-overs ago).
// Round 5 retired the experiment builds that lived here (records in profiles/): the load / LDS / store skeleton without
// butterflies (r03_lde_skeleton_vs_full.txt), the column-sequential middle kernel and the 128-VGPR shared-CU build (both spill:
// 2^24 middle pass 549 -> 1063 us / 586 -> 898 us), the ds_bpermute twiddle broadcast (r02_twiddle_shuffle_vs_lds.txt: 1-4 % slower),
// write-through / non-temporal stores (r03_store_policy_ab.txt), the phase stamps (r03_lde_k1_phase_stamps.txt) and the middle
// kernel's extra hand-over for whole-line stores.
constexpr uint32_t NARROW_EDGE_TILES = 2;  // first and last kernel, digits below 11 stages (two cost a workgroup per CU from 11 stages on)
constexpr uint32_t NARROW_MID_TILES = 2;   // middle kernel (one for 1024-thread tiles)
namespace p3 {

struct NarrowArgs {
    const uint32_t* src;
    uint32_t* dst;
    uint32_t n, n1, n2, W, wsl;   // wsl = log2(W / VW): slots (lane vectors of VW words) per row; 0xffffffff when that is not a
                                  // power of two (W = 6, 12: the hiding prover's trace), then spr is divided by
    uint32_t spr;                 // slots per row = W / VW
    uint32_t from_coeffs;         // K2: src holds COEFFICIENTS (natural order): no inverse digit, no K1 before it
    uint32_t added;
    const uint32_t* stage_tw;     // this kernel's direction: stage u at offset 2^u - 1 (reference layout, 12 stages)
    const uint32_t* stage_tw_fwd; // K2: forward table
    const uint32_t* tw_lo;        // inter-digit twiddles w_N^(+-e), two-level
    const uint32_t* tw_hi;
    uint32_t tw_T;
    const uint32_t* twf_lo;       // K2: forward inter-digit twiddles
    const uint32_t* twf_hi;
    uint32_t twf_T;
    const uint32_t* sc_lo[8];     // K2: per coset j, value(k) = (shift g^j)^k / N
    const uint32_t* sc_hi[8];
    uint32_t sc_T;
    uint32_t sc_phi[8];           // (shift g^j)^(N1 * 2^(n2-4))
    uint32_t xcd_remap;           // 1: tile count is a multiple of 32, spread groups of 4 adjacent tiles per XCD
    uint32_t cos_per_block;       // K2: cosets one workgroup transforms (grid.y = 2^added / cos_per_block); splitting the
                                  // cosets over workgroups repeats the inverse digit but doubles a thin grid's waves
    // fp64 kernels (ntt_narrow_f64.hip.h): stage tables as {w, w / P} doubles of CANONICAL values, same layout
    const double2* stage_twd;
    const double2* stage_twd_fwd;
    double neg_pm1, pinv, fbias;  // -(P - 1), 1 / P, -1/2 + 2^-33: uniform operands of the fp64 product / floor reduction
    uint32_t k3_pairs;            // K3: log2 of the consecutive tiles dealt to one XCD.  1 with 2-row tiles over blocked input (partner tiles =
                                  // the two halves of every 128-byte block; K3 then reads a.src out of place: in place the partner's half
                                  // would be overwritten); 4 for wide matrices (neighbours in a row complete each other's lines)
    uint32_t blocked;             // W = 2: the two intermediates are stored in 128-byte blocks of
                                  // 4 x 4 (row of one digit, row of the other) pairs, so that the kernel that reads
                                  // them strided touches whole cache lines instead of 32-byte segments
};

namespace narrow {


// element vector of a lane: VW = 2 columns (uint2, 8-byte accesses) or VW = 1 (uint32_t: twice the lanes and waves for
// the same matrix — on CDNA a SIMD needs >= 2-4 resident waves to reach its VALU issue rate, and the small LDEs
// have only 1-2 waves per SIMD with column pairs)
template <int VW> struct Vec;
template <> struct Vec<1> { using T = uint32_t; };
template <> struct Vec<2> { using T = uint2; };
__device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) { return bb::add(a, b); }
__device__ __forceinline__ uint32_t sub2(uint32_t a, uint32_t b) { return bb::sub(a, b); }
__device__ __forceinline__ uint32_t subl2(uint32_t a, uint32_t b) { return a - b + bb::P; }
__device__ __forceinline__ uint32_t mul2(uint32_t a, uint32_t w) { return bb::mul(a, w); }
__device__ __forceinline__ uint2 add2(uint2 a, uint2 b) { return make_uint2(bb::add(a.x, b.x), bb::add(a.y, b.y)); }
__device__ __forceinline__ uint2 sub2(uint2 a, uint2 b) { return make_uint2(bb::sub(a.x, b.x), bb::sub(a.y, b.y)); }
// a - b + P in (0, 2P): a valid (unreduced) operand of the Montgomery product
__device__ __forceinline__ uint2 subl2(uint2 a, uint2 b) { return make_uint2(a.x - b.x + bb::P, a.y - b.y + bb::P); }
__device__ __forceinline__ uint2 mul2(uint2 a, uint32_t w) { return make_uint2(bb::mul(a.x, w), bb::mul(a.y, w)); }

// point held in register j of thread t when the 4-bit register window sits at bit A: pt = T(t) | (j << A)
template <int A>
__device__ __forceinline__ uint32_t pt_of(uint32_t t, uint32_t j) {
    return ((t >> A) << (A + 4)) | (j << A) | (t & ((1u << A) - 1u));
}
// LDS row of a point: one padding row after every 16 (row = pt + (pt >> 4)).  T(t) and j << A occupy disjoint
// bits, so (T + J) >> 4 = (T >> 4) + (J >> 4) and the row splits into a per-thread base plus a compile-time
// offset per register: sixteen accesses cost one address VGPR and sixteen immediates.  With the odd stride
// 17 every register layout is bank-conflict free for ds_read/write_b64: a half-wave (32 lanes x 8 bytes = all
// 64 banks) covers 32 / NQ consecutive t, whose rows differ in their low log2(32 / NQ) bits (window at bit B-4:
// consecutive rows; window at 0: rows 17 apart; window at B-8: low bits of t are consecutive rows, the next
// bits add 4 * 17 per step).
template <int LQ, int A>
__device__ __forceinline__ uint32_t lds_base(uint32_t t, uint32_t q) {
    const uint32_t T = ((t >> A) << (A + 4)) | (t & ((1u << A) - 1u));
    return ((T + (T >> 4)) << LQ) + q;
}
template <int LQ>
constexpr uint32_t lds_joff(uint32_t J) { return (J + (J >> 4)) << LQ; }
constexpr uint32_t lds_rows(int B) { return (1u << B) + (1u << (B - 4)); }

// DIF stages UHI-1 .. ULO on the registers (window at bit A); stage u pairs points differing in bit u:
// (a, b) -> (a + b, (a - b) * w_{2^(u+1)}^(pt mod 2^u))   [stage semantics of backend_vulkan.rs:881-942, DIF form]
template <int A, int UHI, int ULO, class V>
__device__ __forceinline__ void stage_block(V (&v)[16], const uint32_t* __restrict__ tw, uint32_t t) {
    const uint32_t tlo = t & ((1u << A) - 1u);
#pragma unroll
    for (int u = UHI - 1; u >= ULO; --u) {
        const int d = u - A;
        uint32_t w[8];
        if (u > 0) {
            // an LDS BROADCAST read (the index is (lane >> LQ) & mask: not uniform, not a DPP row pattern); the ds_bpermute form
            // measured 1-4 % slower (profiles/r02_twiddle_shuffle_vs_lds.txt)
#pragma unroll
            for (int jl = 0; jl < 8; jl++)
                if (jl < (1 << d)) w[jl] = tw[(1u << u) - 1u + (tlo | ((uint32_t)jl << A))];
        }
#pragma unroll
        for (int j0 = 0; j0 < 16; j0++) {
            if ((j0 >> d) & 1) continue;
            const int j1 = j0 | (1 << d);
            const V x = v[j0], y = v[j1];
            v[j0] = add2(x, y);
            if (u == 0) v[j1] = sub2(x, y);
            else v[j1] = mul2(subl2(x, y), w[j0 & ((1 << d) - 1)]);
        }
    }
}

// Round 1 (window at bit B-4, stages B-1 .. B-4) with its 15 twiddles already in registers: they are fetched from
// the global table at kernel entry, next to the data loads, so no butterfly ever waits for an L2 round trip.
template <int B>
__device__ __forceinline__ void load_round1_twiddles(const uint32_t* __restrict__ tw, uint32_t t, uint32_t (&w1)[15]) {
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int jl = 0; jl < (1 << d); jl++) w1[(1 << d) - 1 + jl] = tw[(1u << (B - 4 + d)) - 1u + (t | ((uint32_t)jl << (B - 4)))];
}
template <class V>
__device__ __forceinline__ void stage_block_round1(V (&v)[16], const uint32_t (&w1)[15]) {
#pragma unroll
    for (int d = 3; d >= 0; --d) {
#pragma unroll
        for (int j0 = 0; j0 < 16; j0++) {
            if ((j0 >> d) & 1) continue;
            const int j1 = j0 | (1 << d);
            const V x = v[j0], y = v[j1];
            v[j0] = add2(x, y);
            v[j1] = mul2(subl2(x, y), w1[(1 << d) - 1 + (j0 & ((1 << d) - 1))]);
        }
    }
}

// One or two LDS tiles.  With two, consecutive hand-overs alternate between them and need ONE barrier each (between
// the writes and the reads): the tile written now was last read two hand-overs ago, and the barrier of the
// hand-over in between already separates those reads from these writes.  With one tile (TWO = false) every hand-over
// also waits, before writing, for the previous one's reads.
template <class V, bool TWO>
struct Tiles {
    V* a;
    V* b;
    __device__ __forceinline__ V* next() {
        if constexpr (TWO) { V* r = a; a = b; b = r; return r; }
        __syncthreads();
        return a;
    }
};

// registers (window AF) -> LDS -> registers (window AT)
template <int LQ, int AF, int AT, class V, class TL>
__device__ __forceinline__ void exchange(TL& tiles, V (&v)[16], uint32_t t, uint32_t q) {
    V* tile = tiles.next();
    V* wp = tile + lds_base<LQ, AF>(t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) wp[lds_joff<LQ>(j << AF)] = v[j];
    __syncthreads();
    const V* rp = tile + lds_base<LQ, AT>(t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) v[j] = rp[lds_joff<LQ>(j << AT)];
}

// B-stage DIF of the tile: in: v[j] = point pt_of<B-4>(t, j) (natural order); out: v[j] = position pt_of<0>(t, j),
// which holds frequency rev_B(position).  twl: stage table in LDS (stages below B-4 at least).
template <int B, int LQ, class V, class TL>
__device__ __forceinline__ void dif_rounds_after1(V (&v)[16], TL& tile, const uint32_t* twl, uint32_t t, uint32_t q) {
    constexpr int A1 = B - 4, A2 = B > 8 ? B - 8 : 0;
    exchange<LQ, A1, A2, V>(tile, v, t, q);
    stage_block<A2, A1, A2>(v, twl, t);
    if constexpr (B > 8) {
        exchange<LQ, A2, 0, V>(tile, v, t, q);
        stage_block<0, A2, 0>(v, twl, t);
    }
}
template <int B, int LQ, class V, class TL>
__device__ __forceinline__ void dif_rounds(V (&v)[16], TL& tile, const uint32_t (&w1)[15], const uint32_t* twl, uint32_t t, uint32_t q) {
    stage_block_round1(v, w1);
    dif_rounds_after1<B, LQ>(v, tile, twl, t, q);
}
// registers in final layout (position pt_of<0>) -> LDS row = frequency rev_B(position) = (rev4(j) << (B-4)) | rev(t)
// -> registers in the first layout (row pt_of<B-4>): natural frequency order, 16 consecutive rows per 16 lanes.
template <int B, int LQ, class V, class TL>
__device__ __forceinline__ void to_natural(TL& tiles, V (&v)[16], uint32_t t, uint32_t q) {
    V* tile = tiles.next();
    const uint32_t rt = rev_bits(t, B - 4);
    V* wp = tile + (((rt + (rt >> 4)) << LQ) + q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) wp[lds_joff<LQ>(crev(j, 4) << (B - 4))] = v[j];
    __syncthreads();
    const V* rp = tile + lds_base<LQ, B - 4>(t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) v[j] = rp[lds_joff<LQ>(j << (B - 4))];
}
// same hand-over without the bit reversal (K3: position order is already the wanted order)
template <int B, int LQ, class V, class TL>
__device__ __forceinline__ void to_rows(TL& tile, V (&v)[16], uint32_t t, uint32_t q) {
    exchange<LQ, 0, B - 4, V>(tile, v, t, q);
}

// Adjacent tiles (four of them for 32-byte segments) share 128-byte lines of the strided side: keep them on one XCD (workgroups are dealt to
// the 8 XCDs round-robin by blockIdx) so the line is fetched into one L2 only.
template <int LQ, int VW>
__device__ __forceinline__ uint32_t tile_of_block(uint32_t bid, uint32_t remap) {
    if (!remap) return bid;
    // remap = 1: groups of the tiles that share a 128-byte line; remap >= 2: groups of 2^remap consecutive tiles (wide matrices: rows are
    // not multiples of 128 bytes, so a tile's row segments straddle lines that its neighbours in the row complete)
    const uint32_t LG = remap == 1 ? (VW == 2 ? 4u : 5u) - LQ : remap;
    const uint32_t xcd = bid & 7u, s = bid >> 3;
    return ((s >> LG) << (LG + 3)) | (xcd << LG) | (s & ((1u << LG) - 1u));
}

// Global accesses as UNIFORM base + 32-bit per-lane byte offset (global_load/store ... saddr): the sixteen row
// offsets of a lane's points are uniform, so they live in SGPRs instead of sixteen 64-bit VGPR addresses.
template <class V>
__device__ __forceinline__ V ldv(const void* base, uint32_t off) {
    return *reinterpret_cast<const V*>(static_cast<const char*>(base) + off);
}
template <class V>
__device__ __forceinline__ void stv(void* base, uint32_t off, V v) {
    *reinterpret_cast<V*>(static_cast<char*>(base) + off) = v;
}

// v[j] *= c * phi^(idx(j)), idx(j) = REV ? rev4(j) : j
template <bool REV, class V>
__device__ __forceinline__ void scale_ladder(V (&v)[16], uint32_t c, uint32_t phi) {
    uint32_t pw[16];
    power_ladder<16>(c, phi, pw);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) v[j] = mul2(v[j], pw[REV ? crev(j, 4) : j]);
}

}  // namespace narrow

// K3 tiles of 2 row blocks over blocked input: blocks b and b + 8 of the grid share an XCD (round-robin dealing), give them the two
// halves of one group of four row blocks (speed only: the result does not depend on placement)
__device__ __forceinline__ uint32_t k3_tile_of_block(uint32_t bid, uint32_t lg) {  // lg = log2(consecutive tiles per XCD), 0 = as dealt
    if (!lg) return bid;
    const uint32_t xcd = bid & 7u, s = bid >> 3;
    return ((s >> lg) << (lg + 3)) | (xcd << lg) | (s & ((1u << lg) - 1u));
}
// slot index -> (row group, column slot)
__device__ __forceinline__ uint32_t slot_row(const NarrowArgs& a, uint32_t s) { return a.wsl != 0xffffffffu ? s >> a.wsl : s / a.spr; }
__device__ __forceinline__ uint32_t slot_col(const NarrowArgs& a, uint32_t s, uint32_t row) {
    return a.wsl != 0xffffffffu ? s & ((1u << a.wsl) - 1u) : s - row * a.spr;
}

// K1: first inverse digit (the high n1 bits of the row index), transposed store.
template <int B, int LQ, int VW>
__global__ void __launch_bounds__(1 << (B - 4 + LQ)) narrow_inv1_kernel(NarrowArgs a) {
    using namespace narrow;
    using V = typename Vec<VW>::T;
    constexpr uint32_t NQ = 1u << LQ, NTH = 1u << (B - 4 + LQ);
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr uint32_t NT = B >= 11 ? 1 : NARROW_EDGE_TILES;              // two tiles: one barrier per hand-over
    Tiles<V, (NT > 1)> tile{reinterpret_cast<V*>(smem), reinterpret_cast<V*>(smem) + (NT - 1) * (lds_rows(B) << LQ)};
    uint32_t* twl = smem + (NT * VW * lds_rows(B) << LQ);                  // stages below B-4: 2^(B-4) - 1 words
    const uint32_t q = threadIdx.x & (NQ - 1), t = threadIdx.x >> LQ;
    const uint32_t s = tile_of_block<LQ, VW>(blockIdx.x, a.xcd_remap) * NQ + q;  // slot (VW words) within a row group of N2 rows
    const uint32_t lo = slot_row(a, s), cp = slot_col(a, s, lo);
    const uint32_t rowstride = a.W << a.n2;                               // words between r1 and r1 + 1
    const uint32_t ld_off = (VW * s + t * rowstride) * 4u;
    V v[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) v[j] = ldv<V>(a.src + ((uint64_t)j << (B - 4)) * rowstride, ld_off);
    uint32_t w1[15];
    load_round1_twiddles<B>(a.stage_tw, t, w1);
    for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl[i] = a.stage_tw[i];
    // position (t << 4) | j holds k1 = rev_B(position) = (rev4(j) << (B-4)) | rev(t): twiddle w^-(lo * k1)
    const uint32_t c = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo * rev_bits(t, B - 4));
    const uint32_t phi = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo << (B - 4));
    dif_rounds<B, LQ>(v, tile, w1, twl, t, q);
    scale_ladder<true>(v, c, phi);
    to_natural<B, LQ>(tile, v, t, q);
    // T[(lo * N1 + k1) * W + VW cp], k1 = pt_of<B-4>(t, j): consecutive lanes (cp, then k1) are contiguous for W <= NQ * VW
    if (a.blocked) {
        // W = 2: block (r2 >> 2, k1 >> 2) of 128 bytes holds the row pairs (r2 & 3, k1 & 3), 8 bytes each (cp = column
        // for single-column lanes); this tile is the r2 group lo >> 2
        const uint32_t blk_off = ((((lo >> 2) << (B - 2)) + (t >> 2)) * 16u + (lo & 3u) * 4u + (t & 3u)) * 8u + cp * 4u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) stv<V>(a.dst + ((uint64_t)j << (B - 6)) * 32u, blk_off, v[j]);
        return;
    }
    const uint32_t st_off = (((lo << B) + t) * a.W + VW * cp) * 4u;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) stv<V>(a.dst + ((uint64_t)j << (B - 4)) * a.W, st_off, v[j]);
}

// K2: second inverse digit, then per coset: scale, first forward digit, twiddle, strided store.
template <int B, int LQ, int VW>
__global__ void __launch_bounds__(1 << (B - 4 + LQ), 1)
narrow_mid_kernel(NarrowArgs a) {
    using namespace narrow;
    using V = typename Vec<VW>::T;
    constexpr uint32_t NQ = 1u << LQ, NTH = 1u << (B - 4 + LQ);
    constexpr bool LEAN = NTH >= 512;  // at most 256 (1024 threads: 128) VGPRs per lane: rebuild the output ladder per coset
    constexpr uint32_t NT = NTH >= 1024 ? 1 : NARROW_MID_TILES;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // two tiles (one barrier per hand-over): the 15 hand-overs of a blowup-4 middle pass are this kernel's stalls
    Tiles<V, (NT > 1)> tile{reinterpret_cast<V*>(smem), reinterpret_cast<V*>(smem) + (NT - 1) * (lds_rows(B) << LQ)};
    uint32_t* twl_i = smem + (NT * VW * lds_rows(B) << LQ);   // inverse stages below B-4
    uint32_t* twl_f = twl_i + (1u << (B - 4));                // forward stages below B-4
    const uint32_t q = threadIdx.x & (NQ - 1), t = threadIdx.x >> LQ;
    const uint32_t s = tile_of_block<LQ, VW>(blockIdx.x, a.xcd_remap) * NQ + q;  // slot within a group of N1 rows
    const uint32_t k1 = slot_row(a, s);
    const uint32_t rowstride = a.W << a.n1;
    const uint32_t ld_off = (VW * s + t * rowstride) * 4u, st_off = (VW * s + (t << 4) * rowstride) * 4u;
    const bool blocked = a.blocked;
    // blocked (W = 2): pair (row r, k1) sits in block (r >> 2, k1 >> 2) at (r & 3) * 4 + (k1 & 3); this tile is one k1 group
    const uint32_t blk_off = ((((t >> 2) << (a.n1 - 2)) + (k1 >> 2)) * 16u + (t & 3u) * 4u + (k1 & 3u)) * 8u +
                             (VW == 1 ? (s & 1u) * 4u : 0u);
    const uint64_t blk_jstride = ((uint64_t)32u << (B - 6)) << (a.n1 - 2);  // words between rows j << (B-4) apart
    V c[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++)
        c[j] = blocked ? ldv<V>(a.src + (uint64_t)j * blk_jstride, blk_off) : ldv<V>(a.src + ((uint64_t)j << (B - 4)) * rowstride, ld_off);
    {
        uint32_t w1[15];
        load_round1_twiddles<B>(a.stage_tw, t, w1);
        for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl_i[i] = a.stage_tw[i];
        for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl_f[i] = a.stage_tw_fwd[i];
        if (!a.from_coeffs) dif_rounds<B, LQ>(c, tile, w1, twl_i, t, q);
    }
    // forward twiddle w^(k1 * m1), m1 = rev_B(position): the same for every coset
    const uint32_t c0 = two_level(a.twf_lo, a.twf_hi, a.twf_T, (uint64_t)k1 * rev_bits(t, B - 4));
    const uint32_t phi0 = two_level(a.twf_lo, a.twf_hi, a.twf_T, (uint64_t)k1 << (B - 4));
    const uint64_t kbase = (uint64_t)k1 + ((uint64_t)t << a.n1);
    const uint32_t cos0 = blockIdx.y * a.cos_per_block, ncos = cos0 + a.cos_per_block;  // this workgroup's cosets
    uint32_t sc_next = two_level(a.sc_lo[cos0], a.sc_hi[cos0], a.sc_T, kbase);
    // c[j] = coefficient k = k1 + N1 * k2, k2 = pt_of<B-4>(t, j) = (j << (B-4)) | t (what a coefficient matrix's rows already are)
    if (!a.from_coeffs) to_natural<B, LQ>(tile, c, t, q);
    {
        uint32_t pw2[16];
        if constexpr (!LEAN) power_ladder<16>(c0, phi0, pw2);
        for (uint32_t jc = cos0; jc < ncos; jc++) {
            const uint32_t sc = sc_next;
            if (jc + 1 < ncos) sc_next = two_level(a.sc_lo[jc + 1], a.sc_hi[jc + 1], a.sc_T, kbase);
            uint32_t w1[15];  // in flight while the scale ladder runs
            load_round1_twiddles<B>(a.stage_tw_fwd, t, w1);
            V v[16];
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) v[j] = c[j];
            scale_ladder<false>(v, sc, a.sc_phi[jc]);
            dif_rounds<B, LQ>(v, tile, w1, twl_f, t, q);
            uint32_t* o = a.dst + ((uint64_t)rev_bits(jc, a.added) << a.n) * a.W;  // position (t << 4) | j of the coset's block
            if (blocked) {
                // blocked positions straight from the final layout: position (t << 4) | j -> block ((t << 2) | (j >> 2), k1 group)
                const uint32_t fin_off = (((t << 2) << (a.n1 - 2)) + (k1 >> 2)) * 128u + (k1 & 3u) * 8u + (VW == 1 ? (s & 1u) * 4u : 0u);
                uint32_t pwl[16];
                if constexpr (LEAN) power_ladder<16>(c0, phi0, pwl);
#pragma unroll
                for (uint32_t j = 0; j < 16; j++)
                    stv<V>(o + ((uint64_t)(j >> 2) << (a.n1 - 2)) * 32u, fin_off + (j & 3u) * 32u, mul2(v[j], LEAN ? pwl[crev(j, 4)] : pw2[crev(j, 4)]));
                continue;
            }
            if constexpr (LEAN) {
                scale_ladder<true>(v, c0, phi0);
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) stv<V>(o + (uint64_t)j * rowstride, st_off, v[j]);
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) stv<V>(o + (uint64_t)j * rowstride, st_off, mul2(v[j], pw2[crev(j, 4)]));
            }
        }
    }
}

// K3: last forward digit on contiguous blocks of 2^B rows, in place.
template <int B, int LQ, int VW>
__global__ void __launch_bounds__(1 << (B - 4 + LQ)) narrow_fwd2_kernel(NarrowArgs a) {
    using namespace narrow;
    using V = typename Vec<VW>::T;
    constexpr uint32_t NQ = 1u << LQ, NTH = 1u << (B - 4 + LQ);
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr uint32_t NT = B >= 11 ? 1 : NARROW_EDGE_TILES;
    Tiles<V, (NT > 1)> tile{reinterpret_cast<V*>(smem), reinterpret_cast<V*>(smem) + (NT - 1) * (lds_rows(B) << LQ)};
    uint32_t* twl = smem + (NT * VW * lds_rows(B) << LQ);
    const uint32_t q = threadIdx.x & (NQ - 1), t = threadIdx.x >> LQ;
    const uint32_t tile0 = k3_tile_of_block(blockIdx.x, a.k3_pairs);
    const uint32_t s = tile0 * NQ + q;
    const uint32_t blk0 = slot_row(a, tile0 * NQ), blk = slot_row(a, s), cp = slot_col(a, s, blk);
    uint32_t* p = a.dst + ((uint64_t)blk0 << B) * a.W;                    // uniform: the workgroup's first block
    const uint32_t off = ((((blk - blk0) << B) + t) * a.W + VW * cp) * 4u;
    V v[16];
    if (a.blocked) {
        // W = 2: a group of four blocks of 2^B rows arrives as 128-byte blocks (k1 >> 2) of pairs (row block & 3, k1 & 3).  With
        // 32-byte tile rows the workgroup owns the whole group (in place: a.src == a.dst); with 16-byte tile rows (12-stage digits) it
        // owns two of the four and reads them from a.src, a buffer of its own.
        const uint32_t g0 = blk0 & ~3u;
        const uint32_t* ps = a.src + ((uint64_t)g0 << B) * a.W;
        const uint32_t blk_off = ((t >> 2) * 16u + (blk - g0) * 4u + (t & 3u)) * 8u + cp * 4u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = ldv<V>(ps + ((uint64_t)j << (B - 6)) * 32u, blk_off);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = ldv<V>(p + ((uint64_t)j << (B - 4)) * a.W, off);
    }
    uint32_t w1[15];
    load_round1_twiddles<B>(a.stage_tw, t, w1);
    for (uint32_t i = threadIdx.x; i + 1 < (1u << (B - 4)); i += NTH) twl[i] = a.stage_tw[i];
    dif_rounds<B, LQ>(v, tile, w1, twl, t, q);
    to_rows<B, LQ>(tile, v, t, q);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) stv<V>(p + ((uint64_t)j << (B - 4)) * a.W, off, v[j]);
}

}  // namespace p3
