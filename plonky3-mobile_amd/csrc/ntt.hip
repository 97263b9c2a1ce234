// Batched radix-2 NTT / coset LDE over BabyBear for gfx950.
//
// Replaces the reference's per-stage Vulkan dispatch loop (native/src/backend_vulkan.rs:1182-1294,
// kernels native/shaders/fft_stage.wgsl:75-136 and fft_stage_fused.wgsl:67-139), which costs log2(H)
// full HBM round trips.  Here a transform of 2^n rows is cut into 1-3 PASSES of b <= 9 stages (up to 11 supported); a
// workgroup stages a [2^b points] x [RUN independent words] tile in LDS (row stride RUN+1, so both
// the row-wise and the column-wise copies are bank-conflict free), runs the b stages as register
// radix-2^r rounds (r <= 5, one LDS round trip per round), and writes the tile back once.  Passes
// talk to HBM only in runs of RUN*4 = 64..128 contiguous bytes.
//
//   DIT plan (natural in -> natural out): the first pass gathers rows in bit-reversed order (runs of
//     RUN/W consecutive rows), later passes are in place with a pre-twiddle w_{2^(s0+b)}^(rev(pt)*lo).
//   DIF plan (natural in -> bit-reversed or natural out): top digit first, post-twiddle after the tile
//     transform; zero padding and the coset/1/N scaling are folded into the first pass's loads.
//   Narrow plan (ntt_narrow.hip.h, ntt_narrow_f64.hip.h): the coset LDE of matrices up to 16 columns wide (W = 2, 4, 6, 8, 16),
//     bit-reversed output, 2^16..2^24 rows — the fib_air trace / quotient / hiding commitments — as two digits per direction
//     in three launches (two when the input is coefficients); the same plan with 128-byte tile rows for matrices of any
//     width >= 64 at 2^16 rows (BASELINE configs[4]).  Integer or fp64 butterflies, chosen per shape from measurement.
//
// Row-major H x W matrices of Montgomery words, exactly the buffers the reference uploads
// (backend_vulkan.rs:2002-2005).  Stage semantics inside a tile are those of cpu_stage_u32_in_place
// (backend_vulkan.rs:881-942) with the same twiddle-table layout (:977-996: stage k at offset 2^k-1).
#include <algorithm>

#include "bb31.hip.h"
#include "common.h"

namespace p3 {

enum SideKind : uint32_t {
    SIDE_INPLACE = 0,   // word = ((hi << (s0+b)) + (pt << s0)) * W + f          (s0 > 0)
    SIDE_GROUP = 1,     // word = (((h0+t) << b) + pt) * W + c                   (contiguous groups)
    SIDE_GROUP_REV = 2, // word = ((rev(h0+t) << b) + pt) * W + c
    SIDE_STRIDED = 3,   // word = ((rev_b(pt) << (n-b)) + h0 + t) * W + c        (bit-reversal gather/scatter)
};

struct PassArgs {
    const uint32_t* src;
    uint32_t* dst;
    uint64_t src_rows;  // natural rows >= src_rows load as zero (zero padding)
    uint32_t W, wshift; // wshift = log2(W) if W is a power of two, else 0xffffffff
    uint32_t n, b, s0;
    uint32_t dif;       // 0: DIT tile (bit-reversed in, natural out); 1: DIF tile (natural in, bit-reversed out)
    uint32_t load_kind, store_kind;
    uint32_t G;         // groups per tile when W < RUN (group modes)
    uint32_t n_inner;   // inner tile count (column chunks / flattened runs)
    const uint32_t* tile_tw;
    uint32_t has_tw;    // pre (DIT) / post (DIF) twiddle w_{2^(s0+b)}^(rev_b(pt) * lo)
    const uint32_t* tw_lo;
    const uint32_t* tw_hi;
    uint32_t tw_T;
    uint32_t has_sc;    // multiply loaded value by sc(row)
    const uint32_t* sc_lo;
    const uint32_t* sc_hi;
    uint32_t sc_T;
    uint32_t has_us;    // multiply stored value by uscale
    uint32_t uscale;
    uint32_t sc_step;   // fast DIF kernel: shift^(rows between a lane's consecutive loads)
    uint32_t sc_base;   // the scale table's base (shift), host side only
    // fused middle kernel only: forward-direction tables and shift^(16 * 2^s0)
    const uint32_t* tile_tw2;
    const uint32_t* tw2_lo;
    const uint32_t* tw2_hi;
    uint32_t tw2_T;
    uint32_t sc_step16;
};

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t bits) {
    return bits ? (__brev(v) >> (32 - bits)) : 0u;
}

template <int LOG_R, int RR, bool DIF>
__device__ __forceinline__ void radix_round(uint32_t* tile, const uint32_t* twl, uint32_t stride, uint32_t b,
                                            uint32_t k0, uint32_t x, uint32_t g) {
    constexpr uint32_t R = 1u << LOG_R;
    constexpr uint32_t NSUB = 1u << (LOG_R - RR);
    constexpr uint32_t SUB = 1u << RR;
    const uint32_t lomask = (1u << k0) - 1u;
    const uint32_t gspan = (1u << b) >> LOG_R;  // number of thread groups g
    uint32_t v[R];
    uint32_t base[NSUB], olo[NSUB];
#pragma unroll
    for (uint32_t js = 0; js < NSUB; js++) {
        uint32_t o = js * gspan + g;
        olo[js] = o & lomask;
        base[js] = ((o >> k0) << (k0 + RR)) | olo[js];
#pragma unroll
        for (uint32_t ji = 0; ji < SUB; ji++) v[js * SUB + ji] = tile[(base[js] | (ji << k0)) * stride + x];
    }
#pragma unroll
    for (uint32_t js = 0; js < NSUB; js++) {
#pragma unroll
        for (int uu = 0; uu < RR; uu++) {
            const int u = DIF ? (RR - 1 - uu) : uu;
            const uint32_t k = k0 + u;
            const uint32_t tbase = (1u << k) - 1u + olo[js];
#pragma unroll
            for (uint32_t ji = 0; ji < SUB; ji++) {
                if (ji & (1u << u)) continue;
                const uint32_t j0 = js * SUB + ji, j1 = j0 | (1u << u);
                const uint32_t w = twl[tbase + ((ji & ((1u << u) - 1u)) << k0)];
                if (!DIF) {
                    uint32_t t = bb::mul(v[j1], w);
                    uint32_t a = v[j0];
                    v[j0] = bb::add(a, t);
                    v[j1] = bb::sub(a, t);
                } else {
                    uint32_t a = v[j0], c = v[j1];
                    v[j0] = bb::add(a, c);
                    v[j1] = bb::mul(bb::sub(a, c), w);
                }
            }
        }
    }
#pragma unroll
    for (uint32_t js = 0; js < NSUB; js++)
#pragma unroll
        for (uint32_t ji = 0; ji < SUB; ji++) tile[(base[js] | (ji << k0)) * stride + x] = v[js * SUB + ji];
}

template <int LOG_R, bool DIF>
__device__ __forceinline__ void radix_round_dispatch(uint32_t rr, uint32_t* tile, const uint32_t* twl,
                                                     uint32_t stride, uint32_t b, uint32_t k0, uint32_t x,
                                                     uint32_t g) {
    if constexpr (LOG_R >= 5) if (rr == 5) { radix_round<LOG_R, 5, DIF>(tile, twl, stride, b, k0, x, g); return; }
    if constexpr (LOG_R >= 4) if (rr == 4) { radix_round<LOG_R, 4, DIF>(tile, twl, stride, b, k0, x, g); return; }
    if constexpr (LOG_R >= 3) if (rr == 3) { radix_round<LOG_R, 3, DIF>(tile, twl, stride, b, k0, x, g); return; }
    if constexpr (LOG_R >= 2) if (rr == 2) { radix_round<LOG_R, 2, DIF>(tile, twl, stride, b, k0, x, g); return; }
    radix_round<LOG_R, 1, DIF>(tile, twl, stride, b, k0, x, g);
}

__device__ __forceinline__ uint32_t two_level(const uint32_t* lo, const uint32_t* hi, uint32_t T, uint64_t e) {
    uint32_t l = lo[(uint32_t)e & ((1u << T) - 1u)];
    uint32_t h = hi[(uint32_t)(e >> T)];
    return bb::mul(l, h);
}

}  // namespace p3
#include "ntt_fast.hip.h"
#include "ntt_narrow.hip.h"
#include "ntt_narrow_f64.hip.h"
namespace p3 {

// Decodes a copy-loop index into tile coordinates (pt, x), the global word offset and the natural row
// index for one side of the pass.  Returns false when the slot is padding (outside the matrix).
template <uint32_t RUN>
__device__ __forceinline__ bool decode_side(const PassArgs& a, uint32_t kind, uint32_t idx, uint32_t hi,
                                            uint32_t h0, uint32_t c0, uint64_t f0, uint32_t& pt, uint32_t& x,
                                            uint64_t& word, uint64_t& row, uint32_t& lo) {
    const uint32_t b = a.b;
    if (kind == SIDE_INPLACE) {
        x = idx & (RUN - 1);
        pt = idx >> __builtin_ctz(RUN);
        uint64_t f = f0 + x;
        uint64_t F = (uint64_t)a.W << a.s0;
        if (f >= F) return false;
        uint32_t c;
        if (a.wshift != 0xffffffffu) { lo = (uint32_t)(f >> a.wshift); c = (uint32_t)f & (a.W - 1); }
        else { lo = (uint32_t)(f / a.W); c = (uint32_t)(f - (uint64_t)lo * a.W); }
        (void)c;
        row = ((uint64_t)hi << (a.s0 + b)) + ((uint64_t)pt << a.s0) + lo;
        word = (((uint64_t)hi << (a.s0 + b)) + ((uint64_t)pt << a.s0)) * a.W + f;
        return true;
    }
    lo = 0;
    uint32_t t, c;
    const bool narrow = a.W < RUN;
    if (narrow && kind != SIDE_STRIDED) {
        // column-wise order: c fastest, then pt, then group t (contiguous global words within a group)
        uint32_t q;
        if (a.wshift != 0xffffffffu) { c = idx & (a.W - 1); q = idx >> a.wshift; }
        else { q = idx / a.W; c = idx - q * a.W; }
        pt = q & ((1u << b) - 1u);
        t = q >> b;
        if (t >= a.G) return false;
        x = t * a.W + c;
    } else {
        x = idx & (RUN - 1);
        pt = idx >> __builtin_ctz(RUN);
        if (narrow) {
            if (a.wshift != 0xffffffffu) { t = x >> a.wshift; c = x & (a.W - 1); }
            else { t = x / a.W; c = x - t * a.W; }
            if (t >= a.G) return false;
        } else {
            t = 0;
            c = c0 + x;
            if (c >= a.W) return false;
        }
    }
    const uint32_t gbits = a.n - b;
    const uint64_t ngroups = 1ull << gbits;
    uint64_t h = (uint64_t)h0 + t;
    if (h >= ngroups) return false;
    if (kind == SIDE_STRIDED) row = ((uint64_t)rev_bits(pt, b) << gbits) + h;
    else if (kind == SIDE_GROUP_REV) row = ((uint64_t)rev_bits((uint32_t)h, gbits) << b) + pt;
    else row = (h << b) + pt;
    word = row * a.W + c;
    return true;
}

// Row-wise sides (runs of RUN contiguous words per tile row): lane x is fixed per thread and the tile row
// advances linearly with the unrolled slot index j, so all address arithmetic is hoisted and the R loads of
// a lane are issued back to back (one latency per pass instead of R).
struct RowSide {
    uint32_t valid, lo;
    uint64_t base_word, base_row;  // word / row at pt = 0 (INPLACE, GROUP*) or the h part (STRIDED)
    uint32_t c;
};
template <uint32_t RUN>
__device__ __forceinline__ bool side_is_rowwise(const PassArgs& a, uint32_t kind) {
    return kind == SIDE_INPLACE || kind == SIDE_STRIDED || a.W >= RUN;
}
template <uint32_t RUN>
__device__ __forceinline__ RowSide rowside_init(const PassArgs& a, uint32_t kind, uint32_t x, uint32_t hi, uint32_t h0,
                                                uint32_t c0, uint64_t f0) {
    RowSide r{};
    const uint32_t b = a.b;
    if (kind == SIDE_INPLACE) {
        uint64_t f = f0 + x;
        r.valid = f < ((uint64_t)a.W << a.s0);
        r.lo = a.wshift != 0xffffffffu ? (uint32_t)(f >> a.wshift) : (uint32_t)(f / a.W);
        r.base_row = ((uint64_t)hi << (a.s0 + b)) + r.lo;
        r.base_word = ((uint64_t)hi << (a.s0 + b)) * a.W + f;
        return r;
    }
    uint32_t t = 0, c;
    if (a.W < RUN) {
        if (a.wshift != 0xffffffffu) { t = x >> a.wshift; c = x & (a.W - 1); }
        else { t = x / a.W; c = x - t * a.W; }
        r.valid = t < a.G;
    } else {
        c = c0 + x;
        r.valid = c < a.W;
    }
    const uint32_t gbits = a.n - b;
    uint64_t h = (uint64_t)h0 + t;
    if (h >= (1ull << gbits)) r.valid = 0;
    r.c = c;
    if (kind == SIDE_STRIDED) r.base_row = h;
    else r.base_row = (kind == SIDE_GROUP_REV ? (uint64_t)rev_bits((uint32_t)h, gbits) : h) << b;
    return r;
}
__device__ __forceinline__ void rowside_addr(const PassArgs& a, uint32_t kind, const RowSide& r, uint32_t pt,
                                             uint64_t& word, uint64_t& row) {
    if (kind == SIDE_INPLACE) {
        row = r.base_row + ((uint64_t)pt << a.s0);
        word = r.base_word + ((uint64_t)pt << a.s0) * a.W;
    } else if (kind == SIDE_STRIDED) {
        row = ((uint64_t)rev_bits(pt, a.b) << (a.n - a.b)) + r.base_row;
        word = row * a.W + r.c;
    } else {
        row = r.base_row + pt;
        word = row * a.W + r.c;
    }
}

// Column-wise sides (contiguous groups of a narrow power-of-two-width matrix): slot idx -> c fastest, then
// tile row pt, then group t, all by shifts; lanes walk contiguous global words of one group.
__device__ __forceinline__ bool colside_addr(const PassArgs& a, uint32_t kind, uint32_t idx, uint32_t h0, uint32_t& pt,
                                             uint32_t& x, uint64_t& word, uint64_t& row) {
    const uint32_t b = a.b, gbits = a.n - b;
    uint32_t c = idx & (a.W - 1), q = idx >> a.wshift;
    pt = q & ((1u << b) - 1u);
    uint32_t t = q >> b;
    x = (t << a.wshift) + c;
    uint64_t h = (uint64_t)h0 + t;
    if (t >= a.G || h >= (1ull << gbits)) return false;
    row = ((kind == SIDE_GROUP_REV ? (uint64_t)rev_bits((uint32_t)h, gbits) : h) << b) + pt;
    word = (row << a.wshift) + c;
    return true;
}

// Thread count is 2^b * RUN / R: at most 1024 for the wide-run instances, 256 for the RUN = 8 instance (which
// may then keep all 32 elements of a radix-32 round in registers without spilling).
// MODE fixes (load side, store side, direction, twiddle) at compile time so the five pass shapes the plans use
// shed every other path; MODE 0 keeps them as run-time values (rare shapes).
//   1 DIT first (gather, STRIDED -> GROUP_REV)   2 DIT later (INPLACE, pre-twiddle)
//   3 DIF non-last (INPLACE, post-twiddle)       4 DIF last, bit-reversed out (GROUP -> GROUP)
//   5 DIF last, natural out (GROUP_REV -> STRIDED)
template <int LOG_RUN, int LOG_R, int MODE>
__global__ void __launch_bounds__(1024) ntt_pass_kernel(PassArgs a) {
    const uint32_t load_kind = MODE == 0 ? a.load_kind : (MODE == 1 ? (uint32_t)SIDE_STRIDED : (MODE == 2 || MODE == 3) ? (uint32_t)SIDE_INPLACE : MODE == 4 ? (uint32_t)SIDE_GROUP : (uint32_t)SIDE_GROUP_REV);
    const uint32_t store_kind = MODE == 0 ? a.store_kind : (MODE == 1 ? (uint32_t)SIDE_GROUP_REV : (MODE == 2 || MODE == 3) ? (uint32_t)SIDE_INPLACE : MODE == 4 ? (uint32_t)SIDE_GROUP : (uint32_t)SIDE_STRIDED);
    const uint32_t dif = MODE == 0 ? a.dif : (MODE >= 3 ? 1u : 0u);
    const uint32_t has_tw = MODE == 0 ? a.has_tw : ((MODE == 2 || MODE == 3) ? 1u : 0u);
    constexpr uint32_t RUN = 1u << LOG_RUN, STRIDE = RUN + 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t b = a.b, npts = 1u << b;
    uint32_t* tile = smem;
    uint32_t* twl = smem + npts * STRIDE;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t bid = blockIdx.x;

    for (uint32_t i = tid; i + 1 < npts; i += nth) twl[i] = a.tile_tw[i];

    uint32_t hi = 0, h0 = 0, c0 = 0;
    uint64_t f0 = 0;
    if (a.s0 == 0) {
        if (a.W >= RUN) { h0 = bid / a.n_inner; c0 = (bid % a.n_inner) * RUN; }
        else h0 = bid * a.G;
    } else {
        hi = bid / a.n_inner;
        f0 = (uint64_t)(bid % a.n_inner) * RUN;
    }

    const uint32_t total = npts * RUN;
    constexpr uint32_t R = 1u << LOG_R;
    const bool full = nth == (total >> LOG_R);  // one slot per (thread, j): the unrolled fast paths apply
    const uint32_t xr = tid & (RUN - 1), pt0 = tid >> LOG_RUN, dpt = nth >> LOG_RUN;
    // ---- load ----
    if (full && side_is_rowwise<RUN>(a, load_kind)) {
        const RowSide rs = rowside_init<RUN>(a, load_kind, xr, hi, h0, c0, f0);
        uint32_t v[R];
#pragma unroll
        for (uint32_t j = 0; j < R; j++) {
            uint64_t word, row;
            rowside_addr(a, load_kind, rs, pt0 + j * dpt, word, row);
            v[j] = (rs.valid && row < a.src_rows) ? a.src[word] : 0u;
        }
        if (a.has_sc) {
#pragma unroll
            for (uint32_t j = 0; j < R; j++) {
                uint64_t word, row;
                rowside_addr(a, load_kind, rs, pt0 + j * dpt, word, row);
                if (rs.valid && row < a.src_rows) v[j] = bb::mul(v[j], two_level(a.sc_lo, a.sc_hi, a.sc_T, row));
            }
        }
        if (has_tw && !dif) {
#pragma unroll
            for (uint32_t j = 0; j < R; j++)
                v[j] = bb::mul(v[j], two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)rev_bits(pt0 + j * dpt, b) * rs.lo));
        }
#pragma unroll
        for (uint32_t j = 0; j < R; j++) tile[(pt0 + j * dpt) * STRIDE + xr] = v[j];
    } else if (full && a.wshift != 0xffffffffu) {
        // narrow power-of-two width, contiguous-group side (s0 == 0: no twiddle on this side)
        uint32_t v[R];
#pragma unroll
        for (uint32_t j = 0; j < R; j++) {
            uint32_t pt, x;
            uint64_t word, row;
            bool ok = colside_addr(a, load_kind, tid + j * nth, h0, pt, x, word, row);
            v[j] = (ok && row < a.src_rows) ? a.src[word] : 0u;
        }
        if (a.has_sc) {
#pragma unroll
            for (uint32_t j = 0; j < R; j++) {
                uint32_t pt, x;
                uint64_t word, row;
                if (colside_addr(a, load_kind, tid + j * nth, h0, pt, x, word, row) && row < a.src_rows)
                    v[j] = bb::mul(v[j], two_level(a.sc_lo, a.sc_hi, a.sc_T, row));
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < R; j++) {
            uint32_t pt, x;
            uint64_t word, row;
            if (colside_addr(a, load_kind, tid + j * nth, h0, pt, x, word, row)) tile[pt * STRIDE + x] = v[j];
        }
    } else {
        for (uint32_t idx = tid; idx < total; idx += nth) {
            uint32_t pt, x, lo;
            uint64_t word, row;
            bool ok = decode_side<RUN>(a, load_kind, idx, hi, h0, c0, f0, pt, x, word, row, lo);
            uint32_t v = 0;
            if (ok) {
                if (row < a.src_rows) {
                    v = a.src[word];
                    if (a.has_sc) v = bb::mul(v, two_level(a.sc_lo, a.sc_hi, a.sc_T, row));
                }
                if (has_tw && !dif)
                    v = bb::mul(v, two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)rev_bits(pt, b) * lo));
                tile[pt * STRIDE + x] = v;
            } else if (load_kind == SIDE_INPLACE || !(a.W < RUN && load_kind != SIDE_STRIDED)) {
                tile[pt * STRIDE + x] = 0;  // row-wise decode: (pt, x) are valid tile coordinates even for padding
            }
        }
    }
    __syncthreads();

    // ---- b stages as register-radix rounds ----
    const uint32_t nwork = total >> LOG_R;
    const uint32_t x = xr, g = pt0;
    if (!dif) {
        for (uint32_t k0 = 0; k0 < b;) {
            uint32_t rr = (b - k0) < (uint32_t)LOG_R ? (b - k0) : (uint32_t)LOG_R;
            if (tid < nwork) radix_round_dispatch<LOG_R, false>(rr, tile, twl, STRIDE, b, k0, x, g);
            __syncthreads();
            k0 += rr;
        }
    } else {
        for (uint32_t top = b; top > 0;) {
            uint32_t rr = top < (uint32_t)LOG_R ? top : (uint32_t)LOG_R;
            if (tid < nwork) radix_round_dispatch<LOG_R, true>(rr, tile, twl, STRIDE, b, top - rr, x, g);
            __syncthreads();
            top -= rr;
        }
    }

    // ---- store ----
    if (full && side_is_rowwise<RUN>(a, store_kind)) {
        const RowSide rs = rowside_init<RUN>(a, store_kind, xr, hi, h0, c0, f0);
        if (!rs.valid) return;
        uint32_t v[R];
#pragma unroll
        for (uint32_t j = 0; j < R; j++) v[j] = tile[(pt0 + j * dpt) * STRIDE + xr];
        if (has_tw && dif) {
#pragma unroll
            for (uint32_t j = 0; j < R; j++)
                v[j] = bb::mul(v[j], two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)rev_bits(pt0 + j * dpt, b) * rs.lo));
        }
        if (a.has_us) {
#pragma unroll
            for (uint32_t j = 0; j < R; j++) v[j] = bb::mul(v[j], a.uscale);
        }
#pragma unroll
        for (uint32_t j = 0; j < R; j++) {
            uint64_t word, row;
            rowside_addr(a, store_kind, rs, pt0 + j * dpt, word, row);
            a.dst[word] = v[j];
        }
        return;
    }
    if (full && a.wshift != 0xffffffffu) {
        uint32_t v[R];
#pragma unroll
        for (uint32_t j = 0; j < R; j++) {
            uint32_t pt, x;
            uint64_t word, row;
            bool ok = colside_addr(a, store_kind, tid + j * nth, h0, pt, x, word, row);
            v[j] = ok ? tile[pt * STRIDE + x] : 0u;
        }
        if (a.has_us) {
#pragma unroll
            for (uint32_t j = 0; j < R; j++) v[j] = bb::mul(v[j], a.uscale);
        }
#pragma unroll
        for (uint32_t j = 0; j < R; j++) {
            uint32_t pt, x;
            uint64_t word, row;
            if (colside_addr(a, store_kind, tid + j * nth, h0, pt, x, word, row)) a.dst[word] = v[j];
        }
        return;
    }
    for (uint32_t idx = tid; idx < total; idx += nth) {
        uint32_t pt, xx, lo;
        uint64_t word, row;
        if (!decode_side<RUN>(a, store_kind, idx, hi, h0, c0, f0, pt, xx, word, row, lo)) continue;
        uint32_t v = tile[pt * STRIDE + xx];
        if (has_tw && dif)
            v = bb::mul(v, two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)rev_bits(pt, b) * lo));
        if (a.has_us) v = bb::mul(v, a.uscale);
        a.dst[word] = v;
    }
}

__global__ void bit_reverse_rows_kernel(const uint32_t* src, uint32_t* dst, uint32_t log_h, uint32_t W,
                                        uint64_t total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint64_t r = i / W;
    uint32_t c = (uint32_t)(i - r * W);
    uint64_t sr = log_h ? (uint64_t)(__brevll(r) >> (64 - log_h)) : 0;
    dst[i] = src[sr * W + c];
}

int bit_reverse_rows(hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height, uint32_t width) {
    uint64_t total = height * width;
    if (!total) return OK;
    if (!is_pow2(height)) return fail(ERR_BAD_ARG, "bit_reverse_rows: height must be a power of two");
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    hipLaunchKernelGGL(bit_reverse_rows_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, log2u(height),
                       width, total);
    P3_HIP(hipGetLastError());
    return OK;
}

// ---------------------------------------------------------------------------------------------
// host-side planning
// ---------------------------------------------------------------------------------------------
namespace {

// Largest tile (log2 points per pass) of the general plans: 9, measured best on MI355X (2^20 / 2^21 rows -> three 7-stage passes).
constexpr uint32_t b_max() { return 9u; }

// Digits, lowest position digit first.
std::vector<uint32_t> split_digits(uint32_t n) {
    std::vector<uint32_t> d;
    if (n == 0) return d;
    const uint32_t B_MAX = b_max();
    uint32_t passes = (n + B_MAX - 1) / B_MAX;
    uint32_t rem = n;
    for (uint32_t i = 0; i < passes; i++) {
        uint32_t b = (rem + (passes - i) - 1) / (passes - i);
        d.push_back(b);
        rem -= b;
    }
    // The contiguous-group pass (lowest digit) has lean kernels up to 8 stages, the in-place passes up to 10:
    // when a digit exceeds 8 put the smallest digit lowest; otherwise the largest (measured faster for narrow
    // matrices: fewer, fatter group tiles).
    if (d.front() > 8) std::reverse(d.begin(), d.end());
    return d;
}

template <int LOG_RUN, int LOG_R, int MODE>
int launch_pass_m(Context& cx, hipStream_t stream, const PassArgs& a, uint32_t blocks) {
    constexpr uint32_t RUN = 1u << LOG_RUN;
    uint32_t npts = 1u << a.b;
    uint32_t threads = (npts * RUN) >> LOG_R;
    if (threads < 64) threads = 64;
    if (threads > 1024u) return fail(ERR_INTERNAL, "ntt: tile needs more threads than the kernel's launch bound");
    size_t lds = (size_t)npts * (RUN + 1) * 4 + (size_t)npts * 4;
    auto kern = ntt_pass_kernel<LOG_RUN, LOG_R, MODE>;
    { int rc = cx.ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, a);
    P3_HIP(hipGetLastError());
    return OK;
}

inline int pass_mode(const PassArgs& a) {
    if (!a.dif && a.load_kind == SIDE_STRIDED && a.store_kind == SIDE_GROUP_REV && !a.has_tw) return 1;
    if (!a.dif && a.load_kind == SIDE_INPLACE && a.store_kind == SIDE_INPLACE && a.has_tw) return 2;
    if (a.dif && a.load_kind == SIDE_INPLACE && a.store_kind == SIDE_INPLACE && a.has_tw) return 3;
    if (a.dif && a.load_kind == SIDE_GROUP && a.store_kind == SIDE_GROUP && !a.has_tw) return 4;
    if (a.dif && a.load_kind == SIDE_GROUP_REV && a.store_kind == SIDE_STRIDED && !a.has_tw) return 5;
    return 0;
}
// specialised shapes only for the tile geometries the plans actually pick; everything else runs MODE 0
template <int LOG_RUN, int LOG_R, bool FAST>
int launch_pass_t(Context& cx, hipStream_t stream, const PassArgs& a, uint32_t blocks) {
    if constexpr (FAST) {
        switch (pass_mode(a)) {
            case 1: return launch_pass_m<LOG_RUN, LOG_R, 1>(cx, stream, a, blocks);
            case 2: return launch_pass_m<LOG_RUN, LOG_R, 2>(cx, stream, a, blocks);
            case 3: return launch_pass_m<LOG_RUN, LOG_R, 3>(cx, stream, a, blocks);
            case 4: return launch_pass_m<LOG_RUN, LOG_R, 4>(cx, stream, a, blocks);
            case 5: return launch_pass_m<LOG_RUN, LOG_R, 5>(cx, stream, a, blocks);
            default: break;
        }
    }
    return launch_pass_m<LOG_RUN, LOG_R, 0>(cx, stream, a, blocks);
}

template <int B, int MODE>
int launch_fast_t(Context& cx, hipStream_t stream, const PassArgs& a, uint32_t blocks) {
    constexpr uint32_t NPTS = 1u << B, NR = B >= 9 ? 32 : 16;
    size_t lds = (size_t)NPTS * 33 * 4 + (size_t)NPTS * 4;
    if constexpr (B >= 9) {
        int rc = cx.ensure_dynamic_lds(reinterpret_cast<const void*>(ntt_fast_kernel<B, MODE>), 160 * 1024);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((ntt_fast_kernel<B, MODE>), dim3(blocks), dim3(NPTS * 32 / NR), lds, stream, a);
    P3_HIP(hipGetLastError());
    return OK;
}
template <int B, int MODE>
int launch_fast_group_t(hipStream_t stream, const PassArgs& a, uint32_t blocks) {
    constexpr uint32_t NPTS = 1u << B;
    size_t lds = (size_t)NPTS * 33 * 4 + (size_t)NPTS * 4;
    hipLaunchKernelGGL((ntt_fast_group_kernel<B, MODE>), dim3(blocks), dim3(NPTS * 2), lds, stream, a);
    P3_HIP(hipGetLastError());
    return OK;
}
int launch_fast(Context& cx, hipStream_t stream, const PassArgs& a, uint32_t blocks, int mode) {
    if (mode == 1) {
        switch (a.b) {
            case 6: return launch_fast_group_t<6, 1>(stream, a, blocks);
            case 7: return launch_fast_group_t<7, 1>(stream, a, blocks);
            default: return launch_fast_group_t<8, 1>(stream, a, blocks);
        }
    }
    if (mode == 4) {
        switch (a.b) {
            case 6: return launch_fast_group_t<6, 4>(stream, a, blocks);
            case 7: return launch_fast_group_t<7, 4>(stream, a, blocks);
            default: return launch_fast_group_t<8, 4>(stream, a, blocks);
        }
    }
    if (mode == 2) {
        switch (a.b) {
            case 6: return launch_fast_t<6, 2>(cx, stream, a, blocks);
            case 7: return launch_fast_t<7, 2>(cx, stream, a, blocks);
            case 8: return launch_fast_t<8, 2>(cx, stream, a, blocks);
            case 9: return launch_fast_t<9, 2>(cx, stream, a, blocks);
            default: return launch_fast_t<10, 2>(cx, stream, a, blocks);
        }
    }
    switch (a.b) {
        case 6: return launch_fast_t<6, 3>(cx, stream, a, blocks);
        case 7: return launch_fast_t<7, 3>(cx, stream, a, blocks);
        case 8: return launch_fast_t<8, 3>(cx, stream, a, blocks);
        case 9: return launch_fast_t<9, 3>(cx, stream, a, blocks);
        default: return launch_fast_t<10, 3>(cx, stream, a, blocks);
    }
}

int launch_pass(Context& cx, hipStream_t stream, PassArgs& a) {
    // geometry
    uint32_t log_run = a.b >= 11 ? 4 : 5;
    uint32_t RUN = 1u << log_run;
    a.wshift = is_pow2(a.W) ? log2u(a.W) : 0xffffffffu;
    uint64_t blocks;
    if (a.s0 == 0) {
        uint64_t ngroups = 1ull << (a.n - a.b);
        if (a.W >= RUN) {
            a.G = 1;
            a.n_inner = (a.W + RUN - 1) / RUN;
            blocks = ngroups * a.n_inner;
        } else {
            a.G = RUN / a.W;
            a.n_inner = 1;
            blocks = (ngroups + a.G - 1) / a.G;
        }
    } else {
        uint64_t F = (uint64_t)a.W << a.s0;
        a.G = 1;
        a.n_inner = (uint32_t)((F + RUN - 1) / RUN);
        blocks = (1ull << (a.n - a.s0 - a.b)) * a.n_inner;
    }
    if (blocks > 0x7fffffffull) return fail(ERR_BAD_ARG, "ntt: matrix too large for one launch");
    uint32_t nb = (uint32_t)blocks;
    if (log_run == 5 && a.b >= 6 && a.b <= 10) {
        int mode = pass_mode(a);
        if (mode == 2 || mode == 3) {
            const uint32_t rows_per_lane = a.b >= 9 ? 32 : 16;
            if (mode == 3 && a.has_sc) a.sc_step = bb::pow(a.sc_base, (uint64_t)((1u << a.b) / rows_per_lane) << a.s0);
            return launch_fast(cx, stream, a, nb, mode);
        }
        if (a.b > 8) goto general;
        // group-side passes: narrow widths must be powers of two for the shift-based column-wise copy
        if ((mode == 1 || (mode == 4 && !a.has_sc)) && (a.W >= 32 || a.wshift != 0xffffffffu)) return launch_fast(cx, stream, a, nb, mode);
    }
general:
    if (log_run == 4) return launch_pass_t<4, 5, false>(cx, stream, a, nb);
    uint32_t log_r = a.b >= 10 ? 5 : (a.b >= 4 ? 4 : a.b);
    switch (log_r) {
        case 5: return launch_pass_t<5, 5, false>(cx, stream, a, nb);
        case 4: return launch_pass_t<5, 4, true>(cx, stream, a, nb);
        case 3: return launch_pass_t<5, 3, false>(cx, stream, a, nb);
        case 2: return launch_pass_t<5, 2, false>(cx, stream, a, nb);
        default: return launch_pass_t<5, 1, false>(cx, stream, a, nb);
    }
}

int set_twiddle(Context& cx, hipStream_t stream, PassArgs& a, bool inverse) {
    a.has_tw = a.s0 > 0;
    if (!a.has_tw) return OK;
    TwoLevelTable t;
    int rc = cx.get_root_table(stream, a.s0 + a.b, inverse, &t);
    if (rc) return rc;
    a.tw_lo = t.lo; a.tw_hi = t.hi; a.tw_T = t.T;
    return OK;
}

// DIT plan: natural in -> natural out.  uscale (if has_us) is applied by the last pass.
int run_dit(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint32_t n, uint32_t W,
            bool inverse, bool has_us, uint32_t uscale) {
    std::vector<uint32_t> digits = split_digits(n);
    uint32_t s0 = 0;
    for (size_t i = 0; i < digits.size(); i++) {
        PassArgs a{};
        a.W = W; a.n = n; a.b = digits[i]; a.s0 = s0; a.dif = 0;
        a.tile_tw = cx.tile_tw[inverse ? 1 : 0];
        a.src_rows = 1ull << n;
        if (i == 0) { a.src = src; a.dst = dst; a.load_kind = SIDE_STRIDED; a.store_kind = SIDE_GROUP_REV; }
        else { a.src = dst; a.dst = dst; a.load_kind = SIDE_INPLACE; a.store_kind = SIDE_INPLACE; }
        int rc = set_twiddle(cx, stream, a, inverse);
        if (rc) return rc;
        if (i + 1 == digits.size() && has_us) { a.has_us = 1; a.uscale = uscale; }
        rc = launch_pass(cx, stream, a);
        if (rc) return rc;
        s0 += digits[i];
    }
    return OK;
}

// DIF plan: natural in (first src_rows rows, zero padded to 2^n, optionally scaled per row) ->
// bit-reversed (in place layout) or natural (scattered by the last pass) out.
int run_dif(Context& cx, hipStream_t stream, const uint32_t* src, uint64_t src_rows, uint32_t* dst, uint32_t n,
            uint32_t W, bool inverse, const TwoLevelTable* sc, uint32_t sc_base, bool natural_out) {
    std::vector<uint32_t> digits = split_digits(n);  // lowest first; DIF walks them top-down
    // The natural-order last pass SCATTERS rows (group h0 writes rows that belong to other groups' inputs), so it
    // cannot run in place: with more than one pass the earlier passes work in scratch and the last one goes
    // scratch -> dst.  (Bit-reversed output stays in place: every pass rewrites exactly the rows it read.)
    uint32_t* work = dst;
    if (natural_out && digits.size() > 1) {
        int rc = cx.ws(stream, 0).reserve(((size_t)W << n) * 4);
        if (rc) return rc;
        work = cx.ws(stream, 0).as<uint32_t>();
    }
    uint32_t s0 = n;
    for (size_t ii = digits.size(); ii-- > 0;) {
        bool first = ii + 1 == digits.size(), last = ii == 0;
        s0 -= digits[ii];
        PassArgs a{};
        a.W = W; a.n = n; a.b = digits[ii]; a.s0 = s0; a.dif = 1;
        a.tile_tw = cx.tile_tw[inverse ? 1 : 0];
        a.src = first ? src : work;
        a.dst = last ? dst : work;
        a.src_rows = first ? src_rows : (1ull << n);
        if (first && sc) { a.has_sc = 1; a.sc_lo = sc->lo; a.sc_hi = sc->hi; a.sc_T = sc->T; a.sc_base = sc_base; }
        if (!last) { a.load_kind = SIDE_INPLACE; a.store_kind = SIDE_INPLACE; }
        else if (natural_out) { a.load_kind = SIDE_GROUP_REV; a.store_kind = SIDE_STRIDED; }
        else { a.load_kind = SIDE_GROUP; a.store_kind = SIDE_GROUP; }
        int rc = set_twiddle(cx, stream, a, inverse);
        if (rc) return rc;
        rc = launch_pass(cx, stream, a);
        if (rc) return rc;
    }
    return OK;
}

template <int BI, int A>
int launch_fused_mid_t(hipStream_t stream, const PassArgs& a, uint32_t blocks) {
    constexpr uint32_t NF = 1u << (BI + A);
    size_t lds = (size_t)NF * 33 * 4 + (size_t)NF * 8;
    hipLaunchKernelGGL((ntt_fused_mid_kernel<BI, A>), dim3(blocks), dim3(NF * 2), lds, stream, a);
    P3_HIP(hipGetLastError());
    return OK;
}

// coset LDE with the middle passes fused.  Returns 1 when the shape is not covered.
int lde_fused(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint32_t n, uint32_t added, uint32_t W,
              uint32_t shift, bool bit_reversed_out) {
    if (!bit_reversed_out || added < 1 || added > 2) return 1;
    const uint32_t m = n + added;
    // forward digits (lowest first) e1, e2, e3 with e3 = top; inverse digits e1, e2, e3 - added
    std::vector<uint32_t> fd = split_digits(m);
    if (fd.size() != 3) return 1;
    uint32_t e1 = fd[0], e2 = fd[1], e3 = fd[2];
    // put the largest digit on top so that e3 - added stays >= 5
    if (e1 > e3) std::swap(e1, e3);
    if (e2 > e3) std::swap(e2, e3);
    if (e3 > 8 || e3 < added + 5 || e1 < 6 || e2 < 6 || e1 > 8 || e2 > 8) return 1;
    const uint32_t bi = e3 - added, s0 = e1 + e2;
    if (bi < 5 || (W < 32 && !is_pow2(W))) return 1;
    if (bi < 6) return 1;  // kernels instantiated for BI in {6, 7}
    const uint64_t N = 1ull << n;
    size_t bytes = N * W * 4;
    int rc = cx.ws(stream, 1).reserve(bytes);
    if (rc) return rc;
    uint32_t* coeffs = cx.ws(stream, 1).as<uint32_t>();
    // inverse passes 1 and 2 (digits e1, e2) of the DIT plan
    {
        uint32_t digits[2] = {e1, e2};
        uint32_t s = 0;
        for (int i = 0; i < 2; i++) {
            PassArgs a{};
            a.W = W; a.n = n; a.b = digits[i]; a.s0 = s; a.dif = 0;
            a.tile_tw = cx.tile_tw[1];
            a.src_rows = N;
            if (i == 0) { a.src = src; a.dst = coeffs; a.load_kind = SIDE_STRIDED; a.store_kind = SIDE_GROUP_REV; }
            else { a.src = coeffs; a.dst = coeffs; a.load_kind = SIDE_INPLACE; a.store_kind = SIDE_INPLACE; }
            rc = set_twiddle(cx, stream, a, true);
            if (rc) return rc;
            rc = launch_pass(cx, stream, a);
            if (rc) return rc;
            s += digits[i];
        }
    }
    // fused middle: inverse top digit bi + scale + zero-extension + forward top digit e3
    {
        PassArgs a{};
        a.W = W; a.n = n; a.b = bi; a.s0 = s0;
        a.wshift = is_pow2(W) ? log2u(W) : 0xffffffffu;
        a.src = coeffs; a.dst = dst;
        a.tile_tw = cx.tile_tw[1];
        a.tile_tw2 = cx.tile_tw[0];
        TwoLevelTable ti, tf, sc;
        if ((rc = cx.get_root_table(stream, s0 + bi, true, &ti))) return rc;
        if ((rc = cx.get_root_table(stream, s0 + e3, false, &tf))) return rc;
        uint32_t hinv = bb::inv(bb::to_monty((uint32_t)N));
        if ((rc = cx.get_scale_table(stream, shift, n, hinv, &sc))) return rc;
        a.tw_lo = ti.lo; a.tw_hi = ti.hi; a.tw_T = ti.T;
        a.tw2_lo = tf.lo; a.tw2_hi = tf.hi; a.tw2_T = tf.T;
        a.sc_lo = sc.lo; a.sc_hi = sc.hi; a.sc_T = sc.T;
        a.sc_step = bb::pow(shift, 1ull << s0);
        a.sc_step16 = bb::pow(a.sc_step, 16);
        uint64_t F = (uint64_t)W << s0;
        uint32_t blocks = (uint32_t)((F + 31) / 32);
        if (bi == 6 && added == 1) rc = launch_fused_mid_t<6, 1>(stream, a, blocks);
        else if (bi == 7 && added == 1) rc = launch_fused_mid_t<7, 1>(stream, a, blocks);
        else if (bi == 6 && added == 2) rc = launch_fused_mid_t<6, 2>(stream, a, blocks);
        else return fail(ERR_INTERNAL, "lde_fused: unexpected digit shape");
        if (rc) return rc;
    }
    // forward passes for digits e2 (s0 = e1) and e1 (s0 = 0), in place on dst
    {
        PassArgs a{};
        a.W = W; a.n = m; a.b = e2; a.s0 = e1; a.dif = 1;
        a.tile_tw = cx.tile_tw[0];
        a.src = dst; a.dst = dst; a.src_rows = 1ull << m;
        a.load_kind = SIDE_INPLACE; a.store_kind = SIDE_INPLACE;
        if ((rc = set_twiddle(cx, stream, a, false))) return rc;
        if ((rc = launch_pass(cx, stream, a))) return rc;
        PassArgs b{};
        b.W = W; b.n = m; b.b = e1; b.s0 = 0; b.dif = 1;
        b.tile_tw = cx.tile_tw[0];
        b.src = dst; b.dst = dst; b.src_rows = 1ull << m;
        b.load_kind = SIDE_GROUP; b.store_kind = SIDE_GROUP;
        if ((rc = set_twiddle(cx, stream, b, false))) return rc;
        if ((rc = launch_pass(cx, stream, b))) return rc;
    }
    return OK;
}


template <int B, int LQ, int VW, int K>
int launch_narrow_t(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t blocks, uint32_t grid_y = 1) {
    // padded tile (17 rows per 16 points; the middle kernel alternates two below 1024 threads) + stage-table prefixes
    constexpr size_t tile_bytes = ((size_t)4 * VW * narrow::lds_rows(B)) << LQ;
    constexpr size_t n_tiles = K == 2 ? (B - 4 + LQ >= 10 ? 1 : NARROW_MID_TILES) : (B >= 11 ? 1 : NARROW_EDGE_TILES);
    constexpr size_t lds = tile_bytes * n_tiles + ((size_t)4 << (B - 4)) * (K == 2 ? 2 : 1);
    static_assert(lds <= 160 * 1024, "narrow tile does not fit the LDS");
    void (*kern)(NarrowArgs);
    if constexpr (K == 1) kern = narrow_inv1_kernel<B, LQ, VW>;
    else if constexpr (K == 2) kern = narrow_mid_kernel<B, LQ, VW>;
    else kern = narrow_fwd2_kernel<B, LQ, VW>;
    if constexpr (lds > 64 * 1024) {
        int rc = cx.ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(blocks, K == 2 ? grid_y : 1u), dim3(1u << (B - 4 + LQ)), lds, stream, a);
    P3_HIP(hipGetLastError());
    return OK;
}
// Geometry per digit size: column pairs per lane (VW = 2) or single columns (VW = 1: twice the waves).
// Tile rows are 32 bytes (LQ = 2 for pairs, 3 for single words), 16 bytes for 12-stage digits (64 KB tiles, <= 1024 threads).
constexpr int narrow_lq(int b, int vw) { return (b == 12 ? 1 : 2) + (vw == 1 ? 1 : 0); }
// INTEGER butterflies: digits of 10..12 stages, i.e. heights from 2^20 on, where they win (profiles/r04_lde_f64_vs_int.txt); single
// columns only where lde_narrow picks them (the 2^20 middle kernel).  Smaller digits run the fp64 kernels below; round 5 retired
// the integer instantiations for 8- and 9-stage digits of narrow matrices (3-8 % slower there, profiles/r03_lde_f64_vs_int.txt).
template <int K, int VW>
int launch_narrow_v(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t b, uint32_t blocks, uint32_t gy) {
    if (b == 10) return launch_narrow_t<10, narrow_lq(10, VW), VW, K>(cx, stream, a, blocks, gy);
    if constexpr (VW == 2) {
        if (b == 11) return launch_narrow_t<11, narrow_lq(11, 2), 2, K>(cx, stream, a, blocks, gy);
        if (b == 12) return launch_narrow_t<12, narrow_lq(12, 2), 2, K>(cx, stream, a, blocks, gy);
    }
    return fail(ERR_INTERNAL, "lde_narrow: no integer kernel for this digit / lane vector");
}
template <int K>
int launch_narrow(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t b, uint32_t blocks, int vw, uint32_t gy = 1) {
    return vw == 1 ? launch_narrow_v<K, 1>(cx, stream, a, b, blocks, gy) : launch_narrow_v<K, 2>(cx, stream, a, b, blocks, gy);
}

// fp64 forms (ntt_narrow_f64.hip.h): digits of 8..10 stages (heights up to 2^19, and 2^20 x 2).  NT = 2 LDS tiles (one barrier per
// hand-over) where they fit and the grid gives a CU one workgroup anyway; otherwise one tile, so that two workgroups can share a CU.
template <int B, int LQ, int VW, int K, int NT>
int launch_narrow64_nt(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t blocks, uint32_t grid_y) {
    constexpr size_t lds = narrow64::lds_bytes<B, LQ, VW, NT>(K == 2 ? 2 : 1);
    static_assert(lds <= 160 * 1024, "fp64 narrow tile does not fit the LDS");
    void (*kern)(NarrowArgs);
    if constexpr (K == 1) kern = narrow64_inv1_kernel<B, LQ, VW, NT>;
    else if constexpr (K == 2) kern = narrow64_mid_kernel<B, LQ, VW, NT, (VW == 2)>;
    else kern = narrow64_fwd2_kernel<B, LQ, VW, NT>;
    if constexpr (lds > 64 * 1024) {
        int rc = cx.ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(blocks, K == 2 ? grid_y : 1u), dim3(1u << (B - 4 + LQ)), lds, stream, a);
    P3_HIP(hipGetLastError());
    return OK;
}
template <int B, int LQ, int VW, int K>
int launch_narrow64_t(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t blocks, uint32_t grid_y) {
    constexpr bool fits2 = narrow64::lds_bytes<B, LQ, VW, 2>(K == 2 ? 2 : 1) <= 160 * 1024;
    if constexpr (fits2) {
        const bool two = (size_t)blocks * grid_y <= 256 || narrow64::lds_bytes<B, LQ, VW, 2>(K == 2 ? 2 : 1) <= 80 * 1024;
        if (two) return launch_narrow64_nt<B, LQ, VW, K, 2>(cx, stream, a, blocks, grid_y);
    }
    return launch_narrow64_nt<B, LQ, VW, K, 1>(cx, stream, a, blocks, grid_y);
}
// 1024-thread workgroups (128 VGPRs per lane) are left to the integer kernels: K2 at single columns from 11 stages on
constexpr bool narrow64_has(int b, int vw, int k) { return b <= 10 && b - 4 + narrow_lq(b, vw) <= (k == 2 ? 9 : 10); }
template <int K, int VW>
int launch_narrow64_v(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t b, uint32_t blocks, uint32_t gy) {
#define P3_N64_CASE(BB)                                                                                          \
    case BB:                                                                                                     \
        if constexpr (narrow64_has(BB, VW, K)) return launch_narrow64_t<BB, narrow_lq(BB, VW), VW, K>(cx, stream, a, blocks, gy); \
        else return fail(ERR_INTERNAL, "lde_narrow: fp64 shape not instantiated");
    switch (b) {
        P3_N64_CASE(8) P3_N64_CASE(9) P3_N64_CASE(10)
        default: return fail(ERR_INTERNAL, "lde_narrow: digit out of range");
    }
#undef P3_N64_CASE
}
template <int K>
int launch_narrow64(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t b, uint32_t blocks, int vw, uint32_t gy = 1) {
    return vw == 1 ? launch_narrow64_v<K, 1>(cx, stream, a, b, blocks, gy) : launch_narrow64_v<K, 2>(cx, stream, a, b, blocks, gy);
}

// Narrow-matrix coset LDE in three launches (ntt_narrow.hip.h).  Returns 1 when the shape is not covered.
// WIDE matrices (more than 16 columns, any width incl. odd ones) on the same two-digit plan: single columns per lane and tiles
// of 2^8 rows x 32 words, i.e. 128-byte row segments (the slots of a row group are its words in memory order, so a tile may
// straddle rows).  Instantiated for 8- and 9-stage digits: 2^16 rows (BASELINE configs[4], 2^16 x 2633), 2^17 and 2^18 rows
// (a 10-stage digit would need 2048 threads at this tile width).
// Integer butterflies (2^16 x 2633: 2143 us against 2287 us with the fp64 form, which round 5 retired for this shape).
template <int K>
int launch_narrow_wide(Context& cx, hipStream_t stream, const NarrowArgs& a, uint32_t b, uint32_t blocks, uint32_t gy) {
    if (b == 9) return launch_narrow_t<9, 5, 1, K>(cx, stream, a, blocks, gy);  // 2^17 / 2^18 rows: 1024-thread tiles
    if (b != 8) return fail(ERR_INTERNAL, "lde_narrow: wide digit out of range");
    return launch_narrow_t<8, 5, 1, K>(cx, stream, a, blocks, gy);
}

// from_coeffs: src holds the COEFFICIENTS of the columns (natural order, 2^n rows) instead of evaluations over the subgroup:
// K1 and the inverse half of K2 are skipped (two launches; the hiding prover's blinded quotient chunks arrive that way).
int lde_narrow(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint32_t n, uint32_t added, uint32_t W,
               uint32_t shift, bool bit_reversed_out, bool from_coeffs = false) {
    if (!bit_reversed_out || added < 1 || added > 3) return 1;
    // 32-byte row segments per tile: faster than the general plans up to W = 16 (1.3-2.1x), level from W = 32 on
    // W = 6 (the hiding prover's randomized trace, fib_air.rs:65: 2 columns + 4 random codewords): three column pairs per row
    constexpr uint32_t w_max = 16;
    // 128-byte tile rows, single columns: any width whose byte offsets stay below 2^32 (the kernels index in u32: K1's
    // transposed store reaches rows * W * 4 bytes of the coefficient matrix, K2 / K3 the LDE's (rows << added) * W * 4)
    const bool wide = W > w_max && n >= 16 && n <= 18 && W >= 64 && (((uint64_t)W << (n + added + 2)) < (1ull << 32));
    if (!wide && (W < 2 || W > w_max || !(is_pow2(W) || W == 6))) return 1;
    if (n < 16 || n > 24) return 1;
    if (!wide && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 7u)) return 1;  // 8-byte accesses
    const uint32_t n1 = (n + 1) / 2, n2 = n - n1;
    const uint64_t N = 1ull << n;
    // lane vector: column pairs, or single columns where that doubles a thin grid's waves per SIMD
    int vw[3];  // per kernel
    // measured (tools/lde_sweep.py): single columns win by 15-25 % up to 2^19 rows (1-2 waves per SIMD otherwise), only
    // for the middle kernel at 2^20 (26.5 -> 23.7 us), and lose from 2^21 on (the grid is full; twice the twiddle work)
    for (int k = 0; k < 3; k++) vw[k] = (n <= 19 || (n == 20 && k == 1)) ? 1 : 2;
    if (wide) vw[0] = vw[1] = vw[2] = 1;
    auto lq_of = [&](int k, uint32_t b) -> uint32_t { return wide ? 5u : (uint32_t)narrow_lq((int)b, vw[k]); };
    int rc = cx.ws(stream, 1).reserve(N * W * 4);
    if (rc) return rc;
    uint32_t* T = cx.ws(stream, 1).as<uint32_t>();
    NarrowArgs a{};
    // blocked intermediates (W = 2) need tiles of 4 rows in all three kernels (32-byte tile rows: digits below 12 stages); 12-stage
    // digits have 2-row tiles: K3 then owns half of every 128-byte block and can no longer run in place, so K2 writes the blocked
    // intermediate to a scratch of its own
    a.blocked = W == 2 && !from_coeffs;
    bool k3_out_of_place = a.blocked && n1 >= 12;
    uint32_t* mid = dst;  // K2's output = K3's input
    if (k3_out_of_place) {
        // slot 4 belongs to this function alone: slots 2 / 3 are the host-pointer entry points' staging buffers (c_api.hip
        // dft_host), i.e. possible `src` / `dst` of this very call
        const size_t mid_bytes = (N << added) * W * 4;
        if ((rc = cx.ws(stream, 4).reserve(mid_bytes))) return rc;
        mid = cx.ws(stream, 4).as<uint32_t>();
        // K3 must never read the buffer its partner tiles write: if a caller's dst overlaps the scratch all the same, take
        // the row-major intermediates (K3 in place, no half-owned blocks)
        const uintptr_t m0 = reinterpret_cast<uintptr_t>(mid), d0 = reinterpret_cast<uintptr_t>(dst);
        if (m0 < d0 + mid_bytes && d0 < m0 + mid_bytes) { a.blocked = 0; k3_out_of_place = false; mid = dst; }
    }
    a.from_coeffs = from_coeffs;
    a.n = n; a.n1 = n1; a.n2 = n2; a.W = W; a.added = added;
    TwoLevelTable ti, tf;
    if ((rc = cx.get_root_table(stream, n, true, &ti))) return rc;
    if ((rc = cx.get_root_table(stream, n, false, &tf))) return rc;
    auto geometry = [&](int k, uint32_t b, uint64_t rows) -> uint32_t {  // sets wsl / xcd_remap, returns the tile count
        const uint32_t slots_per_row = W / vw[k];
        a.spr = slots_per_row;
        a.wsl = is_pow2(slots_per_row) ? log2u(slots_per_row) : 0xffffffffu;
        const uint32_t lq = lq_of(k, b);
        const uint32_t tiles = (uint32_t)((rows * slots_per_row) >> lq);
        const uint32_t group = 8u << (wide ? 0 : (vw[k] == 2 ? 4 : 5) - lq);  // tiles per 128-byte line x 8 XCDs
        a.xcd_remap = !wide && k != 2 && tiles % group == 0 && is_pow2(slots_per_row);
        if (wide && tiles % 128 == 0) a.xcd_remap = 4;  // 16 consecutive tiles (2 KB of a row) per XCD
        return tiles;
    };
    // K1
    a.src = src; a.dst = T;
    a.stage_tw = cx.tile_tw[1];
    a.tw_lo = ti.lo; a.tw_hi = ti.hi; a.tw_T = ti.T;
    // Arithmetic per shape, from measurement (profiles/r03_lde_f64_vs_int.txt, r04_lde_f64_vs_int.txt): fp64 butterflies (6
    // instructions instead of 10, ntt_narrow_f64.hip.h) gain 3-8 % up to 2^19 rows and at 2^20 x 2, and LOSE from 2^21 rows on,
    // where twice the LDS per workgroup halves the waves per SIMD; wide matrices: 2143 us integer, 2287 us fp64.
    const bool f64 = !wide && (n <= 19 || (n == 20 && W == 2));
    a.stage_twd = cx.tile_twd[1];
    a.neg_pm1 = -2013265920.0; a.pinv = 1.0 / 2013265921.0; a.fbias = -0.5 + 1.0 / 8589934592.0;
    uint32_t tiles = geometry(0, n1, 1ull << n2);
    if (!from_coeffs)
        if ((rc = wide ? launch_narrow_wide<1>(cx, stream, a, n1, tiles, 1)
                       : f64 ? launch_narrow64<1>(cx, stream, a, n1, tiles, vw[0]) : launch_narrow<1>(cx, stream, a, n1, tiles, vw[0]))) return rc;
    // K2
    a.src = from_coeffs ? src : T; a.dst = mid;
    a.stage_tw = cx.tile_tw[1]; a.stage_tw_fwd = cx.tile_tw[0];
    a.stage_twd = cx.tile_twd[1]; a.stage_twd_fwd = cx.tile_twd[0];
    a.twf_lo = tf.lo; a.twf_hi = tf.hi; a.twf_T = tf.T;
    const uint32_t hinv = from_coeffs ? bb::ONE : bb::inv(bb::to_monty((uint32_t)N));  // the 1 / N of the inverse transform
    const uint32_t g = bb::two_adic_generator(n + added);
    uint32_t base = shift;
    for (uint32_t j = 0; j < (1u << added); j++) {
        TwoLevelTable sc;
        if ((rc = cx.get_scale_table(stream, base, n, hinv, &sc))) return rc;
        a.sc_lo[j] = sc.lo; a.sc_hi[j] = sc.hi; a.sc_T = sc.T;
        a.sc_phi[j] = bb::pow(base, 1ull << (n1 + n2 - 4));
        base = bb::mul(base, g);
    }
    tiles = geometry(1, n2, 1ull << n1);
    // Grids that leave CUs idle (fewer workgroups than CUs): split the cosets over workgroups.  Every workgroup repeats
    // the inverse digit (more arithmetic), but idle CUs take the extra workgroups: 2^16..2^18 rows gain 5-25 %
    // (profiles/r02_lde_sweep_cosplit.txt); from one workgroup per CU on the kernel is bound by its
    // arithmetic and a split only adds to it (2^20 x 2: 52.4 -> 55.7 us), so the split stops at 256 workgroups.
    uint32_t split_log = 0;
    while (split_log < added && ((uint64_t)tiles << (split_log + 1)) <= 256) split_log++;
    a.cos_per_block = (1u << added) >> split_log;
    if ((rc = wide ? launch_narrow_wide<2>(cx, stream, a, n2, tiles, 1u << split_log)
                   : f64 ? launch_narrow64<2>(cx, stream, a, n2, tiles, vw[1], 1u << split_log)
                         : launch_narrow<2>(cx, stream, a, n2, tiles, vw[1], 1u << split_log))) return rc;
    // K3
    a.src = mid; a.dst = dst;
    a.stage_tw = cx.tile_tw[0];
    a.stage_twd = cx.tile_twd[0];
    tiles = geometry(2, n1, (1ull << added) << n2);
    a.k3_pairs = k3_out_of_place && vw[2] == 2 && tiles % 16 == 0 ? 1u : 0u;
    if (wide && tiles % 128 == 0) a.k3_pairs = 4;
    if (wide) return launch_narrow_wide<3>(cx, stream, a, n1, tiles, 1);
    return f64 ? launch_narrow64<3>(cx, stream, a, n1, tiles, vw[2]) : launch_narrow<3>(cx, stream, a, n1, tiles, vw[2]);
}

}  // namespace

// The passes dft_batch / idft_batch run for 2^n rows: one kernel launch per entry, entry = radix-2 stages done by that pass, in
// launch order (run_dit walks the digits lowest position first).  Host-only.
std::vector<uint32_t> ntt_dft_plan(uint32_t n) { return split_digits(n); }

int ntt_dft(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height,
            uint32_t width, bool inverse) {
    if (!height || !width) return OK;
    if (!is_pow2(height)) return fail(ERR_BAD_ARG, "hip backend requires power-of-two height, got " + std::to_string(height));
    uint32_t n = log2u(height);
    if (n > bb::TWO_ADICITY) return fail(ERR_BAD_ARG, "height exceeds BabyBear two-adicity");
    size_t bytes = height * width * 4;
    if (n == 0) {
        if (src != dst) P3_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
        return OK;
    }
    if (src == dst) {  // the gathering first pass is out of place
        int rc = cx.ws(stream, 0).reserve(bytes);
        if (rc) return rc;
        P3_HIP(hipMemcpyAsync(cx.ws(stream, 0).ptr, src, bytes, hipMemcpyDeviceToDevice, stream));
        src = cx.ws(stream, 0).as<uint32_t>();
    }
    uint32_t hinv = inverse ? bb::inv(bb::to_monty((uint32_t)height)) : 0;
    return run_dit(cx, stream, src, dst, n, width, inverse, inverse, hinv);
}

int ntt_coset_dft(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height,
                  uint32_t width, uint32_t shift) {
    if (!height || !width) return OK;
    if (!is_pow2(height)) return fail(ERR_BAD_ARG, "hip backend requires power-of-two height, got " + std::to_string(height));
    uint32_t n = log2u(height);
    if (n > bb::TWO_ADICITY) return fail(ERR_BAD_ARG, "height exceeds BabyBear two-adicity");
    TwoLevelTable sc;
    int rc = cx.reserve_scale_slots(1);
    if (rc) return rc;
    rc = cx.get_scale_table(stream, shift, n, bb::ONE, &sc);
    if (rc) return rc;
    if (n == 0) {  // single row: scale by shift^0 = 1
        if (src != dst) P3_HIP(hipMemcpyAsync(dst, src, height * width * 4, hipMemcpyDeviceToDevice, stream));
        return OK;
    }
    if (src == dst) return fail(ERR_BAD_ARG, "coset_dft: in-place not supported");
    return run_dif(cx, stream, src, height, dst, n, width, false, &sc, shift, true);
}

int ntt_coset_lde_from_coeffs(Context& cx, hipStream_t stream, const uint32_t* coeffs, uint32_t* dst, uint32_t* scratch,
                              uint64_t height, uint32_t width, uint32_t added_bits, uint32_t shift) {
    if (!height || !width) return OK;
    if (!is_pow2(height)) return fail(ERR_BAD_ARG, "hip backend requires power-of-two height, got " + std::to_string(height));
    const uint32_t n = log2u(height);
    if (n + added_bits > bb::TWO_ADICITY) return fail(ERR_BAD_ARG, "LDE height exceeds BabyBear two-adicity");
    { int rcs = cx.reserve_scale_slots(8); if (rcs) return rcs; }
    int rcn = lde_narrow(cx, stream, coeffs, dst, n, added_bits, width, shift, true, true);
    if (rcn != 1) return rcn;
    // shapes outside the narrow plan: evaluations over the subgroup first, then the ordinary LDE
    int rc = ntt_dft(cx, stream, coeffs, scratch, height, width, false);
    if (rc) return rc;
    return ntt_coset_lde(cx, stream, scratch, dst, height, width, added_bits, shift, true);
}

int ntt_coset_lde(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height,
                  uint32_t width, uint32_t added_bits, uint32_t shift, bool bit_reversed_out) {
    if (!height || !width) return OK;
    if (!is_pow2(height)) return fail(ERR_BAD_ARG, "hip backend requires power-of-two height, got " + std::to_string(height));
    uint32_t n = log2u(height);
    uint32_t m = n + added_bits;
    if (m > bb::TWO_ADICITY) return fail(ERR_BAD_ARG, "LDE height exceeds BabyBear two-adicity");
    if (src == dst) return fail(ERR_BAD_ARG, "coset_lde: in-place not supported");
    size_t bytes = height * width * 4;
    // room for every scale table this call may fetch (one per coset of the narrow plan): the bounded cache is only
    // ever emptied here, before any pointer into it is taken
    { int rcs = cx.reserve_scale_slots(8); if (rcs) return rcs; }
    // Narrow matrices (the fib_air trace / quotient): two digits per direction, three launches.
    {
        int rcn = lde_narrow(cx, stream, src, dst, n, added_bits, width, shift, bit_reversed_out);
        if (rcn != 1) return rcn;
    }
    // Fused plan (three-pass shapes whose digits line up): inverse passes 1..2, fused middle, forward passes 2..3.
    {
        int rcf = lde_fused(cx, stream, src, dst, n, added_bits, width, shift, bit_reversed_out);
        if (rcf != 1) return rcf;  // 0 = done, <0 = error, 1 = shape not covered: separate transforms below
    }
    // 1. coefficients (natural order) into scratch: inverse DIT, 1/N folded into the scale table below
    int rc = cx.ws(stream, 1).reserve(bytes);
    if (rc) return rc;
    uint32_t* coeffs = cx.ws(stream, 1).as<uint32_t>();
    if (n == 0) P3_HIP(hipMemcpyAsync(coeffs, src, bytes, hipMemcpyDeviceToDevice, stream));
    else {
        rc = run_dit(cx, stream, src, coeffs, n, width, true, false, 0);
        if (rc) return rc;
    }
    if (m == 0) {
        P3_HIP(hipMemcpyAsync(dst, coeffs, bytes, hipMemcpyDeviceToDevice, stream));
        return OK;
    }
    // 2. forward DIF over 2^m rows: rows >= height are zero, row j scaled by shift^j / N
    TwoLevelTable sc;
    uint32_t hinv = bb::inv(bb::to_monty((uint32_t)height));
    rc = cx.get_scale_table(stream, shift, n, hinv, &sc);
    if (rc) return rc;
    return run_dif(cx, stream, coeffs, height, dst, m, width, false, &sc, shift, !bit_reversed_out);
}

}  // namespace p3
