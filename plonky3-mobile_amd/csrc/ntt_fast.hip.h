// Lean pass kernels for the in-place (strided-run) passes of the NTT plans: the shapes that carry most of the
// LDE's work (two of the three passes in each direction).  Compared with the general ntt_pass_kernel:
//   * tile size is a template parameter (B = 6, 7 or 8 stages = radix-16 round + radix-2^(B-4) round);
//   * the first round runs on the registers the global loads land in and the last round stores straight from
//     registers: ONE LDS exchange and ONE barrier per pass instead of three;
//   * per-thread addressing is a base pointer plus compile-time multiples of one row stride;
//   * the inter-pass twiddles w^(rev(pt) * lo) of a thread's 16 rows are c * phi^r (r = 0..15): two table
//     look-ups per thread and a doubling ladder, instead of two look-ups per element.
// Included by ntt.hip (needs PassArgs, two_level, rev_bits).
#pragma once

namespace p3 {

__device__ __forceinline__ constexpr uint32_t crev(uint32_t v, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

// pw[r] = c * phi^r for r < N (N a power of two), by doubling: N + log2(N) - 1 products, short chains.
template <int N>
__device__ __forceinline__ void power_ladder(uint32_t c, uint32_t phi, uint32_t (&pw)[N]) {
    pw[0] = c;
    uint32_t step = phi;
#pragma unroll
    for (int len = 1; len < N; len <<= 1) {
#pragma unroll
        for (int i = 0; i < len; i++) pw[len + i] = bb::mul(pw[i], step);
        if (len * 2 < N) step = bb::sqr(step);
    }
}

// MODE 2: DIT, in place, pre-twiddle (+ optional uniform scale on store).
// MODE 3: DIF, in place, post-twiddle (+ optional per-row scale and zero padding on load).
// LR = log2(rows per lane): 4 (radix-16 first round) for B <= 8, 5 (radix-32) for B = 9, 10.
template <int B, int MODE>
__global__ void __launch_bounds__(B >= 10 ? 1024 : 512) ntt_fast_kernel(PassArgs a) {
    constexpr int LR = B >= 9 ? 5 : 4;
    constexpr uint32_t NR = 1u << LR;
    constexpr uint32_t RUN = 32, STRIDE = 33, NPTS = 1u << B, NTH = NPTS * RUN / NR, GSPAN = NPTS / NR;
    constexpr int RR2 = B - LR;            // stages of the short round
    constexpr uint32_t NSUB = 1u << (LR - RR2), SUB2 = 1u << RR2;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* tile = smem;
    uint32_t* twl = smem + NPTS * STRIDE;
    const uint32_t tid = threadIdx.x, x = tid & 31, g = tid >> 5;
    for (uint32_t i = tid; i + 1 < NPTS; i += NTH) twl[i] = a.tile_tw[i];

    const uint32_t bid = blockIdx.x;
    const uint32_t hi = bid / a.n_inner;
    const uint64_t f = (uint64_t)(bid % a.n_inner) * RUN + x;
    const bool valid = f < ((uint64_t)a.W << a.s0);
    const uint32_t lo = a.wshift != 0xffffffffu ? (uint32_t)(f >> a.wshift) : (uint32_t)(f / a.W);
    const uint64_t stride = (uint64_t)a.W << a.s0;  // words between consecutive tile rows
    const uint64_t base = ((uint64_t)hi << (a.s0 + B)) * a.W + f;
    uint32_t v[NR];

    if constexpr (MODE == 2) {
        // ---- load rows pt = 16 g + j, pre-twiddle w^(rev_B(pt) * lo) ----
        const uint32_t* p = a.src + base + (uint64_t)(g * NR) * stride;
#pragma unroll
        for (uint32_t j = 0; j < NR; j++) v[j] = valid ? p[j * stride] : 0u;
        {
            uint32_t c = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo * rev_bits(g, B - LR));
            uint32_t phi = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo << (B - LR));
            uint32_t pw[NR];
            power_ladder<(int)NR>(c, phi, pw);
#pragma unroll
            for (uint32_t j = 0; j < NR; j++) v[j] = bb::mul(v[j], pw[crev(j, LR)]);
        }
        __syncthreads();  // twl ready
        // ---- round 1: stages 0..3 on registers ----
#pragma unroll
        for (int u = 0; u < LR; u++) {
#pragma unroll
            for (uint32_t j0 = 0; j0 < NR; j0++) {
                if (j0 & (1u << u)) continue;
                const uint32_t j1 = j0 | (1u << u);
                const uint32_t w = twl[(1u << u) - 1u + (j0 & ((1u << u) - 1u))];
                uint32_t t = bb::mul(v[j1], w), s = v[j0];
                v[j0] = bb::add(s, t);
                v[j1] = bb::sub(s, t);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < NR; j++) tile[(g * NR + j) * STRIDE + x] = v[j];
        __syncthreads();
        // ---- round 2: stages 4..B-1; pt = (ji << 4) | o, o = js * GSPAN + g ----
#pragma unroll
        for (uint32_t js = 0; js < NSUB; js++) {
            const uint32_t o = js * GSPAN + g;
#pragma unroll
            for (uint32_t ji = 0; ji < SUB2; ji++) v[js * SUB2 + ji] = tile[((ji << LR) | o) * STRIDE + x];
#pragma unroll
            for (int u = 0; u < RR2; u++) {
                const uint32_t k = LR + u;
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) {
                    if (ji & (1u << u)) continue;
                    const uint32_t j0 = js * SUB2 + ji, j1 = j0 | (1u << u);
                    const uint32_t w = twl[(1u << k) - 1u + ((ji & ((1u << u) - 1u)) << LR) + o];
                    uint32_t t = bb::mul(v[j1], w), s = v[j0];
                    v[j0] = bb::add(s, t);
                    v[j1] = bb::sub(s, t);
                }
            }
        }
        if (!valid) return;
        uint32_t* q = a.dst + base;
#pragma unroll
        for (uint32_t js = 0; js < NSUB; js++)
#pragma unroll
            for (uint32_t ji = 0; ji < SUB2; ji++) {
                uint32_t val = v[js * SUB2 + ji];
                if (a.has_us) val = bb::mul(val, a.uscale);
                q[(uint64_t)((ji << LR) | (js * GSPAN + g)) * stride] = val;
            }
    } else {
        // ---- DIF: load rows pt = (j << (B-4)) | g (zero padding / row scale on the first forward pass) ----
        const uint64_t row0 = ((uint64_t)hi << (a.s0 + B)) + ((uint64_t)g << a.s0) + lo;
        const uint64_t drow = (uint64_t)GSPAN << a.s0;  // rows between consecutive j
        const uint32_t* p = a.src + base + (uint64_t)g * stride;
#pragma unroll
        for (uint32_t j = 0; j < NR; j++) v[j] = (valid && row0 + j * drow < a.src_rows) ? p[(uint64_t)(j * GSPAN) * stride] : 0u;
        if (a.has_sc && valid && row0 < a.src_rows) {
            // sc(row) = mult * shift^row: c_j = sc(row0) * (shift^drow)^j
            uint32_t c = two_level(a.sc_lo, a.sc_hi, a.sc_T, row0);
            uint32_t pw[NR];
            power_ladder<(int)NR>(c, a.sc_step, pw);
#pragma unroll
            for (uint32_t j = 0; j < NR; j++) v[j] = bb::mul(v[j], pw[j]);  // rows beyond src_rows hold 0 already
        }
        __syncthreads();  // twl ready
        // ---- round A: stages B-1..B-4 on registers (k0 = B-4, low bits of pt = g) ----
#pragma unroll
        for (int u = LR - 1; u >= 0; u--) {
            const uint32_t k = (B - LR) + u;
#pragma unroll
            for (uint32_t j0 = 0; j0 < NR; j0++) {
                if (j0 & (1u << u)) continue;
                const uint32_t j1 = j0 | (1u << u);
                const uint32_t w = twl[(1u << k) - 1u + ((j0 & ((1u << u) - 1u)) << (B - LR)) + g];
                uint32_t s = v[j0], c = v[j1];
                v[j0] = bb::add(s, c);
                v[j1] = bb::mul(bb::sub(s, c), w);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < NR; j++) tile[((j << (B - LR)) | g) * STRIDE + x] = v[j];
        __syncthreads();
        // ---- round B: stages RR2-1..0; pt = (o << RR2) | ji ----
        const uint32_t phi = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo << (B - RR2));
        uint32_t* q = a.dst + base;
#pragma unroll
        for (uint32_t js = 0; js < NSUB; js++) {
            const uint32_t o = js * GSPAN + g;
            uint32_t e[SUB2];
#pragma unroll
            for (uint32_t ji = 0; ji < SUB2; ji++) e[ji] = tile[((o << RR2) | ji) * STRIDE + x];
#pragma unroll
            for (int u = RR2 - 1; u >= 0; u--) {
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) {
                    if (ji & (1u << u)) continue;
                    const uint32_t j1 = ji | (1u << u);
                    const uint32_t w = twl[(1u << u) - 1u + (ji & ((1u << u) - 1u))];
                    uint32_t s = e[ji], c = e[j1];
                    e[ji] = bb::add(s, c);
                    e[j1] = bb::mul(bb::sub(s, c), w);
                }
            }
            // post-twiddle w^(rev_B(pt) * lo), rev_B((o << RR2) | ji) = (rev(ji) << (B-RR2)) | rev(o)
            uint32_t c = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo * rev_bits(o, B - RR2));
            uint32_t pw[SUB2];
            power_ladder<(int)SUB2>(c, phi, pw);
            if (valid) {
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++)
                    q[(uint64_t)((o << RR2) | ji) * stride] = bb::mul(e[ji], pw[crev(ji, RR2)]);
            }
        }
    }
}


// MODE 1: first DIT pass (s0 = 0): bit-reversal gather straight into the first round's registers; results go to
//         contiguous groups (selected in bit-reversed order) — directly for W >= 32, through one more LDS
//         exchange + column-wise copy for narrow power-of-two widths.
// MODE 4: last DIF pass (s0 = 0) in place on contiguous groups: narrow widths are staged through LDS on both
//         sides (column-wise copies keep HBM accesses contiguous), W >= 32 loads/stores directly.
template <int B, int MODE>
__global__ void __launch_bounds__(512) ntt_fast_group_kernel(PassArgs a) {
    constexpr uint32_t RUN = 32, STRIDE = 33, NPTS = 1u << B, NTH = NPTS * 2, GSPAN = NPTS / 16;
    constexpr int RR2 = B - 4;
    constexpr uint32_t NSUB = 1u << (4 - RR2), SUB2 = 1u << RR2;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* tile = smem;
    uint32_t* twl = smem + NPTS * STRIDE;
    const uint32_t tid = threadIdx.x, x = tid & 31, g = tid >> 5;
    for (uint32_t i = tid; i + 1 < NPTS; i += NTH) twl[i] = a.tile_tw[i];

    const uint32_t bid = blockIdx.x;
    const uint32_t gbits = a.n - B;
    const bool narrow = a.W < RUN;
    uint32_t h0, t, c;
    bool valid;
    if (narrow) {
        h0 = bid * a.G;
        t = x >> a.wshift;
        c = x & (a.W - 1);
        valid = t < a.G && ((uint64_t)h0 + t) < (1ull << gbits);
    } else {
        h0 = bid / a.n_inner;
        t = 0;
        c = (bid % a.n_inner) * RUN + x;
        valid = c < a.W;
    }
    const uint64_t h = (uint64_t)h0 + t;
    uint32_t v[16];

    // column-wise cooperative copy between the LDS tile and the tile's contiguous groups (narrow widths)
    auto colwise = [&](bool to_lds, const uint32_t* src, uint32_t* dst, bool rev_sel) {
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t idx = tid + j * NTH;
            const uint32_t cc = idx & (a.W - 1), q = idx >> a.wshift;
            const uint32_t pt = q & (NPTS - 1), tt = q >> B;
            const uint64_t hh = (uint64_t)h0 + tt;
            const bool ok = tt < a.G && hh < (1ull << gbits);
            const uint64_t row = ((rev_sel ? (uint64_t)rev_bits((uint32_t)hh, gbits) : hh) << B) + pt;
            const uint64_t word = (row << a.wshift) + cc;
            if (to_lds) { if (ok) tile[pt * STRIDE + (tt << a.wshift) + cc] = src[word]; }
            else { if (ok) { uint32_t val = tile[pt * STRIDE + (tt << a.wshift) + cc]; if (a.has_us) val = bb::mul(val, a.uscale); dst[word] = val; } }
        }
    };

    if constexpr (MODE == 1) {
        // rows (rev_B(16 g + j) << gbits) + h = h + (rev(g) << gbits) + rev4(j) * 2^(B-4+gbits)
        const uint64_t stride1 = ((uint64_t)a.W << (B - 4)) << gbits;
        const uint32_t* p = a.src + (h + ((uint64_t)rev_bits(g, B - 4) << gbits)) * a.W + c;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = valid ? p[(uint64_t)crev(j, 4) * stride1] : 0u;
        __syncthreads();  // twl ready
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (uint32_t j0 = 0; j0 < 16; j0++) {
                if (j0 & (1u << u)) continue;
                const uint32_t j1 = j0 | (1u << u);
                const uint32_t w = twl[(1u << u) - 1u + (j0 & ((1u << u) - 1u))];
                uint32_t tt = bb::mul(v[j1], w), s = v[j0];
                v[j0] = bb::add(s, tt);
                v[j1] = bb::sub(s, tt);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) tile[(g * 16 + j) * STRIDE + x] = v[j];
        __syncthreads();
#pragma unroll
        for (uint32_t js = 0; js < NSUB; js++) {
            const uint32_t o = js * GSPAN + g;
#pragma unroll
            for (uint32_t ji = 0; ji < SUB2; ji++) v[js * SUB2 + ji] = tile[((ji << 4) | o) * STRIDE + x];
#pragma unroll
            for (int u = 0; u < RR2; u++) {
                const uint32_t k = 4 + u;
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) {
                    if (ji & (1u << u)) continue;
                    const uint32_t j0 = js * SUB2 + ji, j1 = j0 | (1u << u);
                    const uint32_t w = twl[(1u << k) - 1u + ((ji & ((1u << u) - 1u)) << 4) + o];
                    uint32_t tt = bb::mul(v[j1], w), s = v[j0];
                    v[j0] = bb::add(s, tt);
                    v[j1] = bb::sub(s, tt);
                }
            }
        }
        if (narrow) {
#pragma unroll
            for (uint32_t js = 0; js < NSUB; js++)
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) tile[((ji << 4) | (js * GSPAN + g)) * STRIDE + x] = v[js * SUB2 + ji];
            __syncthreads();
            colwise(false, nullptr, a.dst, true);
        } else if (valid) {
            uint32_t* q = a.dst + ((uint64_t)rev_bits(h0, gbits) << B) * a.W + c;
#pragma unroll
            for (uint32_t js = 0; js < NSUB; js++)
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) {
                    uint32_t val = v[js * SUB2 + ji];
                    if (a.has_us) val = bb::mul(val, a.uscale);
                    q[(uint64_t)((ji << 4) | (js * GSPAN + g)) * a.W] = val;
                }
        }
    } else {
        // ---- MODE 4 ----
        if (narrow) {
            colwise(true, a.src, nullptr, false);
            __syncthreads();  // also covers twl
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) v[j] = tile[((j << (B - 4)) | g) * STRIDE + x];
        } else {
            const uint32_t* p = a.src + ((uint64_t)h0 << B) * a.W + c;
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) v[j] = valid ? p[(uint64_t)((j << (B - 4)) | g) * a.W] : 0u;
            __syncthreads();  // twl ready
        }
#pragma unroll
        for (int u = 3; u >= 0; u--) {
            const uint32_t k = (B - 4) + u;
#pragma unroll
            for (uint32_t j0 = 0; j0 < 16; j0++) {
                if (j0 & (1u << u)) continue;
                const uint32_t j1 = j0 | (1u << u);
                const uint32_t w = twl[(1u << k) - 1u + ((j0 & ((1u << u) - 1u)) << (B - 4)) + g];
                uint32_t s = v[j0], d = v[j1];
                v[j0] = bb::add(s, d);
                v[j1] = bb::mul(bb::sub(s, d), w);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) tile[((j << (B - 4)) | g) * STRIDE + x] = v[j];
        __syncthreads();
#pragma unroll
        for (uint32_t js = 0; js < NSUB; js++) {
            const uint32_t o = js * GSPAN + g;
            uint32_t e[SUB2];
#pragma unroll
            for (uint32_t ji = 0; ji < SUB2; ji++) e[ji] = tile[((o << RR2) | ji) * STRIDE + x];
#pragma unroll
            for (int u = RR2 - 1; u >= 0; u--) {
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) {
                    if (ji & (1u << u)) continue;
                    const uint32_t j1 = ji | (1u << u);
                    const uint32_t w = twl[(1u << u) - 1u + (ji & ((1u << u) - 1u))];
                    uint32_t s = e[ji], d = e[j1];
                    e[ji] = bb::add(s, d);
                    e[j1] = bb::mul(bb::sub(s, d), w);
                }
            }
            if (narrow) {
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) tile[((o << RR2) | ji) * STRIDE + x] = e[ji];
            } else if (valid) {
                uint32_t* q = a.dst + ((uint64_t)h0 << B) * a.W + c;
#pragma unroll
                for (uint32_t ji = 0; ji < SUB2; ji++) q[(uint64_t)((o << RR2) | ji) * a.W] = e[ji];
            }
        }
        if (narrow) {
            __syncthreads();
            colwise(false, nullptr, a.dst, false);
        }
    }
}


// Fused middle of the coset LDE: the LAST inverse (DIT) pass and the FIRST forward (DIF) pass share their row
// stride 2^s0, so one workgroup can finish the inverse transform of its rows (BI stages), scale the
// coefficients by shift^row / N, zero-extend to 2^(BI+A) rows and run the forward pass's top BF = BI + A
// stages without the coefficients ever leaving the chip.  Saves one launch and one full write + read of
// the coefficient matrix.  Inverse tile twiddles / tables: a.tile_tw, a.tw_*; forward: a.tile_tw2, a.tw2_*.
template <int BI, int A>
__global__ void __launch_bounds__(512) ntt_fused_mid_kernel(PassArgs a) {
    constexpr int BF = BI + A;
    constexpr uint32_t RUN = 32, STRIDE = 33, NI = 1u << BI, NF = 1u << BF, NTH = NF * 2;
    constexpr int RI2 = BI - 4, RF2 = BF - 4;
    constexpr uint32_t NSUBI = 1u << (4 - RI2), SUBI = 1u << RI2, GSPANI = NI / 16;
    constexpr uint32_t NSUBF = 1u << (4 - RF2), SUBF = 1u << RF2, GSPANF = NF / 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* tile = smem;                    // NF rows
    uint32_t* twl_i = smem + NF * STRIDE;     // inverse stage table (NI - 1 words used)
    uint32_t* twl_f = twl_i + NF;             // forward stage table (NF - 1 words)
    const uint32_t tid = threadIdx.x, x = tid & 31, g = tid >> 5;
    for (uint32_t i = tid; i + 1 < NI; i += NTH) twl_i[i] = a.tile_tw[i];
    for (uint32_t i = tid; i + 1 < NF; i += NTH) twl_f[i] = a.tile_tw2[i];

    const uint64_t f = (uint64_t)blockIdx.x * RUN + x;  // the top digit spans the whole height: hi = 0
    const bool valid = f < ((uint64_t)a.W << a.s0);
    const uint32_t lo = a.wshift != 0xffffffffu ? (uint32_t)(f >> a.wshift) : (uint32_t)(f / a.W);
    const uint64_t stride = (uint64_t)a.W << a.s0;
    uint32_t v[16];

    // ---------------- inverse: DIT over BI stages (threads with g < NI/16) ----------------
    const bool inv_active = g < GSPANI;
    if (inv_active) {
        const uint32_t* p = a.src + f + (uint64_t)(g * 16) * stride;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = valid ? p[j * stride] : 0u;
        uint32_t c = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo * rev_bits(g, BI - 4));
        uint32_t phi = two_level(a.tw_lo, a.tw_hi, a.tw_T, (uint64_t)lo << (BI - 4));
        uint32_t pw[16];
        power_ladder<16>(c, phi, pw);
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = bb::mul(v[j], pw[crev(j, 4)]);
    }
    __syncthreads();  // stage tables ready
    if (inv_active) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (uint32_t j0 = 0; j0 < 16; j0++) {
                if (j0 & (1u << u)) continue;
                const uint32_t j1 = j0 | (1u << u);
                const uint32_t w = twl_i[(1u << u) - 1u + (j0 & ((1u << u) - 1u))];
                uint32_t t = bb::mul(v[j1], w), s = v[j0];
                v[j0] = bb::add(s, t);
                v[j1] = bb::sub(s, t);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) tile[(g * 16 + j) * STRIDE + x] = v[j];
    }
    __syncthreads();
    if (inv_active) {
        // coefficient row = pt * 2^s0 + lo; scale sc(row) = sc(lo) * (shift^(2^s0))^pt
        const uint32_t c0 = valid ? two_level(a.sc_lo, a.sc_hi, a.sc_T, lo) : 0u;
#pragma unroll
        for (uint32_t js = 0; js < NSUBI; js++) {
            const uint32_t o = js * GSPANI + g;
#pragma unroll
            for (uint32_t ji = 0; ji < SUBI; ji++) v[js * SUBI + ji] = tile[((ji << 4) | o) * STRIDE + x];
#pragma unroll
            for (int u = 0; u < RI2; u++) {
                const uint32_t k = 4 + u;
#pragma unroll
                for (uint32_t ji = 0; ji < SUBI; ji++) {
                    if (ji & (1u << u)) continue;
                    const uint32_t j0 = js * SUBI + ji, j1 = j0 | (1u << u);
                    const uint32_t w = twl_i[(1u << k) - 1u + ((ji & ((1u << u) - 1u)) << 4) + o];
                    uint32_t t = bb::mul(v[j1], w), s = v[j0];
                    v[j0] = bb::add(s, t);
                    v[j1] = bb::sub(s, t);
                }
            }
            // pt = (ji << 4) | o: scale = c0 * step^o * (step^16)^ji, step = shift^(2^s0)
            uint32_t pw[SUBI];
            power_ladder<(int)SUBI>(bb::mul(c0, bb::pow(a.sc_step, o)), a.sc_step16, pw);
#pragma unroll
            for (uint32_t ji = 0; ji < SUBI; ji++) v[js * SUBI + ji] = bb::mul(v[js * SUBI + ji], pw[ji]);
        }
    }
    __syncthreads();  // every round-2 read of the tile is done before it is overwritten
    if (inv_active) {
#pragma unroll
        for (uint32_t js = 0; js < NSUBI; js++)
#pragma unroll
            for (uint32_t ji = 0; ji < SUBI; ji++) tile[((ji << 4) | (js * GSPANI + g)) * STRIDE + x] = v[js * SUBI + ji];
    }
    __syncthreads();
    // ---------------- forward: DIF over BF stages, rows >= NI are zero ----------------
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t pt = (j << (BF - 4)) | g;
        v[j] = pt < NI ? tile[pt * STRIDE + x] : 0u;
    }
    __syncthreads();  // all coefficient reads done before round A overwrites the tile
#pragma unroll
    for (int u = 3; u >= 0; u--) {
        const uint32_t k = (BF - 4) + u;
#pragma unroll
        for (uint32_t j0 = 0; j0 < 16; j0++) {
            if (j0 & (1u << u)) continue;
            const uint32_t j1 = j0 | (1u << u);
            const uint32_t w = twl_f[(1u << k) - 1u + ((j0 & ((1u << u) - 1u)) << (BF - 4)) + g];
            uint32_t s = v[j0], d = v[j1];
            v[j0] = bb::add(s, d);
            v[j1] = bb::mul(bb::sub(s, d), w);
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) tile[((j << (BF - 4)) | g) * STRIDE + x] = v[j];
    __syncthreads();
    const uint32_t phi2 = two_level(a.tw2_lo, a.tw2_hi, a.tw2_T, (uint64_t)lo << (BF - RF2));
    uint32_t* q = a.dst + f;
#pragma unroll
    for (uint32_t js = 0; js < NSUBF; js++) {
        const uint32_t o = js * GSPANF + g;
        uint32_t e[SUBF];
#pragma unroll
        for (uint32_t ji = 0; ji < SUBF; ji++) e[ji] = tile[((o << RF2) | ji) * STRIDE + x];
#pragma unroll
        for (int u = RF2 - 1; u >= 0; u--) {
#pragma unroll
            for (uint32_t ji = 0; ji < SUBF; ji++) {
                if (ji & (1u << u)) continue;
                const uint32_t j1 = ji | (1u << u);
                const uint32_t w = twl_f[(1u << u) - 1u + (ji & ((1u << u) - 1u))];
                uint32_t s = e[ji], d = e[j1];
                e[ji] = bb::add(s, d);
                e[j1] = bb::mul(bb::sub(s, d), w);
            }
        }
        uint32_t c = two_level(a.tw2_lo, a.tw2_hi, a.tw2_T, (uint64_t)lo * rev_bits(o, BF - RF2));
        uint32_t pw[SUBF];
        power_ladder<(int)SUBF>(c, phi2, pw);
        if (valid) {
#pragma unroll
            for (uint32_t ji = 0; ji < SUBF; ji++) q[(uint64_t)((o << RF2) | ji) * stride] = bb::mul(e[ji], pw[crev(ji, RF2)]);
        }
    }
}

}  // namespace p3
