// Poseidon2 Merkle-tree MMCS on gfx950: leaf sponge, 2-to-1 compression layers, openings.
// Stands in for Plonky3's MerkleTreeMmcs (the reference passes the Keccak flavour into the PCS at
// native/src/fib_air.rs:40-51; north_star asks for the Poseidon2 one):
//   leaf     PaddingFreeSponge<Perm16, 16, 8, 8>   overwrite-mode absorb, permute per full/partial chunk
//   compress TruncatedPermutation<Perm16, 2, 8, 16> state = left||right, permute, first 8 words
//   tree     digest layer l+1[i] = compress(layer l[2i], layer l[2i+1]); matrices whose height equals a
//            layer's length are hashed row-wise and compressed in (compress_and_inject); all layers stay
//            in HBM for open_batch.
// One permutation per lane, state in VGPRs; digests are 32-byte records read/written as 2 x dwordx4.
#include <algorithm>

#include "common.h"
#include "mmcs.h"
#include "poseidon2.hip.h"
#include "poseidon2_coop.hip.h"
#include "poseidon2_f64.hip.h"
#include "poseidon2_q4.hip.h"
#include "keccak.hip.h"

namespace p3 {

constexpr int MAX_CLASS_MATS = 16;

struct RowSet {  // the matrices of one height class, concatenated row-wise
    const uint32_t* ptr[MAX_CLASS_MATS];
    uint32_t width[MAX_CLASS_MATS];
    uint32_t stride[MAX_CLASS_MATS];  // words between two rows (= width unless the matrix is a column group of a wider one)
    uint32_t count;
    uint32_t total;
};

__device__ __forceinline__ void load_digest(const uint32_t* p, uint32_t* out8) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    out8[0] = a.x; out8[1] = a.y; out8[2] = a.z; out8[3] = a.w;
    out8[4] = b.x; out8[5] = b.y; out8[6] = b.z; out8[7] = b.w;
}
__device__ __forceinline__ void store_digest(uint32_t* p, const uint32_t* s) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(s[0], s[1], s[2], s[3]);
    q[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// PaddingFreeSponge over the concatenated row `r` of the set; leaves the digest in s[0..8).
__device__ __forceinline__ void sponge_row(const RowSet& rs, uint64_t r, uint32_t (&s)[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0;
    if (rs.count == 1) {
        const uint32_t* row = rs.ptr[0] + r * rs.stride[0];
        const uint32_t w = rs.width[0];
        for (uint32_t k = 0; k < w; k += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (k + i < w) s[i] = row[k + i];
            p2::permute(s);
        }
        return;
    }
    uint32_t m = 0, off = 0;  // current matrix and column inside it
    for (uint32_t k = 0; k < rs.total; k += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (k + i < rs.total) {
                while (off >= rs.width[m]) { m++; off = 0; }
                s[i] = rs.ptr[m][r * rs.stride[m] + off];
                off++;
            }
        }
        p2::permute(s);
    }
}

__global__ void __launch_bounds__(256) leaf_hash_kernel(RowSet rs, uint64_t n_rows, uint32_t* digests) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    uint32_t s[16];
    sponge_row(rs, r, s);
    store_digest(digests + r * 8, s);
}

// next[i] = compress(prev[2i], prev[2i+1]), optionally followed by compress(., sponge(row i of the set)).
__global__ void __launch_bounds__(256) compress_layer_kernel(const uint32_t* prev, uint32_t* next, uint64_t n_out,
                                                             RowSet rs, uint32_t inject) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    uint32_t s[16];
    load_digest(prev + i * 16, s);
    load_digest(prev + i * 16 + 8, s + 8);
    p2::permute(s);
    if (inject) {
        uint32_t h[16];
        sponge_row(rs, i, h);
#pragma unroll
        for (int k = 0; k < 8; k++) s[8 + k] = h[k];
        p2::permute(s);
    }
    store_digest(next + i * 8, s);
}

// Finishes a tree whose current layer has at most 2*blockDim digests: all remaining (injection-free)
// levels inside one workgroup, current layer mirrored in LDS.  layers: consecutive layers in HBM,
// layer with n digests followed by the one with n/2.
__global__ void __launch_bounds__(256) tree_top_kernel(uint32_t* layer0, uint32_t n0) {
    P3_LATENCY_BOUND_KERNEL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < n0 * 8; i += blockDim.x) lds[i] = layer0[i];
    __syncthreads();
    uint32_t* out = layer0 + (size_t)n0 * 8;
    for (uint32_t n = n0; n > 1; n >>= 1) {
        uint32_t half = n >> 1;
        uint32_t s[16];
        bool act = tid < half;
        if (act) {
#pragma unroll
            for (int k = 0; k < 16; k++) s[k] = lds[tid * 16 + k];
            p2::permute(s);
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int k = 0; k < 8; k++) lds[tid * 8 + k] = s[k];
            store_digest(out + tid * 8, s);
        }
        __syncthreads();
        out += (size_t)half * 8;
    }
}


// ---- fp64 variants of the two large one-state-per-lane kernels (poseidon2_f64.hip.h): same digests, ~7 % fewer
// issue cycles.  Memory stays Montgomery u32; conversion happens in registers at load/store.
__global__ void __launch_bounds__(256) leaf_hash_f64_kernel(const uint32_t* mat, uint32_t width, uint64_t n_rows, uint32_t* digests) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0.0;
    const uint32_t* row = mat + r * width;
    for (uint32_t k = 0; k < width; k += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (k + i < width) s[i] = p2f::load_elem(row[k + i]);
        p2f::permute(s);
        if (k + 8 < width) {
            // keep magnitudes small between absorptions (the next permutation assumes |s| <= 2^33)
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = p2f::reduce(s[i]);
        }
    }
    uint32_t d[8];
    const p2f::MagicRegs smk = p2f::magic_regs();
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = p2f::store_elem(s[i], smk);
    store_digest(digests + r * 8, d);
}
// ---- one state per DPP quad (poseidon2_q4.hip.h): the layers of 2^12 .. 2^15 digests, whose one-state-per-lane launches cost one
// permutation's issue time (11.5-12.4 us) whatever their size.  Lane q of a quad holds state elements 4q .. 4q+3: a child pair's
// sixteen words are four 16-byte loads of one quad, the digest is two 16-byte stores.
__global__ void __launch_bounds__(256) compress_layer_q4_kernel(const uint32_t* prev, uint32_t* next, uint64_t n_out, uint32_t prio) {
    if (prio) P3_LATENCY_BOUND_KERNEL();
    const uint32_t q = threadIdx.x & 3u;
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const bool act = i < n_out;  // idle quads run the permutation on zeros: every lane of a wave takes part in the exchanges
    const p2q::LaneConsts lc = p2q::lane_consts(q);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (act) {
        const uint4 w = *reinterpret_cast<const uint4*>(prev + i * 16 + 4 * q);
        s[0] = p2f::load_elem(w.x); s[1] = p2f::load_elem(w.y); s[2] = p2f::load_elem(w.z); s[3] = p2f::load_elem(w.w);
    }
    p2q::permute(s, lc, q == 0);
    if (act && q < 2) {
        const p2f::MagicRegs smk = p2f::magic_regs();
        *reinterpret_cast<uint4*>(next + i * 8 + 4 * q) =
            make_uint4(p2f::store_elem(s[0], smk), p2f::store_elem(s[1], smk), p2f::store_elem(s[2], smk), p2f::store_elem(s[3], smk));
    }
}
// the sponge over the rows of ONE dense matrix, one row per quad: lanes 0 and 1 of the quad overwrite their elements with the next
// eight words of the row (overwrite-mode absorb), every lane permutes
__global__ void __launch_bounds__(256) leaf_hash_q4_kernel(const uint32_t* mat, uint32_t width, uint64_t n_rows, uint32_t* digests, uint32_t prio) {
    if (prio) P3_LATENCY_BOUND_KERNEL();
    const uint32_t q = threadIdx.x & 3u;
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const bool act = r < n_rows;
    const p2q::LaneConsts lc = p2q::lane_consts(q);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const uint32_t* row = mat + (act ? r : 0) * width;
    for (uint32_t k = 0; k < width; k += 8) {
        if (act && q < 2) {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (k + 4 * q + i < width) s[i] = p2f::load_elem(row[k + 4 * q + i]);
        }
        p2q::permute(s, lc, q == 0);
        if (k + 8 < width) {
#pragma unroll
            for (int i = 0; i < 4; i++) s[i] = p2f::reduce(s[i]);  // keep magnitudes small between absorptions
        }
    }
    if (act && q < 2) {
        const p2f::MagicRegs smk = p2f::magic_regs();
        *reinterpret_cast<uint4*>(digests + r * 8 + 4 * q) =
            make_uint4(p2f::store_elem(s[0], smk), p2f::store_elem(s[1], smk), p2f::store_elem(s[2], smk), p2f::store_elem(s[3], smk));
    }
}
// WIDE rows (BASELINE configs[4]: 2633 words per row): in the kernel above a lane walks its own row, so every load instruction of a
// wave touches 64 lines 10 KB apart.  Here the workgroup's 256 rows are staged through LDS in chunks of 32 words per row, loaded
// 2 rows x 128 contiguous bytes per wave-instruction (coalesced), handed over transposed (row stride 36 words: the 16-byte reads of
// 16 consecutive lanes cover all 64 banks once), the next chunk in flight while four permutations absorb the current one.
constexpr uint32_t LEAF_WIDE_CHUNK = 32, LEAF_WIDE_STRIDE = 36;
__global__ void __launch_bounds__(256) leaf_hash_f64_wide_kernel(const uint32_t* mat, uint32_t width, uint64_t n_rows, uint32_t* digests) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[256 * LEAF_WIDE_STRIDE];
    const uint32_t tid = threadIdx.x, col = tid & 31u, rsub = tid >> 5;  // loader role: 8 row groups x 32 columns
    const uint64_t r0 = (uint64_t)blockIdx.x * 256u, r = r0 + tid;
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0.0;
    uint32_t pre[32];
    auto fetch = [&](uint32_t c0) {
#pragma unroll
        for (uint32_t i = 0; i < 32; i++) {
            const uint64_t row = r0 + rsub + 8u * i;
            pre[i] = (row < n_rows && c0 + col < width) ? mat[row * width + c0 + col] : 0u;
        }
    };
    fetch(0);
    for (uint32_t c0 = 0; c0 < width; c0 += LEAF_WIDE_CHUNK) {
        __syncthreads();  // the previous chunk has been read by every lane
#pragma unroll
        for (uint32_t i = 0; i < 32; i++) tile[(rsub + 8u * i) * LEAF_WIDE_STRIDE + col] = pre[i];
        __syncthreads();
        if (c0 + LEAF_WIDE_CHUNK < width) fetch(c0 + LEAF_WIDE_CHUNK);  // in flight during the four permutations below
        const uint4* mine = reinterpret_cast<const uint4*>(tile + tid * LEAF_WIDE_STRIDE);
#pragma unroll 1
        for (uint32_t k = 0; k < LEAF_WIDE_CHUNK && c0 + k < width; k += 8) {
            const uint4 a = mine[k / 4], b = mine[k / 4 + 1];
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (c0 + k + i < width) s[i] = p2f::load_elem(w[i]);
            p2f::permute(s);
            if (c0 + k + 8 < width) {
#pragma unroll
                for (int i = 0; i < 16; i++) s[i] = p2f::reduce(s[i]);
            }
        }
    }
    if (r >= n_rows) return;
    uint32_t d[8];
    const p2f::MagicRegs smk = p2f::magic_regs();
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = p2f::store_elem(s[i], smk);
    store_digest(digests + r * 8, d);
}
// the same sponge over the CONCATENATED rows of several matrices of one height (the (matrix, salt) pairs of the hiding
// MMCS, or any multi-matrix commitment): element k of the row comes from the matrix whose column range holds k
__global__ void __launch_bounds__(256) leaf_hash_f64_rowset_kernel(RowSet rs, uint64_t n_rows, uint32_t* digests) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0.0;
    uint32_t m = 0, off = 0;  // current matrix and column inside it
    for (uint32_t k = 0; k < rs.total; k += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (k + i < rs.total) {
                while (off >= rs.width[m]) { m++; off = 0; }
                s[i] = p2f::load_elem(rs.ptr[m][r * rs.stride[m] + off]);
                off++;
            }
        }
        p2f::permute(s);
        if (k + 8 < rs.total) {
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = p2f::reduce(s[i]);
        }
    }
    uint32_t d[8];
    const p2f::MagicRegs smk = p2f::magic_regs();
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = p2f::store_elem(s[i], smk);
    store_digest(digests + r * 8, d);
}
__global__ void __launch_bounds__(256) compress_layer_f64_kernel(const uint32_t* prev, uint32_t* next, uint64_t n_out) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    uint32_t w[16];
    load_digest(prev + i * 16, w);
    load_digest(prev + i * 16 + 8, w + 8);
    double s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = p2f::load_elem(w[k]);
    p2f::permute(s);
    uint32_t d[8];
    const p2f::MagicRegs smk = p2f::magic_regs();
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = p2f::store_elem(s[k], smk);
    store_digest(next + i * 8, d);
}

// ---- lane-cooperative kernels for small layers (16 lanes per permutation, poseidon2_coop.hip.h) ----
// next[i] = compress(prev[2i], prev[2i+1]); one 16-lane row per output digest.
__global__ void __launch_bounds__(256) compress_coop_kernel(const uint32_t* prev, uint32_t* next, uint32_t n_out) {
    P3_LATENCY_BOUND_KERNEL();
    const uint32_t lane16 = threadIdx.x & 15;
    const p2c::LaneConst lc = p2c::lane_constants(lane16);
    uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    bool act = i < n_out;
    uint32_t v = act ? prev[(size_t)i * 16 + lane16] : 0u;
    v = p2c::permute(v, lc);
    if (act && lane16 < 8) next[(size_t)i * 8 + lane16] = v;
}
// leaf digests of ONE matrix (any width): row r absorbed 8 words at a time by lanes 0..7 of its row of lanes.
__global__ void __launch_bounds__(256) leaf_coop_kernel(const uint32_t* mat, uint32_t width, uint32_t n_rows, uint32_t* digests) {
    P3_LATENCY_BOUND_KERNEL();
    const uint32_t lane16 = threadIdx.x & 15;
    const p2c::LaneConst lc = p2c::lane_constants(lane16);
    uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    bool act = r < n_rows;
    uint32_t v = 0;
    for (uint32_t k = 0; k < width; k += 8) {
        if (act && lane16 < 8 && k + lane16 < width) v = mat[(size_t)r * width + k + lane16];
        v = p2c::permute(v, lc);
    }
    if (act && lane16 < 8) digests[(size_t)r * 8 + lane16] = v;
}
// `levels` (<= 7) consecutive injection-free tree levels in ONE launch: each workgroup owns a chunk of up to
// 128 consecutive digests of layer `layer_in` (n_in digests, a power of two), keeps its shrinking layer in LDS
// and writes every intermediate layer to its place in HBM (layers are stored back to back).  16 lanes per
// permutation; waves whose four lane-rows are all idle at a level skip the permutation.  When the last
// level produces the root it is also written to `root_copy` (host-mapped) if given.
__global__ void __launch_bounds__(1024) tree_levels_coop_kernel(uint32_t* layer_in, uint32_t n_in, uint32_t chunk, uint32_t levels,
                                                                uint32_t* root_copy) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ uint32_t lds[128 * 8];
    const uint32_t tid = threadIdx.x, lane16 = tid & 15, grp = tid >> 4;
    const p2c::LaneConst lc = p2c::lane_constants(lane16);
    const uint32_t* src = layer_in + (size_t)blockIdx.x * chunk * 8;
    for (uint32_t i = tid; i < chunk * 8; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    uint32_t* out = layer_in + (size_t)n_in * 8;  // start of the next layer
    uint32_t cur_n = n_in;
    for (uint32_t l = 0; l < levels; l++) {
        const uint32_t half = chunk >> (l + 1);          // permutations of this workgroup at this level
        const bool wave_active = ((tid >> 6) << 2) < half;  // first lane-row of this wave
        uint32_t res = 0;
        if (wave_active) {
            uint32_t v = grp < half ? lds[grp * 16 + lane16] : 0u;
            res = p2c::permute(v, lc);
        }
        __syncthreads();
        if (grp < half && lane16 < 8) {
            lds[grp * 8 + lane16] = res;
            out[((size_t)blockIdx.x * half + grp) * 8 + lane16] = res;
            if (cur_n == 2 && root_copy) root_copy[lane16] = res;  // this level produced the root
        }
        __syncthreads();
        cur_n >>= 1;
        out += (size_t)cur_n * 8;
    }
}

__global__ void poseidon2_permute_kernel(uint32_t* states, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[16];
    uint4* q = reinterpret_cast<uint4*>(states + i * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) { uint4 v = q[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
    p2::permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = make_uint4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
}


__global__ void poseidon2_permute_f64_kernel(uint32_t* states, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s[16];
    uint4* q = reinterpret_cast<uint4*>(states + i * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 v = q[k];
        s[4 * k] = p2f::load_elem(v.x); s[4 * k + 1] = p2f::load_elem(v.y); s[4 * k + 2] = p2f::load_elem(v.z); s[4 * k + 3] = p2f::load_elem(v.w);
    }
    p2f::permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++)
        q[k] = make_uint4(p2f::store_elem(s[4 * k]), p2f::store_elem(s[4 * k + 1]), p2f::store_elem(s[4 * k + 2]), p2f::store_elem(s[4 * k + 3]));
}

// Test probe of the fp64 arithmetic: integer-valued doubles in (any magnitude the contract of the probed part allows),
// canonical Montgomery words out.  mode 0: the whole permutation (|v| <= 2^33); mode 1: the 13 internal rounds only
// (|v| <= 2^37 — the magnitudes the external layer can hand them, which no u32 input of the permutation reaches
// deterministically); mode 2: reduce() alone, lane by lane (rounding ties of the quotient estimate).
__global__ void poseidon2_f64_probe_kernel(const double* in, uint32_t* out, uint64_t n, int mode) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = in[i * 16 + k];
    if (mode == 0) p2f::permute(s);
    else if (mode == 1) p2f::internal_rounds(s, p2f::magic_regs());
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[k] = p2f::reduce(s[k]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) out[i * 16 + k] = p2f::store_elem(s[k]);
}
int poseidon2_f64_probe(hipStream_t stream, const double* d_in, uint32_t* d_out, uint64_t n, int mode) {
    if (!n) return OK;
    hipLaunchKernelGGL(poseidon2_f64_probe_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_in, d_out, n, mode);
    P3_HIP(hipGetLastError());
    return OK;
}
// variant: 0 = int32 Montgomery form, 1 = fp64 form (the two must agree word for word)
int poseidon2_permute_states_variant(hipStream_t stream, uint32_t* d_states, uint64_t n, int variant) {
    if (!n) return OK;
    if (variant == 1) hipLaunchKernelGGL(poseidon2_permute_f64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_states, n);
    else hipLaunchKernelGGL(poseidon2_permute_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_states, n);
    P3_HIP(hipGetLastError());
    return OK;
}

int poseidon2_permute_states(hipStream_t stream, uint32_t* d_states, uint64_t n) {
    return poseidon2_permute_states_variant(stream, d_states, n, 1);  // the fp64 form (poseidon2_f64.hip.h); variant 0 = int32 Montgomery
}

// ---- the reference's own hash configuration (native/src/fib_air.rs:28-38): Keccak-f[1600] sponge over u64 lanes ----
// SerializingHasher: the concatenated row's field elements (Montgomery words) are packed two per u64, first in the
// low half; PaddingFreeSponge<KeccakF, 25, 17, 4>: overwrite 17 lanes per block, permute after every full block and
// after a non-empty partial one; digest = lanes 0..3 = 8 words.
__device__ __forceinline__ uint32_t rowset_elem(const RowSet& rs, uint64_t r, uint32_t k, uint32_t& m, uint32_t& base) {
    // element k of the concatenated row (k only ever increases between calls: m/base are the running cursor)
    while (k - base >= rs.width[m]) { base += rs.width[m]; m++; }
    return rs.ptr[m][r * rs.stride[m] + (k - base)];
}
__device__ __forceinline__ void keccak_sponge_row(const RowSet& rs, uint64_t r, uint64_t (&st)[25]) {
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    const uint32_t n = rs.total, n64 = (n + 1) / 2;
    uint32_t m = 0, base = 0;
    for (uint32_t i = 0; i < n64; i += 17) {
#pragma unroll
        for (int k = 0; k < 17; k++) {
            const uint32_t e = 2 * (i + k);
            if (e < n) {
                uint64_t lo = rowset_elem(rs, r, e, m, base);
                uint64_t hi = e + 1 < n ? rowset_elem(rs, r, e + 1, m, base) : 0u;
                st[k] = lo | (hi << 32);
            }
        }
        if (i + 17 >= n64) kk::permute_digest(st);  // the last block: only the digest words are read
        else kk::permute(st);
    }
}
__device__ __forceinline__ void store_digest64(uint32_t* p, const uint64_t (&st)[25]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4((uint32_t)st[0], (uint32_t)(st[0] >> 32), (uint32_t)st[1], (uint32_t)(st[1] >> 32));
    q[1] = make_uint4((uint32_t)st[2], (uint32_t)(st[2] >> 32), (uint32_t)st[3], (uint32_t)(st[3] >> 32));
}
__global__ void __launch_bounds__(256) keccak_leaf_kernel(RowSet rs, uint64_t n_rows, uint32_t* digests) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    uint64_t st[25];
    keccak_sponge_row(rs, r, st);
    store_digest64(digests + r * 8, st);
}
// SALTED leaves of the hiding MMCS (MerkleTreeHidingMmcs, fib_air.rs:40-51): the tallest class is NP (matrix, salt) pairs, leaf row =
// m0 || s0 || m1 || s1 ..., every matrix W in {4, 6, 8} words wide, every salt 4, at most one rate block (34 words) in all — the
// hiding prover's trace (6 + 4), randomization (8 + 4), FRI layer (8 + 4) and four-chunk quotient (4 x (4 + 4)) commitments.  The
// generic kernel walks the row set one word at a time through a running (matrix, column) cursor; here every row piece is one
// 8- or 16-byte load at a compile-time position of the state.  `stride` = words between two rows of a matrix (the four chunk
// matrices are the column groups of ONE 16-word-wide LDE).
template <int NP>
struct SaltedRows {
    const uint32_t* mat[NP];
    const uint32_t* salt[NP];
    uint32_t stride;
};
template <int W, int NP>
__global__ void __launch_bounds__(256) keccak_leaf_salted_kernel(SaltedRows<NP> a, uint64_t n_rows, uint32_t* digests) {
    static_assert((W == 4 || W == 6 || W == 8) && NP * (W + 4) <= 34, "one rate block");
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        constexpr int L = (W + 4) / 2;  // lanes per pair
        const uint32_t* m = a.mat[p] + r * a.stride;
        if constexpr (W == 6) {
            const uint2* q = reinterpret_cast<const uint2*>(m);
#pragma unroll
            for (int k = 0; k < 3; k++) { const uint2 v = q[k]; st[p * L + k] = (uint64_t)v.x | ((uint64_t)v.y << 32); }
        } else {
            const uint4* q = reinterpret_cast<const uint4*>(m);
#pragma unroll
            for (int k = 0; k < W / 4; k++) {
                const uint4 v = q[k];
                st[p * L + 2 * k] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                st[p * L + 2 * k + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
            }
        }
        const uint4 sv = *reinterpret_cast<const uint4*>(a.salt[p] + r * 4);
        st[p * L + W / 2] = (uint64_t)sv.x | ((uint64_t)sv.y << 32);
        st[p * L + W / 2 + 1] = (uint64_t)sv.z | ((uint64_t)sv.w << 32);
    }
    kk::permute_digest(st);
    store_digest64(digests + r * 8, st);
}
// (matrix, salt) x NP with equal matrix widths and 16-byte-aligned rows?  Fills `a` and returns the width, else 0.
template <int NP>
static uint32_t salted_rows_of(const RowSet& rs, SaltedRows<NP>* a) {
    if (rs.count != 2u * NP) return 0;
    const uint32_t w = rs.width[0];
    if (w != 4 && w != 6 && w != 8) return 0;
    for (int p = 0; p < NP; p++) {
        if (rs.width[2 * p] != w || rs.width[2 * p + 1] != 4 || rs.stride[2 * p] != rs.stride[0] || rs.stride[2 * p + 1] != 4) return 0;
        if (rs.stride[0] & (w == 6 ? 1u : 3u)) return 0;
        if ((reinterpret_cast<uintptr_t>(rs.ptr[2 * p]) & (w == 6 ? 7u : 15u)) || (reinterpret_cast<uintptr_t>(rs.ptr[2 * p + 1]) & 15u)) return 0;
        a->mat[p] = rs.ptr[2 * p]; a->salt[p] = rs.ptr[2 * p + 1];
    }
    a->stride = rs.stride[0];
    return w;
}
// returns true when a salted-leaf kernel was launched for the tallest class
static bool launch_keccak_leaf_salted(hipStream_t stream, const RowSet& rs, uint64_t n_rows, uint32_t* digests) {
    const dim3 grid((uint32_t)((n_rows + 255) / 256)), block(256);
    SaltedRows<1> a1;
    SaltedRows<4> a4;
    if (const uint32_t w = salted_rows_of<1>(rs, &a1)) {
        if (w == 4) hipLaunchKernelGGL((keccak_leaf_salted_kernel<4, 1>), grid, block, 0, stream, a1, n_rows, digests);
        else if (w == 6) hipLaunchKernelGGL((keccak_leaf_salted_kernel<6, 1>), grid, block, 0, stream, a1, n_rows, digests);
        else hipLaunchKernelGGL((keccak_leaf_salted_kernel<8, 1>), grid, block, 0, stream, a1, n_rows, digests);
        return true;
    }
    if (salted_rows_of<4>(rs, &a4) == 4) {
        hipLaunchKernelGGL((keccak_leaf_salted_kernel<4, 4>), grid, block, 0, stream, a4, n_rows, digests);
        return true;
    }
    return false;
}

// WIDE rows under the Keccak sponge (as leaf_hash_f64_wide_kernel): the workgroup's 256 rows staged through LDS one rate block
// (17 u64 = 34 words) at a time, loaded in row order (a wave reads ~2 rows' 136 contiguous bytes per instruction instead of 64 lines
// 10 KB apart), read back by the row's lane as 16-byte pieces (row stride 36 words: conflict-free), the next block in flight
// during the permutation of the current one.
constexpr uint32_t KLEAF_WIDE_BLOCK = 34, KLEAF_WIDE_STRIDE = 36;
__global__ void __launch_bounds__(256) keccak_leaf_wide_kernel(const uint32_t* mat, uint32_t width, uint64_t n_rows, uint32_t* digests) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[256 * KLEAF_WIDE_STRIDE];
    const uint32_t tid = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * 256u, r = r0 + tid;
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    uint32_t pre[KLEAF_WIDE_BLOCK];
    auto fetch = [&](uint32_t c0) {
#pragma unroll
        for (uint32_t i = 0; i < KLEAF_WIDE_BLOCK; i++) {
            const uint32_t e = tid + 256u * i, row = e / KLEAF_WIDE_BLOCK, col = e - row * KLEAF_WIDE_BLOCK;
            pre[i] = (r0 + row < n_rows && c0 + col < width) ? mat[(r0 + row) * width + c0 + col] : 0u;
        }
    };
    fetch(0);
    for (uint32_t c0 = 0; c0 < width; c0 += KLEAF_WIDE_BLOCK) {
        __syncthreads();  // the previous block has been read by every lane
#pragma unroll
        for (uint32_t i = 0; i < KLEAF_WIDE_BLOCK; i++) {
            const uint32_t e = tid + 256u * i, row = e / KLEAF_WIDE_BLOCK, col = e - row * KLEAF_WIDE_BLOCK;
            tile[row * KLEAF_WIDE_STRIDE + col] = pre[i];
        }
        __syncthreads();
        const bool last = c0 + KLEAF_WIDE_BLOCK >= width;
        if (!last) fetch(c0 + KLEAF_WIDE_BLOCK);  // in flight during the permutation below
        const uint4* mine = reinterpret_cast<const uint4*>(tile + tid * KLEAF_WIDE_STRIDE);
        uint32_t w[36];
#pragma unroll
        for (int k = 0; k < 9; k++) { const uint4 q = mine[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
#pragma unroll
        for (int k = 0; k < 17; k++)
            if (c0 + 2 * k < width) st[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);  // words past the row's end were staged as 0
        if (last) kk::permute_digest(st);
        else kk::permute(st);
    }
    if (r >= n_rows) return;
    store_digest64(digests + r * 8, st);
}
// CompressionFunctionFromHasher<U64Hash, 2, 4>: hash of the 8 lanes of the two child digests (one block)
__global__ void __launch_bounds__(256) keccak_compress_kernel(const uint32_t* prev, uint32_t* next, uint64_t n_out,
                                                              RowSet rs, uint32_t inject) {
    if (gridDim.x <= 512u) P3_LATENCY_BOUND_KERNEL();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    uint32_t w[16];
    load_digest(prev + i * 16, w);
    load_digest(prev + i * 16 + 8, w + 8);
    uint64_t st[25];
#pragma unroll
    for (int k = 0; k < 8; k++) st[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
#pragma unroll
    for (int k = 8; k < 25; k++) st[k] = 0;
    kk::permute_digest(st);
    if (inject) {
        uint64_t h[25];
        keccak_sponge_row(rs, i, h);
#pragma unroll
        for (int k = 0; k < 4; k++) st[4 + k] = h[k];
#pragma unroll
        for (int k = 8; k < 25; k++) st[k] = 0;
        kk::permute_digest(st);
    }
    store_digest64(next + i * 8, st);
}
// Several levels of a Keccak tree per launch, one state per lane: each workgroup reduces a chunk of `chunk` (128 by
// default: one wave) consecutive digests of the current layer by `levels` levels inside LDS and writes every intermediate
// layer to HBM.  A level costs one permutation's issue time (~9 us); the layers of 2^13..2^15 digests come here, the
// smaller ones go to the lane-cooperative kernel below (mmcs_commit).
// layer0: consecutive layers in HBM (n_in digests, then n_in / 2, ...).
__global__ void __launch_bounds__(1024) keccak_tree_levels_kernel(uint32_t* layer0, uint32_t n_in, uint32_t chunk, uint32_t levels,
                                                                  uint32_t* root_copy) {
    P3_LATENCY_BOUND_KERNEL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x, blk = blockIdx.x;
    const uint4* src = reinterpret_cast<const uint4*>(layer0) + (size_t)blk * chunk * 2;
    for (uint32_t i = tid; i < chunk * 2; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = src[i];
    __syncthreads();
    uint32_t* out = layer0 + (size_t)n_in * 8;  // next layer
    uint32_t n_layer = n_in >> 1;               // its length
    for (uint32_t k = 0, n = chunk; k < levels; k++, n >>= 1) {
        const uint32_t half = n >> 1;
        uint64_t st[25];
        const bool act = tid < half;
        if (act) {
            const uint64_t* p = reinterpret_cast<const uint64_t*>(lds) + (size_t)tid * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) st[i] = p[i];
#pragma unroll
            for (int i = 8; i < 25; i++) st[i] = 0;
            kk::permute_digest(st);
        }
        __syncthreads();
        if (act) {
            uint64_t* p = reinterpret_cast<uint64_t*>(lds) + (size_t)tid * 4;
#pragma unroll
            for (int i = 0; i < 4; i++) p[i] = st[i];
            store_digest64(out + ((size_t)blk * half + tid) * 8, st);
            if (n_layer == 1 && root_copy) store_digest64(root_copy, st);
        }
        __syncthreads();
        out += (size_t)n_layer * 8;
        n_layer >>= 1;
    }
}
// The last levels of a Keccak tree with the lane-cooperative permutation (kk::f_coop, one compression per WAVE): a
// workgroup of up to 16 waves holds a chunk of <= 32 consecutive digests in LDS and reduces it by up to five levels,
// every intermediate layer written to HBM.  A level costs ~5 us instead of the ~9 us of the one-state-per-lane kernel
// above, at several times its lane-instructions — so only layers of <= 2^12 digests come here (mmcs_commit).
__global__ void __launch_bounds__(1024) keccak_tree_levels_coop_kernel(uint32_t* layer0, uint32_t n_in, uint32_t chunk, uint32_t levels,
                                                                       uint32_t* root_copy) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ uint64_t lds[32 * 4];
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, wv = tid >> 6, idx = kk::coop_index();
    if (tid < chunk * 4) lds[tid] = reinterpret_cast<const uint64_t*>(layer0)[(size_t)blk * chunk * 4 + tid];
    __syncthreads();
    uint64_t* out = reinterpret_cast<uint64_t*>(layer0) + (size_t)n_in * 4;  // next layer
    uint32_t n_layer = n_in >> 1;                                             // its length
    for (uint32_t k = 0, n = chunk; k < levels; k++, n >>= 1) {
        const uint32_t half = n >> 1;
        const bool act = wv < half;  // uniform over the wave
        uint64_t a = 0;
        if (act) {
            a = idx < 8u ? lds[wv * 8 + idx] : 0ull;
            a = kk::f_coop(a);
        }
        __syncthreads();
        if (act && idx < 4u) {
            lds[wv * 4 + idx] = a;
            out[((size_t)blk * half + wv) * 4 + idx] = a;
            if (n_layer == 1 && root_copy) reinterpret_cast<uint64_t*>(root_copy)[idx] = a;
        }
        __syncthreads();
        out += (size_t)n_layer * 4;
        n_layer >>= 1;
    }
}
__global__ void keccak_f_kernel(uint64_t* states, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t st[25];
#pragma unroll
    for (int k = 0; k < 25; k++) st[k] = states[i * 25 + k];
    kk::permute(st);
#pragma unroll
    for (int k = 0; k < 25; k++) states[i * 25 + k] = st[k];
}
int keccak_f_states(hipStream_t stream, uint64_t* d_states, uint64_t n) {
    if (!n) return OK;
    hipLaunchKernelGGL(keccak_f_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_states, n);
    P3_HIP(hipGetLastError());
    return OK;
}

// Gathers one opening (rows of every matrix + sibling path) into a packed staging buffer.
struct OpenArgs {
    const uint32_t* mat[64];
    uint32_t width[64];
    uint32_t stride[64];
    uint32_t shift[64];  // log_max_height - log_height
    uint32_t n_mats;
    uint32_t log_max_height;
};
__global__ void open_gather_kernel(OpenArgs a, const uint32_t* layers, uint64_t index, uint32_t* out) {
    P3_LATENCY_BOUND_KERNEL();
    // rows
    uint32_t off = 0;
    for (uint32_t m = 0; m < a.n_mats; m++) {
        uint64_t r = index >> a.shift[m];
        for (uint32_t c = threadIdx.x; c < a.width[m]; c += blockDim.x) out[off + c] = a.mat[m][r * a.stride[m] + c];
        off += a.width[m];
    }
    // siblings: layer i starts at sum_{j<i} (maxh >> j) * 8 words
    uint64_t base = 0, len = 1ull << a.log_max_height;
    for (uint32_t i = 0; i < a.log_max_height; i++) {
        uint64_t sib = (index >> i) ^ 1;
        if (threadIdx.x < 8) out[off + i * 8 + threadIdx.x] = layers[base + sib * 8 + threadIdx.x];
        base += len * 8;
        len >>= 1;
    }
}

static RowSet make_rowset(const Tree& t, uint64_t h) {
    RowSet rs{};
    for (size_t m = 0; m < t.mats.size(); m++)
        if (t.heights[m] == h) {
            rs.ptr[rs.count] = t.mats[m];
            rs.width[rs.count] = (uint32_t)t.widths[m];
            rs.stride[rs.count] = (uint32_t)t.strides[m];
            rs.total += (uint32_t)t.widths[m];
            rs.count++;
        }
    return rs;
}


static bool has_height(const Tree& t, uint64_t h) {
    for (size_t m = 0; m < t.mats.size(); m++)
        if (t.heights[m] == h) return true;
    return false;
}

Tree::~Tree() {
    if (layers && owns_layers) (void)hipFree(layers);
    if (staging) (void)hipFree(staging);
    for (void* p : owned) (void)hipFree(p);
}

int mmcs_commit(hipStream_t stream, const uint32_t* const* d_mats, const size_t* heights, const size_t* widths,
                size_t n_mats, Tree** out, uint32_t* ext_layers, uint32_t* root_copy, int kind, const size_t* strides, int profile) {
    if (kind != HASH_POSEIDON2 && kind != HASH_KECCAK) return fail(ERR_BAD_ARG, "mmcs_commit: unknown hash configuration");
    if (!n_mats || !d_mats || !heights || !widths || !out) return fail(ERR_BAD_ARG, "mmcs_commit: null/empty argument");
    if (n_mats > 64) return fail(ERR_BAD_ARG, "mmcs_commit: at most 64 matrices per commitment");
    uint64_t maxh = 0;
    for (size_t i = 0; i < n_mats; i++) {
        if (!is_pow2(heights[i])) return fail(ERR_BAD_ARG, "mmcs_commit: heights must be powers of two");
        if (widths[i] > 0xffffffffull) return fail(ERR_BAD_ARG, "mmcs_commit: width too large");
        if (heights[i] > maxh) maxh = heights[i];
    }
    std::unique_ptr<Tree> t(new Tree());
    t->kind = kind;
    for (size_t i = 0; i < n_mats; i++) {
        t->mats.push_back(d_mats[i]); t->heights.push_back(heights[i]); t->widths.push_back(widths[i]);
        const size_t st = strides ? strides[i] : widths[i];
        if (st < widths[i] || st > 0xffffffffull) return fail(ERR_BAD_ARG, "mmcs_commit: row stride below the width");
        t->strides.push_back(st);
    }
    t->log_max_height = log2u(maxh);
    for (uint64_t h = maxh; h >= 1; h >>= 1) {
        size_t cnt = 0;
        for (size_t i = 0; i < n_mats; i++) cnt += heights[i] == h;
        if (cnt > MAX_CLASS_MATS) return fail(ERR_BAD_ARG, "mmcs_commit: more than 16 matrices of one height");
        if (h == 1) break;
    }
    size_t total_digests = 2 * maxh - 1;
    size_t staging_words = 0;
    for (size_t i = 0; i < n_mats; i++) staging_words += widths[i];
    staging_words += (size_t)t->log_max_height * 8 + 8;
    t->staging_words = staging_words;
    if (ext_layers) { t->layers = ext_layers; t->owns_layers = false; }
    else P3_HIP(hipMalloc(reinterpret_cast<void**>(&t->layers), total_digests * 32));
    size_t off = 0;
    for (uint64_t len = maxh; len >= 1; len >>= 1) {
        t->layer_off.push_back(off);
        t->layer_len.push_back(len);
        off += len * 8;
        if (len == 1) break;
    }
    // ---- small-layer policy, by PROFILE (common.h; round 5: was a set of process-global environment switches) ----
    // Layers with fewer than 2^12 permutations cannot fill the chip with one state per lane: they run the 16-lanes-per-state
    // kernels (latency ~3x lower: ~3 us a level against ~8), 2^5 digests per workgroup = four waves, five levels per launch; the
    // last digests finish inside one workgroup.  (Below 2^15 instead of 2^12: a lone 2^20 proof 3.3 against 3.6 ms, but the
    // 16-lane form costs ~3.4x the VALU work and four provers lose 5 %; the quad form below took that range over.)
    constexpr uint64_t COOP_MAX = 1ull << 12;
    constexpr uint32_t COOP_CHUNK_LOG = 5;
    // LATENCY profile: one Poseidon2 state per DPP quad (poseidon2_q4.hip.h) for layers of 2^12 .. 2^15 permutations: 4-7 us a
    // launch instead of ~12 at 1.6x the lane-instructions: a lone 2^20 proof 3.53 -> 3.33-3.38 ms.  THROUGHPUT profile: off — four
    // concurrent provers LOSE 3.8 % with it (587 -> 565 proofs/s): the extra lane-instructions run at raised priority against the
    // other provers' hash layers (profiles/r04_latency_ab.txt).  At 2^16 permutations the quad form loses even alone.
    const uint64_t Q4_MAX = profile == PROFILE_LATENCY ? (1ull << 15) : 0;
    constexpr uint32_t Q4_PRIO = 1;
    // Cooperative Keccak levels (one state per wave, ~13x the lane-instructions of the per-lane form): layers of <= 2^10 digests
    // under the THROUGHPUT profile (four provers: 2^12: 624.6 proofs/s, 2^11: 633, 2^10: 647, 2^9: 636), <= 2^12 under the LATENCY
    // profile (a lone proof: -0.2 ms); 2^4 digests per workgroup
    const uint64_t KCOOP_IN = profile == PROFILE_LATENCY ? (1ull << 12) : (1ull << 10);
    constexpr uint32_t KCOOP_CHUNK_LOG = 4;
    // one-state-per-lane levels kernel: digests per workgroup.  A level costs one permutation's issue time (~9 us) per
    // wave a SIMD holds, so ONE wave per workgroup (128 digests) spreads a layer of <= 2^15 digests over the whole chip;
    // 2048 (sixteen waves on one CU) was 25 us per level
    constexpr uint64_t KLANE_CHUNK = 1ull << 7;
    if (kind == HASH_KECCAK) {
        // one state per lane for the large layers, the lane-cooperative form (shuffles inside a half-wave) for the small ones
        RowSet rs0 = make_rowset(*t, maxh);
        if (rs0.count == 1 && rs0.width[0] >= 68 && rs0.stride[0] == rs0.width[0])
            hipLaunchKernelGGL(keccak_leaf_wide_kernel, dim3((uint32_t)((maxh + 255) / 256)), dim3(256), 0, stream, rs0.ptr[0], rs0.width[0], maxh,
                               t->layers);
        else if (!launch_keccak_leaf_salted(stream, rs0, maxh, t->layers))
            hipLaunchKernelGGL(keccak_leaf_kernel, dim3((uint32_t)((maxh + 255) / 256)), dim3(256), 0, stream, rs0, maxh, t->layers);
        P3_HIP(hipGetLastError());
        for (size_t l = 1; l < t->layer_len.size(); l++) {
            uint64_t len = t->layer_len[l];
            RowSet rs = make_rowset(*t, len);
            if (len < COOP_MAX && !rs.count) {
                // as many injection-free levels as one launch may take.  Layers of <= KCOOP_IN digests (2^10 / 2^12 by profile) go through
                // the lane-cooperative kernel, 2^KCOOP_CHUNK_LOG digests per workgroup; above that one state per lane, chunks of KLANE_CHUNK
                // digests per workgroup, stopping where the cooperative kernel takes over
                const uint64_t n_in = t->layer_len[l - 1];
                const bool coop = n_in <= KCOOP_IN;
                uint32_t levels = 0;
                while (levels < (coop ? KCOOP_CHUNK_LOG : 11u) && l + levels < t->layer_len.size() && !has_height(*t, t->layer_len[l + levels])) levels++;
                if (!coop) while (levels > 1 && (n_in >> levels) < KCOOP_IN) levels--;
                const uint32_t chunk = (uint32_t)std::min<uint64_t>(n_in, coop ? 1ull << KCOOP_CHUNK_LOG : KLANE_CHUNK);  // per workgroup
                while ((1u << levels) > chunk) levels--;
                if (coop)
                    hipLaunchKernelGGL(keccak_tree_levels_coop_kernel, dim3((uint32_t)(n_in / chunk)), dim3(std::max<uint32_t>(64, chunk * 32)), 0,
                                       stream, t->layers + t->layer_off[l - 1], (uint32_t)n_in, chunk, levels, root_copy);
                else
                    hipLaunchKernelGGL(keccak_tree_levels_kernel, dim3((uint32_t)(n_in / chunk)), dim3(std::max<uint32_t>(64, chunk / 2)),
                                       (size_t)chunk * 32, stream, t->layers + t->layer_off[l - 1], (uint32_t)n_in, chunk, levels, root_copy);
                P3_HIP(hipGetLastError());
                if (t->layer_len[l + levels - 1] == 1) t->root_copied = root_copy != nullptr;
                l += levels - 1;  // the loop's l++ completes the step
                continue;
            }
            hipLaunchKernelGGL(keccak_compress_kernel, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, stream,
                               t->layers + t->layer_off[l - 1], t->layers + t->layer_off[l], len, rs, rs.count > 0 ? 1u : 0u);
            P3_HIP(hipGetLastError());
        }
        *out = t.release();
        return OK;
    }
    {
        RowSet rs = make_rowset(*t, maxh);
        const bool dense1 = rs.count == 1 && rs.stride[0] == rs.width[0];  // the single-matrix kernels take (pointer, width)
        if (dense1 && maxh < COOP_MAX && maxh * 16 <= 0x7fffffffull) {
            hipLaunchKernelGGL(leaf_coop_kernel, dim3((uint32_t)((maxh * 16 + 255) / 256)), dim3(256), 0, stream, rs.ptr[0],
                               rs.width[0], (uint32_t)maxh, t->layers);
        } else if (dense1 && maxh <= Q4_MAX && rs.width[0] <= 64 && (reinterpret_cast<uintptr_t>(t->layers) & 15u) == 0) {
            hipLaunchKernelGGL(leaf_hash_q4_kernel, dim3((uint32_t)((maxh * 4 + 255) / 256)), dim3(256), 0, stream, rs.ptr[0], rs.width[0], maxh,
                               t->layers, Q4_PRIO);
        } else if (dense1 && rs.width[0] >= 64) {
            hipLaunchKernelGGL(leaf_hash_f64_wide_kernel, dim3((uint32_t)((maxh + 255) / 256)), dim3(256), 0, stream, rs.ptr[0],
                               rs.width[0], maxh, t->layers);
        } else if (dense1) {
            hipLaunchKernelGGL(leaf_hash_f64_kernel, dim3((uint32_t)((maxh + 255) / 256)), dim3(256), 0, stream, rs.ptr[0],
                               rs.width[0], maxh, t->layers);
        } else if ((maxh >= COOP_MAX || (rs.count == 1 && !dense1))) {
            hipLaunchKernelGGL(leaf_hash_f64_rowset_kernel, dim3((uint32_t)((maxh + 255) / 256)), dim3(256), 0, stream, rs, maxh, t->layers);
        } else {
            hipLaunchKernelGGL(leaf_hash_kernel, dim3((uint32_t)((maxh + 255) / 256)), dim3(256), 0, stream, rs, maxh, t->layers);
        }
        P3_HIP(hipGetLastError());
    }
    for (size_t l = 1; l < t->layer_len.size();) {
        uint64_t len = t->layer_len[l];
        RowSet rs = make_rowset(*t, len);
        bool inject = rs.count > 0;
        if (!inject && len < COOP_MAX) {
            // as many injection-free levels as one launch may take (<= 7, stop before the next injected layer)
            uint32_t levels = 0;
            while (levels < COOP_CHUNK_LOG && l + levels < t->layer_len.size() && !has_height(*t, t->layer_len[l + levels])) levels++;
            uint64_t n_in = t->layer_len[l - 1];
            uint32_t chunk = (uint32_t)std::min<uint64_t>(n_in, 1ull << COOP_CHUNK_LOG);
            while ((1u << levels) > chunk) levels--;
            uint32_t blocks = (uint32_t)(n_in / chunk);
            uint32_t threads = std::max<uint32_t>(64, chunk * 8);
            bool makes_root = t->layer_len[l + levels - 1] == 1;
            hipLaunchKernelGGL(tree_levels_coop_kernel, dim3(blocks), dim3(threads), 0, stream, t->layers + t->layer_off[l - 1],
                               (uint32_t)n_in, chunk, levels, makes_root ? root_copy : nullptr);
            P3_HIP(hipGetLastError());
            if (makes_root) t->root_copied = root_copy != nullptr;
            l += levels;
            continue;
        }
        if (!inject && len <= Q4_MAX) {
            hipLaunchKernelGGL(compress_layer_q4_kernel, dim3((uint32_t)((len * 4 + 255) / 256)), dim3(256), 0, stream,
                               t->layers + t->layer_off[l - 1], t->layers + t->layer_off[l], len, Q4_PRIO);
        } else if (!inject) {
            hipLaunchKernelGGL(compress_layer_f64_kernel, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, stream,
                               t->layers + t->layer_off[l - 1], t->layers + t->layer_off[l], len);
        } else {
            hipLaunchKernelGGL(compress_layer_kernel, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, stream,
                               t->layers + t->layer_off[l - 1], t->layers + t->layer_off[l], len, rs, inject ? 1u : 0u);
        }
        P3_HIP(hipGetLastError());
        l++;
    }
    *out = t.release();
    return OK;
}

int mmcs_root(hipStream_t stream, const Tree& t, uint32_t root_out[8]) {
    P3_HIP(hipMemcpyAsync(root_out, t.layers + t.layer_off.back(), 32, hipMemcpyDeviceToHost, stream));
    P3_HIP(hipStreamSynchronize(stream));
    return OK;
}

int mmcs_open(hipStream_t stream, const Tree& tc, uint64_t index, uint32_t* rows_out, uint32_t* path_out) {
    Tree& t = const_cast<Tree&>(tc);
    if (index >> t.log_max_height) return fail(ERR_BAD_ARG, "mmcs_open: index out of range");
    if (!t.staging) P3_HIP(hipMalloc(reinterpret_cast<void**>(&t.staging), t.staging_words * 4));
    OpenArgs a{};
    a.n_mats = (uint32_t)t.mats.size();
    a.log_max_height = t.log_max_height;
    size_t row_words = 0;
    for (size_t m = 0; m < t.mats.size(); m++) {
        a.mat[m] = t.mats[m];
        a.width[m] = (uint32_t)t.widths[m];
        a.stride[m] = (uint32_t)t.strides[m];
        a.shift[m] = t.log_max_height - log2u(t.heights[m]);
        row_words += t.widths[m];
    }
    hipLaunchKernelGGL(open_gather_kernel, dim3(1), dim3(64), 0, stream, a, t.layers, index, t.staging);
    P3_HIP(hipGetLastError());
    std::vector<uint32_t> host(row_words + (size_t)t.log_max_height * 8);
    if (!host.empty()) {
        P3_HIP(hipMemcpyAsync(host.data(), t.staging, host.size() * 4, hipMemcpyDeviceToHost, stream));
        P3_HIP(hipStreamSynchronize(stream));
    }
    if (row_words) memcpy(rows_out, host.data(), row_words * 4);
    if (t.log_max_height) memcpy(path_out, host.data() + row_words, (size_t)t.log_max_height * 32);
    return OK;
}

}  // namespace p3
