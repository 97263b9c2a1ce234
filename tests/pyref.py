"""Pure-Python big-int statement of the path's mathematics, independent of both the C oracle
and the HIP product.  Used ONLY by tests/golden/make_golden.py to produce the committed
fixtures and by a few direct cross-checks; small sizes only (O(n^2) everywhere)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from gen_poseidon2_rc import gen_constants  # noqa: E402

P = 0x78000001
R = 1 << 32
RINV = pow(R, P - 2, P)
GEN = 31


def to_monty(x):
    return (x % P) * R % P


def from_monty(x):
    return x * RINV % P


def inv(x):
    return pow(x, P - 2, P)


def two_adic_generator(bits):
    return pow(pow(GEN, 15, P), 1 << (27 - bits), P)


def naive_dft_col(col):
    n = len(col)
    w = two_adic_generator(n.bit_length() - 1)
    return [sum(col[i] * pow(w, i * k, P) for i in range(n)) % P for k in range(n)]


def naive_dft(mat):
    """mat: list of rows (canonical ints); DFT of every column, natural order."""
    h, w = len(mat), len(mat[0])
    cols = [naive_dft_col([mat[r][c] for r in range(h)]) for c in range(w)]
    return [[cols[c][r] for c in range(w)] for r in range(h)]


def coset_lde(mat, added_bits, shift):
    """Evaluate the degree<h interpolant of each column (on the order-h subgroup) at
    shift * g^i, g = two_adic_generator(log h + added_bits); natural order."""
    h, w = len(mat), len(mat[0])
    logh = h.bit_length() - 1
    wh = two_adic_generator(logh)
    hinv = inv(h)
    out_n = h << added_bits
    g = two_adic_generator(logh + added_bits)
    res = [[0] * w for _ in range(out_n)]
    for c in range(w):
        col = [mat[r][c] for r in range(h)]
        coeffs = [sum(col[i] * pow(wh, (-i * j) % h, P) for i in range(h)) * hinv % P for j in range(h)]
        for i in range(out_n):
            x = shift * pow(g, i, P) % P
            acc = 0
            for cj in reversed(coeffs):
                acc = (acc * x + cj) % P
            res[i][c] = acc
    return res


def bitrev(i, bits):
    return int(bin(i)[2:].zfill(bits)[::-1], 2) if bits else 0


def fib_trace(a, b, n):
    rows, l, r = [], a % P, b % P
    for _ in range(n):
        rows.append([l, r])
        l, r = r, (l + r) % P
    return rows


def benchmark_input(h, w):
    return [[(17 * (r * w + c) + 3) % P for c in range(w)] for r in range(h)]


# ---------------- Poseidon2 (canonical ints) ----------------
_D = [-2, 1, 2, inv(2), 3, 4, -inv(2), -3, -4, inv(1 << 8), inv(4), inv(8), inv(1 << 27),
      -inv(1 << 8), -inv(16), -inv(1 << 27)]
_D = [d % P for d in _D]
DEFAULT_RC = gen_constants()


def _m4(x):
    a, b, c, d = x
    return [(2 * a + 3 * b + c + d) % P, (a + 2 * b + 3 * c + d) % P,
            (a + b + 2 * c + 3 * d) % P, (3 * a + b + c + 2 * d) % P]


def _ext(s):
    s = sum((_m4(s[i:i + 4]) for i in range(0, 16, 4)), [])
    sums = [sum(s[j] for j in range(k, 16, 4)) % P for k in range(4)]
    return [(s[i] + sums[i % 4]) % P for i in range(16)]


def _int(s):
    tot = sum(s) % P
    return [(s[i] * _D[i] + tot) % P for i in range(16)]


def poseidon2(state, rc=None):
    ei, it, ef = rc or DEFAULT_RC
    s = _ext(list(state))
    for r in range(4):
        s = _ext([pow((s[i] + ei[r][i]) % P, 7, P) for i in range(16)])
    for r in range(13):
        s[0] = pow((s[0] + it[r]) % P, 7, P)
        s = _int(s)
    for r in range(4):
        s = _ext([pow((s[i] + ef[r][i]) % P, 7, P) for i in range(16)])
    return s


def hash_row(items):
    st = [0] * 16
    for i in range(0, len(items), 8):
        chunk = items[i:i + 8]
        st[:len(chunk)] = chunk
        st = poseidon2(st)
    return st[:8]


def compress(l, r):
    return poseidon2(list(l) + list(r))[:8]


def merkle_layers(mats):
    """mats: list of row lists (canonical), power-of-two heights. Returns digest layers."""
    maxh = max(len(m) for m in mats)

    def rows_hash(h, i):
        items = []
        for m in mats:
            if len(m) == h:
                items += m[i]
        return hash_row(items)

    layers = [[rows_hash(maxh, i) for i in range(maxh)]]
    while len(layers[-1]) > 1:
        prev = layers[-1]
        n = len(prev) // 2
        inject = any(len(m) == n for m in mats)
        nxt = []
        for i in range(n):
            d = compress(prev[2 * i], prev[2 * i + 1])
            if inject:
                d = compress(d, rows_hash(n, i))
            nxt.append(d)
        layers.append(nxt)
    return layers


# Xoroshiro128Plus::seed_from_u64 (rand_xoshiro) — only to rebuild Plonky3's KAT constants.
_M64 = (1 << 64) - 1


class Xoroshiro128Plus:
    def __init__(self, seed):
        def splitmix(st):
            st = (st + 0x9E3779B97F4A7C15) & _M64
            z = st
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            return st, z ^ (z >> 31)
        st, self.s0 = splitmix(seed)
        st, self.s1 = splitmix(st)

    def next_u32(self):
        s0, s1 = self.s0, self.s1
        r = (s0 + s1) & _M64
        s1 ^= s0
        rotl = lambda x, k: ((x << k) | (x >> (64 - k))) & _M64  # noqa: E731
        self.s0 = rotl(s0, 24) ^ s1 ^ ((s1 << 16) & _M64)
        self.s1 = rotl(s1, 37)
        return r >> 32


def rng_rc(seed=1):
    """Poseidon2::new_from_rng_128 sampling order: ext-initial, ext-final, internal; each sample
    is `next_u32() >> 1` rejection-sampled below P and taken AS the Montgomery word."""
    rng = Xoroshiro128Plus(seed)

    def s():
        while True:
            v = rng.next_u32() >> 1
            if v < P:
                return from_monty(v)
    ei = [[s() for _ in range(16)] for _ in range(4)]
    ef = [[s() for _ in range(16)] for _ in range(4)]
    it = [s() for _ in range(13)]
    return ei, it, ef
