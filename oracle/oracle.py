"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes/numpy front end of oracle/_build/libp3oracle.so (the plain-C CPU restatement of the
reference's NTT/LDE + Poseidon2-MMCS path).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product (plonky3-mobile_amd/)
never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libp3oracle.so")
P = 0x78000001
_u32p = C.POINTER(C.c_uint32)


def build(force=False):
    """Compile the C restatement with gcc (make)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(s) <= os.path.getmtime(_LIB_PATH) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None
_FLAGS = "-O2 -fopenmp (portable build, oracle/Makefile CFLAGS)"


def build_flags():
    """Compiler flags of the library currently loaded (reported in bench.py's cpu_baseline)."""
    return _FLAGS


def use_native():
    """bench.py's cpu_baseline leg only: switch to a -O3 -march=native build of the same sources, compiled on THIS
    host (keyed by its CPU flags, so a build made on another machine is never loaded).  Falls back to the portable
    build (and says so in build_flags()) when the compiler is missing."""
    global _lib, _LIB_PATH, _FLAGS
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((ln for ln in f if ln.startswith("flags")), "")
    except OSError:
        flags = ""
    key = hashlib.sha1(flags.encode()).hexdigest()[:12]
    ndir = os.path.join(_HERE, "_build", "native-" + key)
    path = os.path.join(ndir, "libp3oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    try:
        if not (os.path.exists(path) and all(os.path.getmtime(s) <= os.path.getmtime(path) for s in srcs)):
            subprocess.check_call(["make", "-s", "-C", _HERE, "native", "NATIVE_DIR=" + ndir])
    except Exception as e:  # keep the baseline alive on the portable build
        _FLAGS = "-O2 -fopenmp (portable build; the native build failed: %s)" % type(e).__name__
        return False
    _LIB_PATH, _lib = path, None
    _FLAGS = "-O3 -march=native -fopenmp (built on the timing host, oracle/Makefile `native`)"
    lib()
    return True


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.p3o_set_threads(C.c_int(1))  # serial by default, like the reference's build (no `parallel` feature)
        for name in ("p3o_to_monty", "p3o_from_monty", "p3o_inv"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_uint32]
        for name in ("p3o_add", "p3o_sub", "p3o_mul"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_uint32, C.c_uint32]
        L.p3o_pow.restype = C.c_uint32
        L.p3o_pow.argtypes = [C.c_uint32, C.c_uint64]
        L.p3o_two_adic_generator.restype = C.c_uint32
        L.p3o_two_adic_generator.argtypes = [C.c_uint]
        L.p3o_mmcs_commit.restype = C.c_void_p
        L.p3o_mmcs_commit_kind.restype = C.c_void_p
        L.p3o_tree_layer.restype = _u32p
        L.p3o_tree_layer.argtypes = [C.c_void_p, C.c_size_t]
        for name in ("p3o_tree_num_layers", "p3o_tree_log_max_height"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_void_p]
        L.p3o_tree_layer_len.restype = C.c_size_t
        L.p3o_tree_layer_len.argtypes = [C.c_void_p, C.c_size_t]
        L.p3o_mmcs_free.argtypes = [C.c_void_p]
        L.p3o_free.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(_u32p)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def to_monty(a):
    """canonical -> Montgomery (numpy, vectorised)."""
    a = np.asarray(a, dtype=np.uint64) % P
    return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


def from_monty(a):
    a = np.asarray(a, dtype=np.uint64)
    rinv = pow(1 << 32, P - 2, P)
    return ((a * np.uint64(rinv)) % np.uint64(P)).astype(np.uint32)


def twiddle_table(log_n):
    out = np.zeros((1 << log_n) - 1, dtype=np.uint32)
    if log_n:
        lib().p3o_twiddle_table(C.c_uint(log_n), _p(out))
    return out


def _mat_call(fn, mat, out_rows, *extra):
    mat = _u32(mat)
    h, w = mat.shape
    out = np.zeros((out_rows, w), dtype=np.uint32)
    rc = fn(_p(mat), _p(out), C.c_size_t(h), C.c_size_t(w), *extra)
    if rc:
        raise ValueError("oracle: power-of-two height required, got %d" % h)
    return out


def naive_dft(mat):
    mat = _u32(mat)
    out = np.zeros_like(mat)
    lib().p3o_naive_dft(_p(mat), _p(out), C.c_size_t(mat.shape[0]), C.c_size_t(mat.shape[1]))
    return out


def dft_batch(mat):
    return _mat_call(lib().p3o_dft_batch, mat, np.shape(mat)[0])


def idft_batch(mat):
    return _mat_call(lib().p3o_idft_batch, mat, np.shape(mat)[0])


def coset_dft_batch(mat, shift_monty):
    return _mat_call(lib().p3o_coset_dft_batch, mat, np.shape(mat)[0], C.c_uint32(int(shift_monty)))


def coset_lde_batch(mat, added_bits, shift_monty, bit_reversed_out=False):
    return _mat_call(lib().p3o_coset_lde_batch, mat, np.shape(mat)[0] << added_bits,
                     C.c_uint(added_bits), C.c_uint32(int(shift_monty)), C.c_int(int(bit_reversed_out)))


def bit_reverse_rows(mat):
    mat = _u32(mat)
    out = np.zeros_like(mat)
    lib().p3o_bit_reverse_rows(_p(out), _p(mat), C.c_size_t(mat.shape[0]), C.c_size_t(mat.shape[1]))
    return out


def poseidon2_permute(state, rc=None):
    st = _u32(state).copy()
    assert st.shape == (16,)
    if rc is None:
        lib().p3o_poseidon2_permute(_p(st))
    else:
        ei, it, ef = (_u32(x) for x in rc)
        lib().p3o_poseidon2_permute_rc(_p(st), _p(ei), _p(it), _p(ef))
    return st


def hash_row(items):
    items = _u32(items).reshape(-1)
    out = np.zeros(8, dtype=np.uint32)
    lib().p3o_hash_row(_p(items), C.c_size_t(items.size), _p(out))
    return out


def compress(left, right):
    out = np.zeros(8, dtype=np.uint32)
    lib().p3o_compress(_p(_u32(left)), _p(_u32(right)), _p(out))
    return out


class Tree:
    """MerkleTreeMmcs prover data: keeps the matrices alive, exposes digest layers."""

    def __init__(self, mats, kind=0):
        self.kind = kind
        self.mats = [_u32(m) for m in mats]
        n = len(self.mats)
        ptrs = (_u32p * n)(*[_p(m) for m in self.mats])
        hs = (C.c_size_t * n)(*[m.shape[0] for m in self.mats])
        ws = (C.c_size_t * n)(*[m.shape[1] for m in self.mats])
        self.root = np.zeros(8, dtype=np.uint32)
        self._h = lib().p3o_mmcs_commit_kind(C.c_int(kind), ptrs, hs, ws, C.c_size_t(n), _p(self.root))
        if not self._h:
            raise ValueError("oracle mmcs: power-of-two heights required")
        self.log_max_height = lib().p3o_tree_log_max_height(self._h)

    def layers(self):
        L = lib()
        out = []
        for l in range(L.p3o_tree_num_layers(self._h)):
            n = L.p3o_tree_layer_len(self._h, l)
            out.append(np.ctypeslib.as_array(L.p3o_tree_layer(self._h, l), shape=(n, 8)).copy())
        return out

    def open_batch(self, index):
        tot = sum(m.shape[1] for m in self.mats)
        rows = np.zeros(max(tot, 1), dtype=np.uint32)
        path = np.zeros((max(self.log_max_height, 1), 8), dtype=np.uint32)
        rc = lib().p3o_mmcs_open_batch(C.c_void_p(self._h), C.c_size_t(index), _p(rows), _p(path))
        if rc:
            raise IndexError(index)
        return rows[:tot], path[: self.log_max_height]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().p3o_mmcs_free(C.c_void_p(self._h))
            self._h = None


HASH_POSEIDON2, HASH_KECCAK = 0, 1


def mmcs_commit(mats, kind=HASH_POSEIDON2):
    """kind HASH_KECCAK = the reference's own hash configuration (native/src/fib_air.rs:28-38, keccak.c)."""
    t = Tree(mats, kind)
    return t.root.copy(), t


def keccak_f(state):
    """Keccak-f[1600] on 25 u64 lanes (numpy uint64[25]) -> new array."""
    a = np.ascontiguousarray(state, dtype=np.uint64).copy()
    lib().p3o_keccak_f(a.ctypes.data_as(C.c_void_p))
    return a


def keccak_hash_row(items):
    out = np.zeros(8, dtype=np.uint32)
    items = _u32(items).reshape(-1)
    lib().p3o_keccak_hash_row(_p(items), C.c_size_t(items.size), _p(out))
    return out


def keccak_compress(left, right):
    out = np.zeros(8, dtype=np.uint32)
    lib().p3o_keccak_compress(_p(_u32(left)), _p(_u32(right)), _p(out))
    return out


def mmcs_verify_batch(root, dims, index, rows, path, kind=HASH_POSEIDON2):
    n = len(dims)
    hs = (C.c_size_t * n)(*[d[0] for d in dims])
    ws = (C.c_size_t * n)(*[d[1] for d in dims])
    rows = _u32(rows).reshape(-1)
    path = _u32(path).reshape(-1, 8)
    return lib().p3o_mmcs_verify_batch_kind(C.c_int(kind), _p(_u32(root)), hs, ws, C.c_size_t(n), C.c_size_t(index),
                                            _p(rows), _p(path), C.c_size_t(path.shape[0])) == 0


def set_threads(n):
    """Threads of the oracle's OpenMP loops; 1 reproduces the reference's serial CPU build."""
    lib().p3o_set_threads(C.c_int(int(n)))


def max_threads():
    return int(lib().p3o_max_threads())


def test_threads():
    """Threads for the big oracle proofs of the GPU tests: the cores the host REALLY has for this job — at most 16, a one-GPU box's CPU
    share.  (The box shows 256 logical cores; with 256 OpenMP threads on that share a 2^20-row proof takes 35 s, with 16 it takes 1.7 s:
    tools/oracle_thread_sweep.py.)"""
    return max(1, min(max_threads(), 16))


# ---- fib_air prover / verifier (stark.c) ----
class FriParams:
    """p3_fri::FriParameters; defaults = Plonky3's create_benchmark_fri_params (log_blowup 1,
    log_final_poly_len 0, 100 queries, 16 proof-of-work bits) [UPSTREAM-RECALL]."""

    def __init__(self, log_blowup=1, log_final_poly_len=0, num_queries=100, proof_of_work_bits=16):
        self.log_blowup, self.log_final_poly_len = log_blowup, log_final_poly_len
        self.num_queries, self.proof_of_work_bits = num_queries, proof_of_work_bits

    def astuple(self):
        return (self.log_blowup, self.log_final_poly_len, self.num_queries, self.proof_of_work_bits)


def prove_fib_air(a, b, log_n, params=None, hash=HASH_POSEIDON2):
    """hash=HASH_KECCAK: the reference's own hash configuration (Keccak MMCS + SerializingChallenger32 over a
    Keccak-256 HashChallenger, fib_air.rs:28-53), non-hiding."""
    params = params or FriParams()
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    L = lib()
    rc = L.p3o_prove_fib_air_hash(C.c_int(hash), C.c_uint64(a), C.c_uint64(b), C.c_uint(log_n),
                                  *[C.c_uint(v) for v in params.astuple()], C.byref(out), C.byref(n))
    if rc:
        raise ValueError("oracle prove_fib_air: bad parameters")
    data = C.string_at(out, n.value)
    L.p3o_free(out)
    return data


def verify_fib_air(proof, a, b, x, log_n, params=None, hash=HASH_POSEIDON2):
    """0 = accept, otherwise the code of the failed check."""
    params = params or FriParams()
    buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
    return lib().p3o_verify_fib_air_hash(C.c_int(hash), buf, C.c_size_t(len(proof)), C.c_uint64(a), C.c_uint64(b),
                                         C.c_uint64(x), C.c_uint(log_n), *[C.c_uint(v) for v in params.astuple()])


def prove_fib_air_hiding(a, b, log_n, params=None, hash=HASH_POSEIDON2, seed=1):
    """The reference's hiding configuration (fib_air.rs:40-65): MerkleTreeHidingMmcs + HidingFriPcs with
    SmallRng::seed_from_u64(seed); wire format version 2.  [UPSTREAM-RECALL], parity unpinned."""
    params = params or FriParams()
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    L = lib()
    rc = L.p3o_prove_fib_air_hiding(C.c_int(hash), C.c_uint64(a), C.c_uint64(b), C.c_uint(log_n),
                                    *[C.c_uint(v) for v in params.astuple()], C.c_uint64(seed), C.byref(out), C.byref(n))
    if rc:
        raise ValueError("oracle prove_fib_air_hiding: bad parameters")
    data = C.string_at(out, n.value)
    L.p3o_free(out)
    return data


def verify_fib_air_hiding(proof, a, b, x, log_n, params=None, hash=HASH_POSEIDON2):
    """0 = accept, otherwise the code of the failed check."""
    params = params or FriParams()
    buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
    return lib().p3o_verify_fib_air_hiding(C.c_int(hash), buf, C.c_size_t(len(proof)), C.c_uint64(a), C.c_uint64(b),
                                           C.c_uint64(x), C.c_uint(log_n), *[C.c_uint(v) for v in params.astuple()])


def rng_seed_from_u64(seed):
    s = (C.c_uint64 * 4)()
    lib().p3o_rng_seed_from_u64(s, C.c_uint64(seed))
    return s


def rng_next_u64(state):
    lib().p3o_rng_next_u64.restype = C.c_uint64
    return int(lib().p3o_rng_next_u64(state))


def rng_fill_field(state, n):
    out = np.zeros(max(n, 1), dtype=np.uint32)
    lib().p3o_rng_fill_field(state, _p(out), C.c_size_t(n))
    return out[:n]


def keccak256(data):
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
    lib().p3o_keccak256(buf, C.c_size_t(len(data)), out)
    return bytes(out)


def fib_public_x(a, b, n):
    """canonical value of the last row's right column (the public value x, fib_air.rs:57,68)."""
    l, r = a % P, b % P
    for _ in range(n - 1):
        l, r = r, (l + r) % P
    return r


# ---- workload generators (reference native/src/fib_air.rs) ----
def benchmark_input(height, width):
    """fib_air.rs:77-86 benchmark_input: v_i = (17 i + 3) mod P, as Montgomery words."""
    i = np.arange(height * width, dtype=np.uint64)
    return to_monty((i * np.uint64(17) + np.uint64(3)) % np.uint64(P)).reshape(height, width)


def generate_trace_rows(a, b, n):
    """fib_air.rs:266-284 generate_trace_rows: row0=(a,b); row i=(right, left+right)."""
    out = np.zeros((n, 2), dtype=np.uint64)
    l, r = a % P, b % P
    for i in range(n):
        out[i, 0], out[i, 1] = l, r
        l, r = r, (l + r) % P
    return to_monty(out)
