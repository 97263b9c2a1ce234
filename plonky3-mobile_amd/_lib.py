"""ctypes binding of libp3hip.so (include/p3hip.h).  There is no fallback: if the HIP library is
missing or a call fails, this raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P3HIP_LIB") or os.path.join(_HERE, "libp3hip.so")  # P3HIP_LIB: experiment builds
ROOT = os.path.dirname(_HERE)

u32p = C.POINTER(C.c_uint32)
_SIGS = {
    "p3hip_set_backend": (C.c_int, [C.c_char_p]),
    "p3hip_get_backend": (C.c_int, []),
    "p3hip_is_available": (C.c_int, [C.c_char_p, C.c_size_t]),
    "p3hip_take_last_error": (C.c_char_p, []),
    "p3hip_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "p3hip_free": (C.c_int, [C.c_void_p]),
    "p3hip_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "p3hip_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "p3hip_sync": (C.c_int, [C.c_void_p]),
    "p3hip_last_timing_line": (C.c_char_p, []),
    "p3hip_release_thread_context": (None, []),
    "p3hip_dft_batch_bb31": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "p3hip_idft_batch_bb31": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "p3hip_coset_dft_batch_bb31": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32]),
    "p3hip_coset_lde_batch_bb31": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint,
                                             C.c_uint32, C.c_int]),
    "p3hip_dft_batch_bb31_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "p3hip_idft_batch_bb31_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "p3hip_coset_dft_batch_bb31_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                                 C.c_void_p]),
    "p3hip_coset_lde_batch_bb31_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint,
                                                 C.c_uint32, C.c_int, C.c_void_p]),
    "p3hip_coset_lde_from_coeffs_bb31_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint32,
                                                       C.c_void_p]),
    "p3hip_bit_reverse_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "p3hip_dft_plan_bb31": (C.c_int, [C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]),
    "p3hip_fib_trace_dev": (C.c_int, [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p]),
    "p3hip_poseidon2_permute_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "p3hip_poseidon2_permute": (C.c_int, [C.c_void_p, C.c_size_t]),
    "p3hip_poseidon2_permute_variant_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "p3hip_poseidon2_f64_probe_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "p3hip_mmcs_commit_dev": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                        C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "p3hip_mmcs_commit_async_dev": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_void_p),
                                              C.c_void_p]),
    "p3hip_mmcs_commit_hash_dev": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                             C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "p3hip_mmcs_commit_hash": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                         C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "p3hip_mmcs_layer_words": (C.c_size_t, [C.c_size_t]),
    "p3hip_mmcs_commit_into_dev": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                             C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "p3hip_keccak_f_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "p3hip_mmcs_root": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "p3hip_mmcs_log_max_height": (C.c_size_t, [C.c_void_p]),
    "p3hip_mmcs_num_layers": (C.c_size_t, [C.c_void_p]),
    "p3hip_mmcs_layer_dev": (C.c_void_p, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "p3hip_mmcs_open_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "p3hip_mmcs_free": (None, [C.c_void_p]),
    "p3hip_set_thread_profile": (C.c_int, [C.c_int]),
    "p3hip_get_thread_profile": (C.c_int, []),
    "p3hip_fib_prover_create_profile": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.POINTER(C.c_void_p)]),
    "p3hip_fib_prover_create": (C.c_int, [C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "p3hip_fib_prover_create_hash": (C.c_int, [C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "p3hip_verify_fib_air_hash": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint,
                                            C.c_void_p]),
    "p3hip_fib_prover_prove": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.POINTER(C.c_uint8)),
                                         C.POINTER(C.c_size_t)]),
    "p3hip_fib_prover_prove_into": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "p3hip_fib_prover_stage_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "p3hip_fib_prover_enqueue": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "p3hip_fib_prover_finish": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]),
    "p3hip_fib_prover_destroy": (None, [C.c_void_p]),
    "p3hip_run_fib_air_zk": (C.c_int, [C.c_char_p, C.c_size_t]),
    "p3hip_run_dft_benchmark": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]),
    "p3hip_fib_prover_grind_miss_probe": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t,
                                                    C.POINTER(C.c_size_t)]),
    "p3hip_fib_prover_create_hiding": (C.c_int, [C.c_int, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "p3hip_verify_fib_air_hiding": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint,
                                              C.c_void_p]),
    "p3hip_rng_create": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "p3hip_rng_fill_field_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "p3hip_rng_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "p3hip_rng_destroy": (None, [C.c_void_p]),
    "p3hip_mmcs_commit_hiding_dev": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                               C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "p3hip_fib_batch_create": (C.c_int, [C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_void_p)]),
    "p3hip_fib_batch_create_hash": (C.c_int, [C.c_int, C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_void_p)]),
    "p3hip_fib_batch_create_hiding": (C.c_int, [C.c_int, C.c_uint, C.c_void_p, C.c_uint64, C.c_uint, C.POINTER(C.c_void_p)]),
    "p3hip_fib_batch_prove": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]),
    "p3hip_fib_batch_submit": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "p3hip_fib_batch_collect": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]),
    "p3hip_fib_batch_destroy": (None, [C.c_void_p]),
    "p3hip_verify_fib_air": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_void_p]),
    "p3hip_mmcs_commit": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                    C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
}


class P3HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libp3hip error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libp3hip.so (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", csrc])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libp3hip.so is not built (run __graft_entry__.build()); "
                              "there is no CPU fallback in this package")
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 / libhsa-runtime64; load it
        # FIRST so libp3hip's NEEDED libamdhip64.so.7 resolves to the already-loaded copy.  Loading the
        # system runtime first and torch's second leaves torch with "No HIP GPUs are available".
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_sha = None


def lib_sha256():
    """SHA-256 of the libp3hip.so this process loads: the build identity the committed PMC profiles are keyed by (a counter
    figure divided by a time measured in this run describes this run only if both come from the same code)."""
    global _sha
    if _sha is None:
        import hashlib
        with open(LIB_PATH, "rb") as f:
            _sha = hashlib.sha256(f.read()).hexdigest()
    return _sha


_src_sha = None


def src_sha256():
    """SHA-256 over the library's SOURCES (csrc/*.hip, *.h, *.inc, the Makefile, include/p3hip.h; names and contents, sorted): the build
    identity that survives a rebuild on another machine.  The committed PMC profiles carry it next to the binary's hash."""
    global _src_sha
    if _src_sha is None:
        import glob
        import hashlib
        csrc = os.path.join(_HERE, "csrc")
        files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.inc")) +
                       [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "p3hip.h")])
        h = hashlib.sha256()
        for f in files:
            h.update(os.path.relpath(f, ROOT).encode() + b"\0")
            with open(f, "rb") as fh:
                h.update(fh.read())
            h.update(b"\0")
        _src_sha = h.hexdigest()
    return _src_sha


def declared_symbols():
    return sorted(_SIGS)


def take_last_error():
    """gpu_dft.rs:65-68 take_last_vulkan_error: returns and clears the pending message."""
    m = lib().p3hip_take_last_error()
    return m.decode() if m else None


def check(rc):
    if rc != 0:
        raise P3HipError(rc, take_last_error() or "")
