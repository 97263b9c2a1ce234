"""Host-side mirror of the reference's workload harness (native/src/fib_air.rs)."""
import ctypes as C

from . import _lib
from .gpu_dft import _stream_ptr

NUM_FIBONACCI_COLS = 2  # fib_air.rs:25


def generate_trace_rows(a, b, n, device="cuda"):
    """fib_air.rs:266-284 on the device: returns an (n, 2) int32 CUDA tensor of Montgomery words."""
    import torch
    if n & (n - 1):
        raise AssertionError("n must be a power of two")  # fib_air.rs:267 assert!(n.is_power_of_two())
    out = torch.empty((n, NUM_FIBONACCI_COLS), dtype=torch.int32, device=device)
    _lib.check(_lib.lib().p3hip_fib_trace_dev(a, b, n, C.c_void_p(out.data_ptr()), _stream_ptr()))
    return out


class FriParameters:
    """p3_fri::FriParameters.  Defaults: Plonky3's create_benchmark_fri_params (log_blowup 1,
    log_final_poly_len 0, 100 queries, 16 proof-of-work bits); the reference itself calls
    create_test_fri_params(challenge_mmcs, 2) (fib_air.rs:62)."""

    def __init__(self, log_blowup=1, log_final_poly_len=0, num_queries=100, proof_of_work_bits=16):
        self.log_blowup, self.log_final_poly_len = log_blowup, log_final_poly_len
        self.num_queries, self.proof_of_work_bits = num_queries, proof_of_work_bits

    def _c(self):
        arr = (C.c_uint32 * 4)(self.log_blowup, self.log_final_poly_len, self.num_queries, self.proof_of_work_bits)
        return arr


class FibAirProver:
    """prove(&config, &FibonacciAir{}, generate_trace_rows(a, b, 2^log_n), &[a, b, x]) (fib_air.rs:61-70)
    on the hip backend.  One instance = one HBM arena + one stream; use one per host thread."""

    def __init__(self, log_n, log_blowup=1, params=None, own_stream=True):
        self.params = params or FriParameters(log_blowup=log_blowup)
        self.log_n = log_n
        self._h = C.c_void_p()
        stream = None if own_stream else _stream_ptr()
        _lib.check(_lib.lib().p3hip_fib_prover_create(log_n, C.cast(self.params._c(), C.c_void_p), stream,
                                                      1 if own_stream else 0, C.byref(self._h)))

    def prove(self, a, b):
        """Returns the proof bytes (wire format: DESIGN.md)."""
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        _lib.check(_lib.lib().p3hip_fib_prover_prove(self._h, a, b, C.byref(out), C.byref(n)))
        return C.string_at(out, n.value)

    @staticmethod
    def proof_bytes(proof):
        return proof

    def stage_times(self, reset=True):
        ms = (C.c_double * 6)()
        cnt = C.c_uint64()
        _lib.check(_lib.lib().p3hip_fib_prover_stage_times(self._h, ms, C.byref(cnt), 1 if reset else 0))
        k = max(cnt.value, 1)
        names = ["trace_commit", "quotient_commit", "open", "fri_commit_phase", "grind", "queries"]
        return {"prover_" + nm: v / k for nm, v in zip(names, ms)}

    def stage_breakdown(self):
        self.stage_times(reset=True)
        for i in range(3):
            self.prove(i, i + 1)
        return self.stage_times(reset=True)

    def close(self):
        if self._h:
            _lib.lib().p3hip_fib_prover_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
