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
