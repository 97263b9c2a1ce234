"""plonky3-mobile_amd — MI355X (gfx950) backend for the Plonky3 fib_air hot path:
batched BabyBear NTT / coset LDE and the Poseidon2 Merkle-tree MMCS, as hand-written HIP kernels
behind the C ABI in include/p3hip.h.  This package is the host-side mirror of the reference's
selector (native/src/gpu_dft.rs) and of Plonky3's Mmcs contract; it has no CPU fallback."""
from . import _lib
from ._lib import P3HipError, build
from .gpu_dft import (BackendKind, GpuDft, bit_reverse_rows, coset_lde_from_coeffs, dev_u32, get_backend_kind, host_u32, is_available, last_timing_line,
                      set_backend_kind, set_backend_kind_from_str, take_last_error, GENERATOR_MONTY, MONTY_ONE, P)
from .fib_air import (BENCHMARK_CASES, DeviceRng, FibAirBatchProver, FibAirProver, FriParameters, benchmark_input, fib_public_x,
                      generate_trace_rows, get_thread_profile, percentile_ms, run_dft_benchmark, run_dft_benchmark_report, run_fib_air, run_fib_air_zk_report,
                      set_thread_profile, verify_fib_air)
from .mmcs import MerkleTree, MerkleTreeHidingMmcs, MerkleTreeMmcs, keccak_f, poseidon2_permute
from . import plan

__all__ = ["BackendKind", "GpuDft", "MerkleTree", "MerkleTreeMmcs", "MerkleTreeHidingMmcs", "P3HipError", "bit_reverse_rows", "coset_lde_from_coeffs", "build",
           "dev_u32", "generate_trace_rows", "FibAirProver", "FibAirBatchProver", "DeviceRng", "FriParameters", "BENCHMARK_CASES", "benchmark_input", "percentile_ms", "run_dft_benchmark", "run_dft_benchmark_report", "run_fib_air", "run_fib_air_zk_report", "verify_fib_air", "fib_public_x", "get_backend_kind", "get_thread_profile", "set_thread_profile", "host_u32", "is_available", "last_timing_line", "keccak_f", "poseidon2_permute", "set_backend_kind",
           "set_backend_kind_from_str", "take_last_error", "GENERATOR_MONTY", "MONTY_ONE", "P"]
