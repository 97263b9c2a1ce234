"""Host-side mirror of the reference's plan / parameter helpers (native/src/backend_vulkan.rs:784-1031), for callers
written against that surface (the reference's own benchmark drives the raw-u32 entry, fib_air.rs:128-134):

  FftStageParams / params_for_stage    backend_vulkan.rs:784-808   width, height, stage, log_n, twiddle_base
  ComputePlan / prepare_compute_plan   :959-975                    params + the LAUNCH PLAN of the backend
  twiddles_for_stage / twiddle_table   :977-996                    stage s at offset 2^s - 1, H - 1 words in all
  reverse_bits_len / write_bit_reversed_rows_u32   :998-1026
  setup_pipeline_plan(plan, words)     :1028-1031                  Montgomery words in natural row order in and out

The reference's VulkanComputePlan carries the workgroup counts of ONE stage dispatch (the host loops log2(height) of them)
and the length of a SPIR-V blob.  Neither exists here: `ComputePlan.dispatch` is what libp3hip will actually launch for the
shape — the radix-2 stages of each LDS-tiled pass, one kernel launch per entry (p3hip_dft_plan_bb31, host-only) — and the
device never streams the H - 1 word twiddle table; twiddle_table / write_bit_reversed_rows_u32 below are the reference's
layouts as pure integer code, used by the tests that pin the CPU restatement to them."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from .gpu_dft import BackendKind, GpuDft

P = 0x78000001
_GEN_27 = pow(31, 15, P)  # two_adic_generator(27) = 31^15


def _to_monty(v):
    return (int(v) << 32) % P


def two_adic_generator(bits):
    """canonical generator of the order-2^bits subgroup: (31^15)^(2^(27-bits))"""
    if bits > 27:
        raise ValueError("BabyBear two-adicity is 27")
    return pow(_GEN_27, 1 << (27 - bits), P)


@dataclass
class FftStageParams:  # backend_vulkan.rs:784-795 (the three padding words of the #[repr(C)] uniform block have no meaning here)
    width: int
    height: int
    stage: int
    log_n: int
    twiddle_base: int


def params_for_stage(width, height, stage, log_n, twiddle_base):
    return FftStageParams(width, height, stage, log_n, twiddle_base)


def launch_plan(height, width):
    """radix-2 stages per pass (= per kernel launch) of dft_batch / idft_batch for a height x width matrix, from the library"""
    from . import _lib
    n = C.c_size_t(0)
    buf = (C.c_uint32 * 8)()
    _lib.check(_lib.lib().p3hip_dft_plan_bb31(height, width, buf, 8, C.byref(n)))
    return tuple(int(buf[i]) for i in range(min(n.value, 8)))


@dataclass
class ComputePlan:
    params: FftStageParams
    dispatch: tuple  # the hip backend's launch plan: stages per LDS-tiled pass, one launch each (sum = log2 height)


def prepare_compute_plan(width, height, stage, log_n):
    params = params_for_stage(width, height, stage, log_n, 1)
    return ComputePlan(params, launch_plan(height, width))


def twiddles_for_stage(log_n, stage):
    """step = root^(2^(log_n - stage - 1)); entries step^0 .. step^(2^stage - 1) as Montgomery words"""
    half = 1 << stage
    step = pow(two_adic_generator(log_n), 1 << (log_n - stage - 1), P)
    out, acc = np.empty(half, dtype=np.uint32), 1
    for i in range(half):
        out[i] = _to_monty(acc)
        acc = acc * step % P
    return out


def twiddle_table(log_n):
    if log_n == 0:
        return np.zeros(0, dtype=np.uint32)
    return np.concatenate([twiddles_for_stage(log_n, s) for s in range(log_n)])


def reverse_bits_len(x, bits):
    y = 0
    for _ in range(bits):
        y = (y << 1) | (x & 1)
        x >>= 1
    return y


def write_bit_reversed_rows_u32(src, width):
    """dst[r] = src[bitrev(r)], whole rows; non power-of-two heights are copied unchanged (backend_vulkan.rs:1009-1012)"""
    src = np.ascontiguousarray(src, dtype=np.uint32).reshape(-1)
    if width == 0 or src.size == 0:
        return src.copy()
    height = src.size // width
    if height == 0 or height & (height - 1):
        return src.copy()
    bits = height.bit_length() - 1
    idx = np.array([reverse_bits_len(r, bits) for r in range(height)], dtype=np.int64)
    return src.reshape(height, width)[idx].reshape(-1)


def setup_pipeline_plan(plan, words):
    """setup_vulkan_pipeline_plan(&plan, &[u32]) -> Result<Vec<u32>, String>: the DFT of the height x width matrix of
    Montgomery words (natural row order in and out).  Errors surface as P3HipError, never as a CPU fallback."""
    w, h = plan.params.width, plan.params.height
    words = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1)
    if words.size != w * h:
        raise ValueError("input length %d does not match %d x %d" % (words.size, h, w))
    if h == 0 or w == 0:
        return words.copy()
    out = GpuDft.with_backend(BackendKind.Hip).dft_batch(words.reshape(h, w))
    return np.asarray(out, dtype=np.uint32).reshape(-1)
