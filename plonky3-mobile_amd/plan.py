"""Host-side mirror of the reference's plan / parameter helpers (native/src/backend_vulkan.rs:784-1031), for callers
written against that surface (the reference's own benchmark drives the raw-u32 entry, fib_air.rs:128-134):

  FftStageParams / params_for_stage    backend_vulkan.rs:784-808   the 32-byte #[repr(C)] parameter block
  ComputePlan / prepare_compute_plan   :959-975                    params + dispatch dimensions
  dispatch_dims                        :818-839
  twiddles_for_stage / twiddle_table   :977-996                    stage s at offset 2^s - 1, H - 1 words in all
  reverse_bits_len / write_bit_reversed_rows_u32   :998-1026
  setup_pipeline_plan(plan, words)     :1028-1031                  Montgomery words in natural row order in and out

On MI355X the per-stage dispatch loop these describe does not exist: setup_pipeline_plan hands the whole transform to
libp3hip (1-3 LDS-tiled passes), and the device never streams the H - 1 word table — the host functions below
exist for parity of the surface and are pure integer code (no device)."""
import struct
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
class FftStageParams:  # #[repr(C)], 32 bytes
    width: int
    height: int
    stage: int
    log_n: int
    twiddle_base: int
    _pad0: int = 0
    _pad1: int = 0
    _pad2: int = 0

    def pack(self):
        return struct.pack("<8I", self.width, self.height, self.stage, self.log_n, self.twiddle_base, 0, 0, 0)


def params_for_stage(width, height, stage, log_n, twiddle_base):
    return FftStageParams(width, height, stage, log_n, twiddle_base)


def dispatch_dims(params):
    """ceil(width / 8) x ceil((height / 2) / 8) x 1 workgroups of 8 x 8 (backend_vulkan.rs:818-839)"""
    half = max(params.height // 2, 1)
    return ((params.width + 7) // 8, (half + 7) // 8, 1)


@dataclass
class ComputePlan:
    params: FftStageParams
    dispatch: tuple
    spv_len: int = 0  # no SPIR-V here: the kernels live in libp3hip.so


def prepare_compute_plan(width, height, stage, log_n):
    params = params_for_stage(width, height, stage, log_n, 1)
    return ComputePlan(params, dispatch_dims(params))


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
