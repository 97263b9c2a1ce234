"""Host-side mirror of the reference's backend selector (native/src/gpu_dft.rs) for the hip backend.

Same names and behaviour: BackendKind, set_backend_kind[_from_str], get_backend_kind,
take_last_error (gpu_dft.rs:14-68) and GpuDft with the TwoAdicSubgroupDft methods
(gpu_dft.rs:70-115; idft/coset methods are Plonky3's provided trait methods).  Matrices are
row-major (height, width) arrays of BabyBear Montgomery words: numpy uint32 on the host (the
"e2e" path: upload, kernels, download) or torch int32/uint32 device tensors (torch device "cuda" is HIP on ROCm; device-resident path,
enqueued on the current torch stream).

Difference from the reference, on purpose: no CPU fallback.  gpu_dft.rs:100-112 swallows backend
errors and reruns Plonky3's Radix2DitParallel; this package ships no CPU prover, so errors raise.
"""
import ctypes as C
import enum

import numpy as np

from . import _lib

P = 0x78000001
MONTY_ONE = 0x0FFFFFFE
GENERATOR_MONTY = (31 << 32) % P  # Val::GENERATOR in Montgomery form


class BackendKind(enum.IntEnum):  # gpu_dft.rs:14-40 + Hip
    Cpu = 0
    Vulkan = 1
    Metal = 2
    WebGpu = 3
    Hip = 4


def set_backend_kind_from_str(value):
    """gpu_dft.rs:53-63.  Raises ValueError("unknown backend '<x>'") for unknown names."""
    rc = _lib.lib().p3hip_set_backend(str(value).encode())
    if rc != 0:
        raise ValueError(_lib.take_last_error() or "unknown backend")


def set_backend_kind(kind):
    set_backend_kind_from_str(BackendKind(kind).name)


def get_backend_kind():
    return BackendKind(_lib.lib().p3hip_get_backend())


def take_last_error():
    return _lib.take_last_error()


def last_timing_line():
    """The reference's per-call log line (backend_vulkan.rs:1385-1423) for this thread's last host-pointer DFT call."""
    m = _lib.lib().p3hip_last_timing_line()
    return m.decode() if m else None


def is_available():
    """lib.rs:167-179 isVulkanAvailable: (ok, message)."""
    buf = C.create_string_buffer(256)
    rc = _lib.lib().p3hip_is_available(buf, 256)
    if rc != 0:
        _lib.take_last_error()
    return rc == 0, buf.value.decode()


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_u32(a, device="cuda"):
    """numpy uint32 -> torch int32 tensor on the GPU (same bits)."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).to(device)


def host_u32(t):
    """torch int32/uint32 tensor -> numpy uint32 (same bits)."""
    import torch
    if t.dtype != torch.int32:
        t = t.view(torch.int32)
    return t.detach().cpu().numpy().view(np.uint32)


class GpuDft:
    """GpuDft<BabyBear> (gpu_dft.rs:70-115) bound to the hip backend."""

    def __init__(self, backend=None):
        self.backend = get_backend_kind() if backend is None else BackendKind(backend)  # Default: gpu_dft.rs:76-83

    @classmethod
    def with_backend(cls, backend):  # gpu_dft.rs:86-92
        return cls(backend)

    def _require_hip(self):
        if self.backend != BackendKind.Hip:
            raise _lib.P3HipError(-3, "backend %s is not provided by this package: only 'hip' runs here "
                                      "(the CPU path is Plonky3's Radix2DitParallel on the Rust side)"
                                  % self.backend.name)

    def _run(self, host_fn, dev_fn, mat, out_rows, *extra):
        self._require_hip()
        L = _lib.lib()
        if _is_torch(mat):
            import torch
            if not mat.is_cuda or mat.dim() != 2 or mat.element_size() != 4:
                raise ValueError("expected a 2-D device-resident tensor (torch device 'cuda' = HIP) of 32-bit words")
            mat = mat.contiguous()
            h, w = mat.shape
            out = torch.empty((out_rows, w), dtype=mat.dtype, device=mat.device)
            _lib.check(getattr(L, dev_fn)(C.c_void_p(mat.data_ptr()), C.c_void_p(out.data_ptr()), h, w, *extra,
                                          _stream_ptr()))
            return out
        mat = np.ascontiguousarray(mat, dtype=np.uint32)
        if mat.ndim != 2:
            raise ValueError("expected a 2-D matrix")
        h, w = mat.shape
        out = np.zeros((out_rows, w), dtype=np.uint32)
        _lib.check(getattr(L, host_fn)(mat.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), h, w, *extra))
        return out

    # TwoAdicSubgroupDft ---------------------------------------------------------------------
    def dft_batch(self, mat):
        return self._run("p3hip_dft_batch_bb31", "p3hip_dft_batch_bb31_dev", mat, mat.shape[0])

    def idft_batch(self, mat):
        return self._run("p3hip_idft_batch_bb31", "p3hip_idft_batch_bb31_dev", mat, mat.shape[0])

    def coset_dft_batch(self, mat, shift_monty):
        return self._run("p3hip_coset_dft_batch_bb31", "p3hip_coset_dft_batch_bb31_dev", mat, mat.shape[0],
                         C.c_uint32(int(shift_monty)))

    def coset_lde_batch(self, mat, added_bits, shift_monty, bit_reversed_out=False):
        return self._run("p3hip_coset_lde_batch_bb31", "p3hip_coset_lde_batch_bb31_dev", mat,
                         mat.shape[0] << added_bits, C.c_uint(added_bits), C.c_uint32(int(shift_monty)),
                         C.c_int(int(bool(bit_reversed_out))))


def coset_lde_from_coeffs(coeffs, added_bits, shift_monty):
    """Device tensor of COEFFICIENTS (h x w, natural order) -> evaluations over shift*<g_{h << added_bits}>, bit-reversed rows
    (= coset_dft_batch of the zero-padded matrix + bit_reverse_rows; p3hip_coset_lde_from_coeffs_bb31_dev)."""
    import torch
    h, w = coeffs.shape
    out = torch.empty((h << added_bits, w), dtype=torch.int32, device=coeffs.device)
    _lib.check(_lib.lib().p3hip_coset_lde_from_coeffs_bb31_dev(C.c_void_p(coeffs.data_ptr()), C.c_void_p(out.data_ptr()), h, w,
                                                               C.c_uint(added_bits), C.c_uint32(int(shift_monty)), _stream_ptr()))
    return out


def bit_reverse_rows(t):
    import torch
    out = torch.empty_like(t)
    h, w = t.shape
    _lib.check(_lib.lib().p3hip_bit_reverse_rows_dev(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), h, w,
                                                     _stream_ptr()))
    return out
