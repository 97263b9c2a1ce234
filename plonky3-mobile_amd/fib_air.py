"""Host-side mirror of the reference's workload harness (native/src/fib_air.rs)."""
import ctypes as C

from . import _lib
from .gpu_dft import _stream_ptr

NUM_FIBONACCI_COLS = 2  # fib_air.rs:25


def generate_trace_rows(a, b, n, device="cuda"):
    """fib_air.rs:266-284 on the device: returns an (n, 2) int32 device tensor (torch device "cuda" = HIP on ROCm) of Montgomery words."""
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


def _hash_kind(hash):
    kinds = {"poseidon2": 0, "keccak": 1}
    if hash not in kinds:
        raise ValueError("unknown hash configuration %r" % (hash,))
    return kinds[hash]


PROFILE_THROUGHPUT, PROFILE_LATENCY = 1, 2  # include/p3hip.h P3HIP_PROFILE_*


def profile_kind(profile):
    try:
        return {"throughput": PROFILE_THROUGHPUT, "latency": PROFILE_LATENCY, PROFILE_THROUGHPUT: PROFILE_THROUGHPUT,
                PROFILE_LATENCY: PROFILE_LATENCY}[profile]
    except KeyError:
        raise ValueError("unknown profile %r (throughput | latency)" % (profile,)) from None


def set_thread_profile(profile):
    """Profile of the free functions (MMCS commits) called on THIS thread from now on; objects carry their own."""
    _lib.check(_lib.lib().p3hip_set_thread_profile(profile_kind(profile)))


def get_thread_profile():
    return {PROFILE_THROUGHPUT: "throughput", PROFILE_LATENCY: "latency"}.get(_lib.lib().p3hip_get_thread_profile())


class FibAirProver:
    """prove(&config, &FibonacciAir{}, generate_trace_rows(a, b, 2^log_n), &[a, b, x]) (fib_air.rs:61-70)
    on the hip backend.  One instance = one HBM arena + one stream; use one per host thread."""

    def __init__(self, log_n, log_blowup=1, params=None, own_stream=True, hash="poseidon2", hiding=False, seed=1, profile="latency"):
        """hash="keccak": the reference's own hash configuration (fib_air.rs:28-53: Keccak MMCS +
        SerializingChallenger32 over a Keccak-256 HashChallenger).  hiding=True: the reference's MerkleTreeHidingMmcs +
        HidingFriPcs with both SmallRng streams seeded by `seed` (fib_air.rs:40-65; wire format version 2).
        profile (include/p3hip.h PROFILES), fixed at creation like the reference's backend (gpu_dft.rs:85-92): "latency" = one proof
        at a time, what the reference does (the default, as p3hip_fib_prover_create*); "throughput" = this prover shares the chip
        with others (bench.py's prover threads).  The proof bytes are the same."""
        self.params = params or FriParameters(log_blowup=log_blowup)
        self.log_n = log_n
        self.hash = hash
        self.hiding = hiding
        self.profile = profile
        self._h = C.c_void_p()
        stream = None if own_stream else _stream_ptr()
        _lib.check(_lib.lib().p3hip_fib_prover_create_profile(profile_kind(profile), _hash_kind(hash), 1 if hiding else 0, seed if hiding else 0,
                                                              log_n, C.cast(self.params._c(), C.c_void_p), stream, 1 if own_stream else 0,
                                                              C.byref(self._h)))

    def prove(self, a, b):
        """Returns the proof bytes (wire format: DESIGN.md)."""
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        _lib.check(_lib.lib().p3hip_fib_prover_prove(self._h, a, b, C.byref(out), C.byref(n)))
        return C.string_at(out, n.value)

    def prove_into(self, a, b, address, cap):
        """Proves and writes the bytes at `address` (an int: e.g. the data pointer of a pinned staging row of `cap` bytes);
        returns the proof's length.  No Python bytes object is made: the C call runs without the GIL."""
        n = C.c_size_t()
        _lib.check(_lib.lib().p3hip_fib_prover_prove_into(self._h, a, b, C.c_void_p(address), cap, C.byref(n)))
        return n.value

    def enqueue(self, a, b):
        """Queues the proof's launches and returns at once (at most two proofs in flight; not for the hiding prover)."""
        _lib.check(_lib.lib().p3hip_fib_prover_enqueue(self._h, a, b))

    def finish(self):
        """Waits for the oldest enqueued proof and returns its bytes."""
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        _lib.check(_lib.lib().p3hip_fib_prover_finish(self._h, C.byref(out), C.byref(n)))
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

    def grind_miss_probe(self):
        """(misses, indices): how many proofs needed the proof-of-work continuation, and the device's query-index buffer
        as it stood right after the last miss (numpy uint32; empty before the first miss)."""
        import numpy as np
        misses, n = C.c_uint64(), C.c_size_t()
        buf = (C.c_uint32 * 4096)()
        _lib.check(_lib.lib().p3hip_fib_prover_grind_miss_probe(self._h, C.byref(misses), buf, 4096, C.byref(n)))
        return int(misses.value), np.frombuffer(buf, dtype=np.uint32, count=min(n.value, 4096)).copy()

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


# ---- the reference's DFT benchmark harness (native/src/fib_air.rs:77-222) -----------------------------
_P = 0x78000001
BENCHMARK_CASES = [(256, 8), (1024, 8), (4096, 8), (16384, 8), (4096, 32), (16384, 32), (4096, 64), (4096, 128),
                   (16384, 64), (16384, 128), (256, 16000)]  # fib_air.rs:103-117


def benchmark_input(height, width):
    """fib_air.rs:77-86: v_i = (17 i + 3) mod P as BabyBear (Montgomery words), numpy uint32 (height, width)."""
    import numpy as np
    i = np.arange(height * width, dtype=np.uint64)
    v = (i * np.uint64(17) + np.uint64(3)) % np.uint64(_P)
    return ((v << np.uint64(32)) % np.uint64(_P)).astype(np.uint32).reshape(height, width)


def percentile_ms(samples, q):
    """fib_air.rs:88-96: nearest-rank percentile."""
    import math
    if not samples:
        return 0.0
    s = sorted(samples)
    idx = min(max(int(math.ceil(q * len(s))) - 1, 0), len(s) - 1)
    return s[idx]


def run_dft_benchmark(cases=None, warmup=1, repeats=10, e2e_batch_size=4, cpu_dft=None):
    """run_dft_benchmark (fib_air.rs:98-222) for the hip backend: per shape avg/median/p95 of
      hip_e2e (host matrix in, host matrix out: upload + kernels + download + sync),
      hip_e2e_batched (e2e_batch_size DFTs per synchronisation, per-DFT time),
      hip_kernel (device resident, HIP events),
    and — when `cpu_dft` (a callable matrix -> matrix, e.g. the test oracle) is given — the CPU column, the
    speedups and the reference's equality check (fib_air.rs:193-196: mismatch is a hard error).
    Returns (text report, list of per-shape dicts)."""
    import time

    import numpy as np
    import torch

    from .gpu_dft import BackendKind, GpuDft, dev_u32, host_u32, take_last_error
    dft = GpuDft.with_backend(BackendKind.Hip)
    lines = ["dft benchmark (repeats=%d, warmup=%d, stats=avg/median/p95)" % (repeats, warmup)]
    rows = []

    def stats(v):
        return sum(v) / len(v), percentile_ms(v, 0.50), percentile_ms(v, 0.95)

    for h, w in (cases or BENCHMARK_CASES):
        x = benchmark_input(h, w)
        xd = dev_u32(x)
        take_last_error()
        for _ in range(warmup):
            dft.dft_batch(x)
            dft.dft_batch(xd)
        e2e, out = [], None
        for _ in range(repeats):
            t = time.perf_counter()
            out = dft.dft_batch(x)
            e2e.append((time.perf_counter() - t) * 1e3)
        batched = []
        pinned_in = torch.from_numpy(x.view(np.int32)).pin_memory()
        outs = [torch.empty((h, w), dtype=torch.int32).pin_memory() for _ in range(e2e_batch_size)]
        for _ in range(repeats):
            t = time.perf_counter()
            for k in range(e2e_batch_size):
                d = pinned_in.to("cuda", non_blocking=True)
                outs[k].copy_(dft.dft_batch(d), non_blocking=True)
            torch.cuda.synchronize()
            batched.append((time.perf_counter() - t) * 1e3 / e2e_batch_size)
        kern = []
        for _ in range(repeats):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            yd = dft.dft_batch(xd)
            e.record()
            torch.cuda.synchronize()
            kern.append(s.elapsed_time(e))
        if take_last_error() is not None:
            raise RuntimeError("hip benchmark error at h=%d, w=%d" % (h, w))
        row = {"h": h, "w": w, "hip_e2e": stats(e2e), "hip_e2e_batched": stats(batched), "hip_kernel": stats(kern)}
        line = "h=%d, w=%d:" % (h, w)
        if cpu_dft is not None:
            cpu, cpu_out = [], None
            for _ in range(max(1, min(repeats, 3))):
                t = time.perf_counter()
                cpu_out = cpu_dft(x)
                cpu.append((time.perf_counter() - t) * 1e3)
            if not (np.array_equal(cpu_out, out) and np.array_equal(cpu_out, host_u32(yd))):
                raise RuntimeError("dft benchmark mismatch at h=%d, w=%d" % (h, w))
            row["cpu"] = stats(cpu)
            line += " cpu(avg=%.3f med=%.3f p95=%.3f)ms" % row["cpu"]
        for key in ("hip_e2e", "hip_e2e_batched", "hip_kernel"):
            line += " %s(avg=%.3f med=%.3f p95=%.3f)ms" % ((key,) + row[key])
            if cpu_dft is not None:
                line += " speedup_%s(avg)=%.2fx" % (key[4:], row["cpu"][0] / row[key][0] if row[key][0] > 0 else 0.0)
        lines.append(line)
        rows.append(row)
    return "\n".join(lines), rows


def fib_public_x(a, b, n):
    """Last row's right value (the public value x, fib_air.rs:57,68), canonical integer."""
    l, r = a % _P, b % _P
    for _ in range(n - 1):
        l, r = r, (l + r) % _P
    return r


def verify_fib_air(proof, a, b, x, log_n, params=None, hash="poseidon2", hiding=False):
    """verify(&config, &FibonacciAir{}, &proof, &[a, b, x]) (fib_air.rs:71-72), host side.  Raises
    P3HipError("fib_air verification failed: <check>") on rejection, like the reference's map_err."""
    params = params or FriParameters()
    buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
    fn = _lib.lib().p3hip_verify_fib_air_hiding if hiding else _lib.lib().p3hip_verify_fib_air_hash
    _lib.check(fn(_hash_kind(hash), buf, len(proof), a, b, x, log_n, C.cast(params._c(), C.c_void_p)))


def run_fib_air(log_n=3, a=0, b=1, params=None, hash="poseidon2", hiding=False, seed=1):
    """run_fib_air_zk (fib_air.rs:27-75) on the hip backend: prove, verify, report.  Default n = 8, x = 21 as in the
    reference (fib_air.rs:56-57).  hash="keccak", hiding=True is the reference's own configuration: Keccak hashes,
    MerkleTreeHidingMmcs and HidingFriPcs with SmallRng::seed_from_u64(1) (fib_air.rs:28-65)."""
    params = params or FriParameters()
    n = 1 << log_n
    x = fib_public_x(a, b, n)
    prover = FibAirProver(log_n, params=params, hash=hash, hiding=hiding, seed=seed)
    try:
        proof = prover.prove(a, b)
    finally:
        prover.close()
    verify_fib_air(proof, a, b, x, log_n, params, hash=hash, hiding=hiding)
    return "fib_air %sok (n=%d, x=%d)" % ("zk " if hiding else "", n, x)


def run_fib_air_zk_report():
    """p3hip_run_fib_air_zk: the String the reference's runFibAirZk() hands to Java (lib.rs:37-83), produced by the library
    itself — the reference's own instance and configuration on whatever backend the selector names."""
    buf = C.create_string_buffer(1024)
    _lib.lib().p3hip_run_fib_air_zk(buf, len(buf))
    return buf.value.decode()


CPU_DFT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t)


def run_dft_benchmark_report(cpu_dft=None):
    """p3hip_run_dft_benchmark: the String of runDftBenchmark() (lib.rs:86-131) from the library; `cpu_dft` (matrix -> matrix
    on numpy uint32, e.g. the test oracle) supplies the CPU column and the equality check through the C callback the Rust shim
    would fill with Radix2DitParallel."""
    import numpy as np
    cb = None
    if cpu_dft is not None:
        def _cb(_user, pin, pout, h, w):
            try:
                x = np.ctypeslib.as_array(pin, shape=(h, w)).copy()
                np.ctypeslib.as_array(pout, shape=(h, w))[:] = cpu_dft(x)
                return 0
            except Exception:  # never raise through the C boundary
                return 1
        cb = CPU_DFT_FN(_cb)
    buf = C.create_string_buffer(1 << 14)
    _lib.lib().p3hip_run_dft_benchmark(C.cast(cb, C.c_void_p) if cb else None, None, buf, len(buf))
    return buf.value.decode()


class DeviceRng:
    """rand 0.9.2 `SmallRng::seed_from_u64(seed)` as a device-resident stream of BabyBear elements (rng.hip)."""

    def __init__(self, seed=1):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().p3hip_rng_create(seed, C.byref(self._h)))

    def fill_field(self, n):
        """The next n elements of the stream as an int32 device tensor (torch device "cuda" = HIP on ROCm) of Montgomery words (enqueued on the current stream)."""
        import torch
        out = torch.empty((max(n, 1),), dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib().p3hip_rng_fill_field_dev(self._h, C.c_void_p(out.data_ptr()), n, _stream_ptr()))
        return out[:n]

    def state(self):
        s = (C.c_uint64 * 4)()
        _lib.check(_lib.lib().p3hip_rng_state(self._h, s, _stream_ptr()))
        return list(s)

    def close(self):
        if self._h:
            _lib.lib().p3hip_rng_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FibAirBatchProver:
    """A pool of provers inside libp3hip (one host thread + stream + HBM arena each) proving batches of
    independent instances — BASELINE configs[3] on one GPU; across GPUs see batch.py."""

    def __init__(self, log_n, n_provers=8, params=None, hash="poseidon2", hiding=False, seed=1):
        self.params = params or FriParameters()
        self.hash = hash
        self._h = C.c_void_p()
        if hiding:
            _lib.check(_lib.lib().p3hip_fib_batch_create_hiding(_hash_kind(hash), log_n, C.cast(self.params._c(), C.c_void_p),
                                                                seed, n_provers, C.byref(self._h)))
        else:
            _lib.check(_lib.lib().p3hip_fib_batch_create_hash(_hash_kind(hash), log_n, C.cast(self.params._c(), C.c_void_p),
                                                              n_provers, C.byref(self._h)))

    def prove(self, instances):
        """instances: list of (a, b).  Returns the list of proof bytes in the same order."""
        n = len(instances)
        a = (C.c_uint64 * n)(*[i[0] for i in instances])
        b = (C.c_uint64 * n)(*[i[1] for i in instances])
        ptrs = (C.POINTER(C.c_uint8) * n)()
        lens = (C.c_size_t * n)()
        _lib.check(_lib.lib().p3hip_fib_batch_prove(self._h, n, a, b, ptrs, lens))
        return [C.string_at(ptrs[i], lens[i]) for i in range(n)]

    def submit(self, instances):
        """Queues a batch behind the ones already submitted and returns a ticket at once (at most 8 in flight)."""
        n = len(instances)
        a = (C.c_uint64 * n)(*[i[0] for i in instances])
        b = (C.c_uint64 * n)(*[i[1] for i in instances])
        ticket = C.c_uint64()
        _lib.check(_lib.lib().p3hip_fib_batch_submit(self._h, n, a, b, C.byref(ticket)))
        return (ticket.value, n)

    def collect(self, ticket):
        """Waits for a submitted batch; returns its proofs in instance order."""
        t, n = ticket
        ptrs = (C.POINTER(C.c_uint8) * max(n, 1))()
        lens = (C.c_size_t * max(n, 1))()
        _lib.check(_lib.lib().p3hip_fib_batch_collect(self._h, t, ptrs, lens))
        return [C.string_at(ptrs[i], lens[i]) for i in range(n)]

    def close(self):
        if self._h:
            _lib.lib().p3hip_fib_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
