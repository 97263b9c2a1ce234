"""Benchmark job for bench.py: a batch of independent fib_air instances resident in HBM.
Product-side only (no test oracle here); the CPU baseline leg lives in bench.py."""
import ctypes as C

import torch

from . import _lib
from .gpu_dft import GENERATOR_MONTY, BackendKind, GpuDft, _stream_ptr
from .fib_air import generate_trace_rows
from .mmcs import MerkleTreeMmcs


class FibAirJob:
    def __init__(self, p3, log_height, log_blowup, batch, first_instance=0):
        self.p3 = p3
        self.log_height, self.log_blowup, self.batch = log_height, log_blowup, batch
        self.first = first_instance
        self.n = 1 << log_height
        self.dft = GpuDft.with_backend(BackendKind.Hip)
        self.mmcs = MerkleTreeMmcs()
        self.prover = getattr(p3, "FibAirProver", None)
        if self.prover is not None:
            self.prover = self.prover(log_height, log_blowup)
        self.last = None

    def metric_name(self):
        if self.prover is not None:
            return "fib_air proofs/sec"
        return "fib_air trace commitments/sec (coset LDE + Poseidon2 MMCS only; full prover not built yet)"

    def unit(self):
        return "proofs/s" if self.prover is not None else "commitments/s"

    def workload_name(self):
        return "fib_air 2^%d-row trace, BabyBear+Poseidon2, blowup %d (BASELINE configs[1])" % (
            self.log_height, 1 << self.log_blowup)

    def step(self):
        outs = []
        for i in range(self.batch):
            a = self.first + i
            if self.prover is not None:
                outs.append(self.prover.prove(a, a + 1))
            else:
                trace = generate_trace_rows(a, a + 1, self.n)
                lde = self.dft.coset_lde_batch(trace, self.log_blowup, GENERATOR_MONTY, bit_reversed_out=True)
                root, tree = self.mmcs.commit([lde])
                outs.append(root)
                tree.free()
        self.last = outs
        return outs

    def _time(self, fn, reps):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for s, e in evs:
            s.record()
            fn()
            e.record()
        torch.cuda.synchronize()
        return sum(s.elapsed_time(e) for s, e in evs) / reps  # ms

    def lde_roofline(self, reps=20):
        """coset LDE of one 2^h x 2 trace: algorithmic bytes = 4*h*w*(1+blowup)."""
        trace = generate_trace_rows(0, 1, self.n)
        out = torch.empty((self.n << self.log_blowup, 2), dtype=torch.int32, device="cuda")
        L = _lib.lib()

        def run():
            _lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(trace.data_ptr()), C.c_void_p(out.data_ptr()),
                                                        self.n, 2, self.log_blowup, GENERATOR_MONTY, 1, _stream_ptr()))
        ms = self._time(run, reps)
        nbytes = 4 * self.n * 2 * (1 + (1 << self.log_blowup))
        res = {"bytes": nbytes, "avg_us": ms * 1e3, "gbps": nbytes / (ms * 1e-3) / 1e9}
        # same unit over a wide batch (64 traces side by side = 2^h x 128): out of L2/Infinity-cache regime
        wide = torch.randint(0, 0x78000001, (self.n, 128), dtype=torch.int32, device="cuda")
        wout = torch.empty((self.n << self.log_blowup, 128), dtype=torch.int32, device="cuda")

        def run_wide():
            _lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(wide.data_ptr()), C.c_void_p(wout.data_ptr()),
                                                        self.n, 128, self.log_blowup, GENERATOR_MONTY, 1, _stream_ptr()))
        wms = self._time(run_wide, max(3, reps // 4))
        res["batched_gbps"] = 64 * nbytes / (wms * 1e-3) / 1e9
        return res

    def stage_breakdown(self):
        trace = generate_trace_rows(0, 1, self.n)
        t_trace = self._time(lambda: generate_trace_rows(0, 1, self.n), 5)
        lde = self.dft.coset_lde_batch(trace, self.log_blowup, GENERATOR_MONTY, bit_reversed_out=True)
        t_lde = self._time(lambda: self.dft.coset_lde_batch(trace, self.log_blowup, GENERATOR_MONTY, True), 5)

        def commit():
            _, t = self.mmcs.commit([lde])
            t.free()
        t_commit = self._time(commit, 5)
        out = {"trace_gen": t_trace, "trace_lde": t_lde, "trace_commit": t_commit}
        if self.prover is not None and hasattr(self.prover, "stage_breakdown"):
            out.update(self.prover.stage_breakdown())
        return out
