"""Benchmark job for bench.py: a batch of independent fib_air instances proved on one GPU.
Product-side only (no test oracle here); the CPU baseline leg lives in bench.py.

Independent proofs overlap on the device: each of `threads` host threads owns one prover (HBM arena +
non-blocking stream + per-thread context) and proves its share of the step's instances; the transcript
round trips of one proof hide behind the kernels of the others."""
import ctypes as C
import os
import queue
import time
import threading

import torch

from . import _lib
from .fib_air import FibAirProver, FriParameters, generate_trace_rows
from .gpu_dft import GENERATOR_MONTY, BackendKind, GpuDft, _stream_ptr
from .mmcs import MerkleTreeMmcs


def _profile_json(names):
    """First readable profiles/<name> of `names` -> (name, parsed, stale): stale = the file records the SHA-256 of the libp3hip.so
    it was measured on and the library loaded now is another build (round-4 advisor finding: a checked-in instruction count
    divided by a time measured in this run was presented as a measurement of this run whatever the code had become)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in names:
        try:
            with open(os.path.join(root, "profiles", name)) as f:
                pmc = json.load(f)
        except Exception:
            continue
        # the SOURCES' hash where the profile has one (it survives a rebuild of the same sources elsewhere), else the binary's
        if pmc.get("src_sha256") is not None:
            return name, pmc, pmc["src_sha256"] != _lib.src_sha256()
        sha = pmc.get("lib_sha256")
        return name, pmc, (None if sha is None else sha != _lib.lib_sha256())
    return None, None, None


def _stale_note(stale):
    return {None: "the profile predates build identities (round 4 or earlier): may describe older kernels",
            True: "STALE: the profile was measured on other sources / another build of libp3hip.so than this tree's",
            False: "same build: the profile's source / library hash equals this tree's"}[stale]


def lde_valu_view(key, unit_us, algorithmic_bytes):
    """The coset-LDE unit against the roofline that actually binds it from ~2^21 rows on: VALU issue.  `key` = cfg2 | cfg3 | cfg5 in
    profiles/r05_pmc_lde_valu.json (SQ_INSTS_VALU of the unit's launches, tools/r05_lde_valu.sh; round 4's file as a fallback).
    valu_frac = wave-instructions of the unit / measured unit time / the chip's measured issue rate (36 T lane-ops/s = 562.5 G
    wave-instructions/s, profiles/r01_microbench2_valu_issue_rates.txt); hbm_frac_ceiling_at_this_instruction_count = the best HBM
    fraction this many instructions allow, i.e. algorithmic bytes / (instructions / issue rate) / 8 TB/s.  `valu_profile_build` says
    whether the counts come from the library that is loaded now."""
    try:
        name, pmc, stale = _profile_json(("r05_pmc_lde_valu.json", "r04_pmc_lde_valu.json"))
        e = pmc[key]
        peak = pmc.get("peak_wave_instr_per_s", 36e12 / 64)
        wi = e["unit_valu_wave_instr"]
        return {"valu_frac": wi / (unit_us * 1e-6) / peak,
                "valu_wave_instructions_per_unit": wi,
                "valu_instructions_per_butterfly": e["valu_lane_instr_per_butterfly"],
                "valu_peak": "36 T lane-ops/s measured (profiles/r01_microbench2_valu_issue_rates.txt)",
                "hbm_frac_ceiling_at_this_instruction_count": algorithmic_bytes / (wi / peak) / 8e12,
                "valu_source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU over the unit's launches)" % name,
                "valu_profile_build": _stale_note(stale)}
    except Exception:
        return {"valu_frac": None, "valu_source": None}


def proof_valu_view(key):
    """VALU wave-instructions of ONE WHOLE proof of workload `key` (cfg2 | cfg2_keccak | cfg2_keccak_hiding | cfg3), counted by PMC over
    every kernel the prover launches (profiles/r05_pmc_proofs.json, tools/r05_pmc_proofs.sh).  bench.py multiplies by the measured
    proofs/s: the proof-level VALU issue fraction is the ruler that binds these workloads."""
    name, pmc, stale = _profile_json(("r05_pmc_proofs.json",))
    if pmc is None or key not in pmc.get("workloads", {}):
        return None
    w = pmc["workloads"][key]
    return {"valu_wave_instr_per_proof": w["valu_wave_instr_per_proof"], "hash_kernels_share": w["hash_kernels_share"],
            "peak_wave_instr_per_s": pmc["peak_wave_instr_per_s"], "top_kernels": [(k["kernel"], k["share"]) for k in w["kernels"][:4]],
            "source": "profiles/%s, workload %s: rocprofv3 --pmc SQ_INSTS_VALU over every launch of one prover's proofs" % (name, key),
            "profile_build": _stale_note(stale), "stale": stale}


class _Worker(threading.Thread):
    def __init__(self, device, log_height, params, stagger_s=0.0, hash="poseidon2", hiding=False, profile="throughput"):
        super().__init__(daemon=True)
        self.hash, self.hiding, self.profile = hash, hiding, profile
        self.ahead = not hiding and os.environ.get("P3HIP_BENCH_AHEAD", "0") == "1"
        self.stagger_s = stagger_s
        self.device, self.log_height, self.params = device, log_height, params
        self.inbox, self.outbox = queue.Queue(), queue.Queue()
        self.prover = None
        self.start()

    def run(self):
        try:
            torch.cuda.set_device(self.device)
            from .fib_air import set_thread_profile
            set_thread_profile(self.profile)  # of this thread's free MMCS calls; the prover carries its own
            self.prover = FibAirProver(self.log_height, params=self.params, hash=self.hash, hiding=self.hiding, profile=self.profile)
            self.outbox.put(("ready", None))
        except Exception as e:  # surfaced by the caller
            self.outbox.put(("error", e))
            return
        while True:
            job = self.inbox.get()
            if job is None:
                return
            kind, arg = job
            try:
                if kind == "prove":
                    # work-stealing over the step's shared queue of (slot, a); the first step is staggered so the
                    # provers do not run their small-kernel phases (tree tops, FRI tail) in lockstep
                    if self.stagger_s:
                        time.sleep(self.stagger_s)
                        self.stagger_s = 0.0
                    jobs, sink = arg
                    done = []

                    def emit(i, pf):
                        if sink is not None:
                            sink(i, pf)  # e.g. straight into the pinned staging row of the step's gather
                        done.append((i, pf))
                    ahead = None  # P3HIP_BENCH_AHEAD=1: a second proof enqueued behind the one being collected (helps one prover: +4 %; nothing at four)
                    while True:
                        try:
                            i, a = jobs.get_nowait()
                        except queue.Empty:
                            break
                        if getattr(sink, "direct", False):
                            # the proof goes straight into the staging row of the step's gather: no bytes object on the way
                            address, cap = sink.buffer(i)
                            sink.done(i, self.prover.prove_into(a, a + 1, address, cap))
                            done.append((i, None))
                            continue
                        if not self.ahead:
                            emit(i, self.prover.prove(a, a + 1))
                            continue
                        self.prover.enqueue(a, a + 1)
                        if ahead is not None:
                            emit(ahead, self.prover.finish())
                        ahead = i
                    if ahead is not None:
                        emit(ahead, self.prover.finish())
                    self.outbox.put(("ok", done))
                elif kind == "stages":
                    self.outbox.put(("ok", self.prover.stage_breakdown()))
            except Exception as e:
                self.outbox.put(("error", e))

    def result(self):
        status, val = self.outbox.get()
        if status == "error":
            raise val
        return val


class FibAirJob:
    def __init__(self, p3, log_height, log_blowup, batch, first_instance=0, threads=4, hash="poseidon2", config_label=None, hiding=False):
        self.p3 = p3
        self.hash, self.hiding = hash, hiding
        self.config_label = config_label  # "configs[1]" / "configs[2]" when the size IS that BASELINE config
        self.device = torch.cuda.current_device()
        self.log_height, self.log_blowup, self.batch = log_height, log_blowup, batch
        self.first = first_instance
        self.n = 1 << log_height
        self.dft = GpuDft.with_backend(BackendKind.Hip)
        self.mmcs = MerkleTreeMmcs()
        self.params = FriParameters(log_blowup=log_blowup)
        self.threads = max(1, min(threads, batch))
        stag = float(os.environ.get("P3HIP_BENCH_STAGGER_MS", "1.0")) * 1e-3
        # provers that share the chip run the THROUGHPUT profile; a job of one prover is a lone prover: LATENCY (include/p3hip.h)
        self.profile = "throughput" if self.threads > 1 else "latency"
        self.workers = [_Worker(self.device, log_height, self.params, stag * t, hash, hiding, self.profile) for t in range(self.threads)]
        for w in self.workers:
            w.result()
        self.last = None
        self._begun = []

    def close(self):
        for w in self.workers:
            w.inbox.put(None)

    def metric_name(self):
        return "fib_air proofs/sec"

    def unit(self):
        return "proofs/s"

    def workload_name(self):
        if self.hiding:
            return ("fib_air 2^%d-row trace, the reference's own configuration%s: hiding MMCS (4 salt elements) + HidingFriPcs (4 random "
                    "codewords), SmallRng streams seeded with 1 (fib_air.rs:28-65), blowup %d" % (
                        self.log_height, " with its Keccak hashes" if self.hash == "keccak" else " but Poseidon2 hashes", 1 << self.log_blowup))
        if self.hash == "keccak":
            return ("fib_air 2^%d-row trace, BabyBear + the reference's own Keccak hashes (fib_air.rs:28-53, non-hiding), "
                    "blowup %d" % (self.log_height, 1 << self.log_blowup))
        return "fib_air 2^%d-row trace, BabyBear+Poseidon2, blowup %d%s" % (
            self.log_height, 1 << self.log_blowup, " (BASELINE %s)" % self.config_label if self.config_label else "")

    def config(self):
        return {"log_height": self.log_height, "width": 2, "log_blowup": self.log_blowup, "batch_per_gpu": self.batch,
                "concurrent_provers_per_gpu": self.threads, "hiding": self.hiding, "profile": self.profile,
                "fri": {"log_final_poly_len": self.params.log_final_poly_len, "num_queries": self.params.num_queries,
                        "proof_of_work_bits": self.params.proof_of_work_bits}}

    def roofline(self):
        """The dominant HBM-bound unit of a proof: the coset LDE (HIP events on the launch stream; algorithmic bytes
        4*h*w*(1+blowup)); `traffic` from the committed PMC passes of the same unit."""
        import json
        roof = self.lde_roofline(reps=20)
        traffic, src = None, None
        name, pmc, stale = _profile_json(("r05_pmc_lde.json", "r04_pmc_lde.json", "r03_pmc_lde.json", "r02_pmc_lde.json", "r01_pmc_lde_v2.json"))
        key = {(20, 1): "cfg2_lde_2^20x2_blowup2", (24, 2): "cfg3_lde_2^24x2_blowup4"}.get((self.log_height, self.log_blowup))
        if pmc is not None and key and key in pmc:
            traffic = pmc[key]["total_bytes"]
            src = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, summed over "
                   "the unit's launches (tools/pmc_probe.py, tools/pmc_summarize.py); %s" % (name, _stale_note(stale)))
        out = {"bound": "hbm", "achieved": roof["gbps"], "peak": 8000.0, "unit": "GB/s", "frac": roof["gbps"] / 8000.0,
               "frac_of_achievable_6300": roof["gbps"] / 6300.0, "traffic": traffic, "traffic_source": src,
               "kernel": "coset_lde_batch (narrow plan: one unit of " + str(roof.get("launches", "3")) + " launches)",
               "algorithmic_bytes": roof["bytes"], "avg_us": roof["avg_us"],
               "concurrent_gbps": roof.get("concurrent_gbps"), "concurrent_streams": roof.get("concurrent_streams")}
        key = {(20, 1): "cfg2", (24, 2): "cfg3"}.get((self.log_height, self.log_blowup))
        if key:
            out.update(lde_valu_view(key, roof["avg_us"], roof["bytes"]))
        return out

    def extra_report(self):
        out = {"valu_roofline": self.hash_roofline()}
        if not self.hiding:  # the hiding prover keeps no per-stage events
            out["stages_ms"] = self.stage_breakdown()
        return out

    # ---- the hash layers: what a proof spends its time in, and the VALU evidence of the configuration that ran ----
    def tree_shapes(self):
        """(rows, words per leaf row) of every Merkle tree one proof commits, in order."""
        big = self.n << self.log_blowup
        if self.hiding:  # prover_hiding.hip.inc: randomized trace on 2h rows; leaf rows carry 4 salt words per matrix
            big <<= 1
            shapes = [(big, 6 + 4), (big, 4 * (4 + 4)), (big, 8 + 4)]
            fri_row = 8 + 4
        else:
            shapes = [(big, 2), (big, 4)]
            fri_row = 8
        log_big = big.bit_length() - 1
        rounds = log_big - self.params.log_blowup - self.params.log_final_poly_len
        shapes += [(big >> (r + 1), fri_row) for r in range(rounds)]
        return shapes

    def permutations_per_proof(self):
        """Leaf sponge + compression permutations of one proof (the grind and transcript permutations, a few thousand, left out).
        Poseidon2: rate 8 words; Keccak: 17 u64 = 34 field elements per permutation (fib_air.rs:31-38)."""
        rate = 34 if self.hash == "keccak" else 8
        return sum(rows * ((w + rate - 1) // rate) + rows - 1 for rows, w in self.tree_shapes())

    def pmc_key(self):
        """This job's workload in profiles/r05_pmc_proofs.json, or None (sizes the PMC pass did not cover)."""
        return {("poseidon2", False, 20, 1): "cfg2", ("keccak", False, 20, 1): "cfg2_keccak", ("keccak", True, 20, 1): "cfg2_keccak_hiding",
                ("poseidon2", False, 24, 2): "cfg3"}.get((self.hash, bool(self.hiding), self.log_height, self.log_blowup))

    def hash_roofline(self):
        """The ruler that binds a proof workload: VALU ISSUE over the whole proof.  A proof is a fixed number of VALU wave-instructions
        (counted by PMC over every kernel one prover launches for it: profiles/r05_pmc_proofs.json, keyed by the build of libp3hip.so);
        bench.py fills in
          achieved = proofs/s of the timed region x that count   [G wave-instructions/s],   frac = achieved / 562.5 (measured issue peak).
        Beside it, as context: which kernels the instructions belong to (the hash layers: 78-97 %), the hash kernels' own rate
        (kernel_ceiling_gperm_s: one 2^24 x 2 commit timed in this run) and what the timed region sustained in permutations
        (sustained_gperm_s, sustained_frac_of_kernel_ceiling: filled in by bench.py), and one commit of this job's trace tree ALONE
        (`lone_tree`: a latency figure — a lone 2^21-leaf tree is seven launches down to 2^15 plus a latency-bound top; round 4 carried its
        VALU-busy, 0.76, as `frac`, which described neither the four-prover run beside it nor the current kernels)."""
        n = self.tree_shapes()[0][0]  # rows of the trace tree (twice as many under hiding: the randomized trace has 2h rows)
        L = _lib.lib()
        kind = 1 if self.hash == "keccak" else 0

        def commit_ms(rows, reps, best=False):
            x = torch.randint(0, 0x78000001, (rows, 2), dtype=torch.int32, device="cuda")
            layers = torch.empty((L.p3hip_mmcs_layer_words(rows),), dtype=torch.int32, device="cuda")
            ptrs, hs, ws = (C.c_void_p * 1)(x.data_ptr()), (C.c_size_t * 1)(rows), (C.c_size_t * 1)(2)

            def commit():
                h = C.c_void_p()
                _lib.check(L.p3hip_mmcs_commit_into_dev(kind, ptrs, hs, ws, 1, C.c_void_p(layers.data_ptr()), C.byref(h), _stream_ptr()))
                L.p3hip_mmcs_free(h)
            if not best:
                return self._time(commit, reps)
            return min(self._time(commit, 1) for _ in range(reps))  # a ceiling: the fastest of `reps` timed commits
        ms = commit_ms(n, 5)
        perms = 2 * n - 1
        big_rows = 1 << 24  # a tree whose layers fill the chip for many workgroup generations: the kernels' own ceiling
        ms_big = commit_ms(big_rows, 4, best=True)
        hname = "Keccak-f[1600]" if kind else "Poseidon2-BabyBear-16"
        out = {"bound": "valu", "unit": "G wave-instr/s", "peak": 36e12 / 64 / 1e9, "achieved": None, "frac": None,
               "peak_source": "36 T lane-ops/s measured (profiles/r01_microbench2_valu_issue_rates.txt)",
               "peak_note": "562.5 G wave-instructions/s is the measured issue rate of the 4-cycle VALU classes (32-bit multiplies, v_mad_u64_u32, fp64, "
                            "v_min / v_add3 / shifts: 512-566 G/s at >= 2 waves per SIMD in that microbenchmark); v_add / v_sub / v_and-class "
                            "instructions issue in ~3 cycles (830-870 G/s), so a proof whose mix is mostly 32-bit logic (Keccak without hiding) can read "
                            "above 1.0 on this ruler; the Poseidon2 (fp64) and hiding workloads cannot",
               "hash": hname, "permutations_per_proof": self.permutations_per_proof(),
               "kernel_ceiling_gperm_s": (2 * big_rows - 1) / (ms_big * 1e-3) / 1e9,
               "bare_permute_kernel_gperm_s": (self.keccak_rate() if kind else self.poseidon2_rate()) / 1e9,
               "lone_tree": {"what": "one commit of 2^%d x 2 into pre-allocated layers, nothing else on the chip (thread profile: %s)" % (
                                 self.log_height + self.log_blowup, "latency"),
                             "gperm_s": perms / (ms * 1e-3) / 1e9, "permutations": perms, "avg_us": ms * 1e3}}
        key = self.pmc_key()
        view = proof_valu_view(key) if key else None
        if view is None:
            out["valu_wave_instr_per_proof"] = None
            out["source"] = "no PMC pass of whole proofs committed for this workload (profiles/r05_pmc_proofs.json covers cfg2, cfg2 under Keccak, Keccak + hiding, cfg3)"
            return out
        out.update({"valu_wave_instr_per_proof": view["valu_wave_instr_per_proof"], "hash_kernels_share": view["hash_kernels_share"],
                    "top_kernels": view["top_kernels"], "source": view["source"], "profile_build": view["profile_build"]})
        return out

    def step_begin(self, instances=None, sink=None):
        """Hands the step's instances to the prover threads and returns at once; step_end() joins them.
        `sink(slot, proof bytes)` is called by the prover thread as soon as a proof is serialised."""
        jobs = queue.Queue()
        todo = instances if instances is not None else [(i, self.first + i) for i in range(self.batch)]
        for item in todo:
            jobs.put(item)
        for w in self.workers:
            w.inbox.put(("prove", (jobs, sink)))
        self._begun.append(instances is not None)  # steps may overlap: step_end() retires them in order

    def step_end(self):
        got = {}
        for w in self.workers:
            for i, pf in w.result():
                got[i] = pf
        if self._begun.pop(0):
            self.last = got
            return got
        res = [got[i] for i in range(self.batch)]
        self.last = res
        return res

    def step(self, instances=None, sink=None):
        """Proves independent instances; default: (a, b) = (first+i, first+i+1) for i < batch.
        `instances`: list of (slot, a) as dealt by batch.scatter_descriptors.  Returns {slot: proof bytes}
        as a list ordered by slot when called with the default."""
        self.step_begin(instances, sink)
        return self.step_end()

    def prove_one(self, a, b):
        w = self.workers[0]
        jobs = queue.Queue()
        jobs.put((0, a))
        w.inbox.put(("prove", (jobs, None)))
        assert b == a + 1
        return w.result()[0][1]

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
        """coset LDE of one 2^h x 2 trace: algorithmic bytes = 4*h*w*(1+blowup).
        HIP events on the stream the kernels are launched on (torch's current stream)."""
        trace = generate_trace_rows(0, 1, self.n)
        out = torch.empty((self.n << self.log_blowup, 2), dtype=torch.int32, device="cuda")
        L = _lib.lib()

        def run():
            _lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(trace.data_ptr()), C.c_void_p(out.data_ptr()),
                                                        self.n, 2, self.log_blowup, GENERATOR_MONTY, 1, _stream_ptr()))
        ms = self._time(run, reps)
        nbytes = 4 * self.n * 2 * (1 + (1 << self.log_blowup))
        res = {"bytes": nbytes, "avg_us": ms * 1e3, "gbps": nbytes / (ms * 1e-3) / 1e9}
        # the same unit the way the proving loop runs it: `threads` independent LDEs in flight at once, one per host
        # thread, each on its own stream and per-thread context (own scratch), timed wall-clock around all of them
        res["concurrent_gbps"], res["concurrent_streams"] = self._concurrent_lde(nbytes), self.threads
        return res

    def _concurrent_lde(self, nbytes, reps=40):
        L = _lib.lib()
        go, done = threading.Barrier(self.threads + 1), threading.Barrier(self.threads + 1)
        errors = []

        def worker():
            try:
                torch.cuda.set_device(self.device)
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    trace = generate_trace_rows(0, 1, self.n)
                    out = torch.empty((self.n << self.log_blowup, 2), dtype=torch.int32, device="cuda")

                    def run():
                        _lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(trace.data_ptr()), C.c_void_p(out.data_ptr()),
                                                                    self.n, 2, self.log_blowup, GENERATOR_MONTY, 1,
                                                                    C.c_void_p(st.cuda_stream)))
                    run()
                    st.synchronize()
                    go.wait()
                    for _ in range(reps):
                        run()
                    st.synchronize()
            except Exception as e:  # pragma: no cover
                errors.append(e)
                go.abort()
            done.wait()

        ths = [threading.Thread(target=worker, daemon=True) for _ in range(self.threads)]
        for t in ths:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        done.wait()
        dt = time.perf_counter() - t0
        for t in ths:
            t.join()
        if errors:
            raise errors[0]
        return self.threads * reps * nbytes / dt / 1e9

    def poseidon2_rate(self, reps=5):
        """Poseidon2 permutations/s of the one-state-per-lane kernel (the dominant kernel of a proof by time),
        HIP events on the launch stream, 2^22 random states in HBM."""
        n = 1 << 22
        st = torch.randint(0, 0x78000001, (n, 16), dtype=torch.int32, device="cuda")
        L = _lib.lib()
        ms = self._time(lambda: _lib.check(L.p3hip_poseidon2_permute_dev(C.c_void_p(st.data_ptr()), n, _stream_ptr())), reps)
        return n / (ms * 1e-3)

    def keccak_rate(self, reps=5):
        """Keccak-f[1600] permutations/s of the one-state-per-lane kernel: 2^22 random states (25 x u64 each) in HBM.  The full
        24 rounds with all 25 lanes loaded and stored: the tree kernels' digest-only last round and 4-word stores are cheaper."""
        n = 1 << 22
        st = torch.randint(-(1 << 62), 1 << 62, (n, 25), dtype=torch.int64, device="cuda")
        L = _lib.lib()
        ms = self._time(lambda: _lib.check(L.p3hip_keccak_f_dev(C.c_void_p(st.data_ptr()), n, _stream_ptr())), reps)
        return n / (ms * 1e-3)

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
        w = self.workers[0]
        w.inbox.put(("stages", None))
        out.update(w.result())  # single-proof latency split, host wall clock
        return out


class WideCommitJob:
    """BASELINE configs[4]: the wide trace 2^16 x 2633 (the reference's benchmark_input, fib_air.rs:77-86, standing in
    for the Keccak-f AIR trace), bit-reversed coset LDE at blowup 2 + MMCS commit of the 2^17-row result.
    step() = one LDE + one commit on the current stream, inputs resident in HBM."""

    def __init__(self, p3, log_height=16, log_blowup=1, width=2633, hash="poseidon2"):
        from .fib_air import benchmark_input
        from .gpu_dft import dev_u32
        self.p3, self.hash = p3, hash
        self.log_height, self.log_blowup, self.width = log_height, log_blowup, width
        self.h = 1 << log_height
        self.dft = GpuDft.with_backend(BackendKind.Hip)
        self.mmcs = MerkleTreeMmcs(hash)
        self.host_x = benchmark_input(self.h, width)
        self.x = dev_u32(self.host_x)
        self.lde = torch.empty((self.h << log_blowup, width), dtype=torch.int32, device="cuda")
        self.batch = 1
        self.threads = 1
        self.last_root = None
        torch.cuda.synchronize()

    def close(self):
        pass

    def metric_name(self):
        return "wide-trace coset LDE + MMCS commits/sec"

    def unit(self):
        return "commits/s"

    def workload_name(self):
        return "2^%d x %d trace (benchmark_input), blowup %d, bit-reversed coset LDE + %s MMCS commit (BASELINE configs[4])" % (
            self.log_height, self.width, 1 << self.log_blowup, "Poseidon2" if self.hash == "poseidon2" else "Keccak")

    def config(self):
        return {"log_height": self.log_height, "width": self.width, "log_blowup": self.log_blowup, "batch_per_gpu": 1}

    def _lde(self):
        L = _lib.lib()
        _lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(self.x.data_ptr()), C.c_void_p(self.lde.data_ptr()), self.h,
                                                    self.width, self.log_blowup, GENERATOR_MONTY, 1, _stream_ptr()))

    def _commit(self):
        root, t = self.mmcs.commit([self.lde])
        t.free()
        return root

    def step(self, instances=None):
        self._lde()
        self.last_root = self._commit()
        return self.last_root

    def step_begin(self, instances=None, sink=None):  # one matrix per step: nothing to overlap
        self._res = self.step()

    def step_end(self):
        return self._res

    _time = FibAirJob._time

    def roofline(self):
        H = self.h << self.log_blowup
        lde_bytes = 4 * self.h * self.width * (1 + (1 << self.log_blowup))
        commit_bytes = 4 * H * self.width + 32 * (2 * H - 1)
        t_lde = self._time(self._lde, 10)
        t_commit = self._time(self._commit, 10)
        perms = H * ((self.width + 7) // 8) + H - 1 if self.hash == "poseidon2" else H * ((((self.width + 1) // 2) + 16) // 17) + H - 1
        traffic, src = None, None
        try:
            name, pmc, stale = _profile_json(("r05_pmc_lde.json", "r04_pmc_lde.json", "r03_pmc_lde.json"))
            if (self.log_height, self.width, self.log_blowup) == (16, 2633, 1) and "cfg5_lde_2^16x2633_blowup2" in pmc:
                traffic = pmc["cfg5_lde_2^16x2633_blowup2"]["total_bytes"]
                src = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, separate passes, summed over the unit's launches "
                       "(tools/pmc_probe.py, tools/pmc_summarize.py).  FETCH_SIZE x2 (gfx950) for K1 / K2, x1 for K3: a calibration kernel with "
                       "K3's access shape (4-byte lanes on 128-byte segments of 10532-byte rows) shows those reads tallied in full "
                       "(profiles/r04_fetch_calib.json: factor 1.016 against 2.000 for the same access on 128-byte-multiple rows), so round 3's "
                       "2.73 GB for K3's 1.38 GB input was the correction, not a double fetch.  The structure moves 6.2 GB = 9 matrix sweeps; what "
                       "is above that is write amplification on rows that are not multiples of 128 bytes; %s" % (name, _stale_note(stale)))
        except Exception:
            pass
        return {"bound": "hbm", "kernel": "coset_lde_batch of the wide matrix (two-digit plan with 128-byte tile rows: narrow_inv1 / narrow_mid / narrow_fwd2 <8, 5, 1>)",
                "achieved": lde_bytes / (t_lde * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": lde_bytes / (t_lde * 1e-3) / 1e9 / 8000.0,
                "frac_of_achievable_6300": lde_bytes / (t_lde * 1e-3) / 1e9 / 6300.0,
                "algorithmic_bytes": lde_bytes, "avg_us": t_lde * 1e3, "traffic": traffic, "traffic_source": src,
                **(lde_valu_view("cfg5", t_lde * 1e3, lde_bytes) if (self.log_height, self.width, self.log_blowup) == (16, 2633, 1) else {}),
                "commit": {"bound": "valu (hash)", "algorithmic_bytes": commit_bytes, "avg_us": t_commit * 1e3,
                           "achieved_gbps": commit_bytes / (t_commit * 1e-3) / 1e9, "permutations": perms,
                           "gperm_s": perms / (t_commit * 1e-3) / 1e9}}

    def extra_report(self):
        return {}
