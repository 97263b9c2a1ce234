#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): fib_air proofs/sec + LDE achieved-HBM GB/s, BabyBear 2^20-row trace,
blowup 2, on N MI355X (one process per GPU; proofs are independent, so ranks shard the batch with no
data-path collective — weak scaling).

  python bench.py --gpus N --steps K --warmup W          (N>1 under torch.distributed.run: WORLD_SIZE/RANK from the env;
                                                           N>1 WITHOUT a launcher: this process starts N rank processes itself,
                                                           before it has touched the GPU, relays rank 0's line, exits with their rc)
  python bench.py --workload cfg3|cfg4|cfg5 ...           (the other BASELINE configs, same JSON schema)

Workloads (BASELINE.json `configs`):
  cfg2 (default)  configs[1]: fib_air 2^20-row trace, blowup 2.  A "step" proves `--batch` independent instances
                  (a, b) = (i, i+1) per rank with all inputs generated in HBM.
  cfg4            configs[3]: a batch of 64 independent 2^20 proofs in TOTAL per step, instance i -> rank i mod N
                  (strong scaling: the per-rank share shrinks as N grows); rank 0 scatters descriptors, gathers proof bytes.
  cfg3            configs[2]: fib_air 2^24-row trace, blowup 4 (FRI-fold-heavy); step = `--batch` proofs.
  cfg5            configs[4]: the wide trace 2^16 x 2633 (benchmark_input, fib_air.rs:77-86, standing in for the
                  Keccak-f AIR trace): step = one bit-reversed coset LDE (blowup 2) + Poseidon2 MMCS commit of
                  the 2^17-row result.
Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects: `roofline` (the
dominant HBM-bound unit, algorithmic bytes per SURVEY.md §8d, timed with HIP events on the launch stream) and
`cpu_baseline` (the C oracle timed on the host cores)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# concurrent provers each own a stream; give them distinct hardware queues (ROCm default is 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBPS = 6300.0  # same guide: ~6.3 TB/s achievable


def cpu_baseline_fib(log_height, job, hash_kind=0, hiding=False):
    """The reference CPU prover (Rust + Plonky3) cannot be built here, so the baseline is the
    repo's C restatement (oracle/, kind "port"), single-threaded like the reference build
    (native/Cargo.toml:32-43 enables no `parallel` feature), timed on this host on the SAME instance.
    Bounded sample: one full proof at 2^20 (about 20 s of CPU); at larger sizes a 2^20 proof stands in and the
    sample says so (the prover is O(n log n): the figure is NOT extrapolated)."""
    from oracle import oracle as o
    o.build()
    o.use_native()
    fp = o.FriParams(*[getattr(job.params, k) for k in ("log_blowup", "log_final_poly_len", "num_queries",
                                                         "proof_of_work_bits")])
    sample_log = min(log_height, 18 if hiding else 20)  # the hiding prover commits three times the columns on twice the rows
    prove = (lambda *a, **k: o.prove_fib_air_hiding(*a, seed=1, **k)) if hiding else o.prove_fib_air
    verify = o.verify_fib_air_hiding if hiding else o.verify_fib_air
    o.set_threads(1)
    t0 = time.perf_counter()
    proof = prove(0, 1, sample_log, fp, hash=hash_kind)
    dt = time.perf_counter() - t0
    # the same port with its OpenMP loops (Merkle layers, quotient, openings, folds) on every host core
    # the GPU box gives one GPU's job a share of about 16 host cores whatever nproc says
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = o.max_threads()
    cores = max(1, min(o.max_threads(), avail, int(os.environ.get("P3HIP_BENCH_CPU_THREADS", "16"))))
    o.set_threads(cores)
    t1 = time.perf_counter()
    proof_mt = prove(0, 1, sample_log, fp, hash=hash_kind)
    dt_mt = time.perf_counter() - t1
    o.set_threads(1)
    out = {"value": 1.0 / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
           "sample": "1 full fib_air proof, instance (a,b)=(0,1), 2^%d rows, same FRI parameters" % sample_log,
           "seconds": dt, "proof_bytes": len(proof), "build_flags": o.build_flags(),
           "all_cores": {"value": 1.0 / dt_mt, "unit": "proofs/s", "cores": cores, "seconds": dt_mt,
                         "same_bytes": bool(proof_mt == proof),
                         "note": "same C port, OpenMP over the hashing/opening/folding loops; transforms threaded over blocks of rows / the butterflies of a stage since round 5 (oracle/dft.c); the transcript, the proof-of-work search and (hiding) the random draws are serial"}}
    if sample_log == log_height:
        gpu = job.prove_one(0, 1)
        out["proof_bytes_equal_to_gpu"] = bool(gpu == proof)
        out["oracle_verifier_accepts_gpu_proof"] = bool(
            verify(gpu, 0, 1, o.fib_public_x(0, 1, 1 << log_height), log_height, fp, hash=hash_kind) == 0)
    else:
        out["sample"] += " (the 2^%d instance itself takes minutes on one core: a 2^%d proof is the bounded sample)" % (
            log_height, sample_log)
        gpu = job.prove_one(0, 1)
        out["oracle_verifier_accepts_gpu_proof"] = bool(
            verify(gpu, 0, 1, o.fib_public_x(0, 1, 1 << log_height), log_height, fp, hash=hash_kind) == 0)
    return out


def cpu_baseline_wide(job):
    """Bounded sample on one core: the LDE of 128 of the 2633 columns and the leaf hashes of 2^12 of the 2^17 rows,
    through the C oracle (native build), checked against the GPU result; seconds scaled to the whole job are
    reported beside the measured sample."""
    import numpy as np
    import torch
    from plonky3_mobile_amd.gpu_dft import GENERATOR_MONTY
    from oracle import oracle as o
    o.build()
    o.use_native()
    o.set_threads(1)
    cols = np.arange(0, job.width, job.width // 128)[:128]
    sub = np.ascontiguousarray(job.host_x[:, cols])
    t0 = time.perf_counter()
    exp = o.coset_lde_batch(sub, job.log_blowup, GENERATOR_MONTY, True)
    t_lde = time.perf_counter() - t0
    job._lde()
    torch.cuda.synchronize()
    got = job.lde[:, torch.from_numpy(cols).cuda()].contiguous().cpu().numpy().view(np.uint32)
    same = bool(np.array_equal(got, exp))
    rows = job.lde[: 1 << 12].contiguous().cpu().numpy().view(np.uint32)
    hr = o.hash_row if job.hash == "poseidon2" else o.keccak_hash_row
    t1 = time.perf_counter()
    digs = np.stack([hr(r) for r in rows])
    t_hash = time.perf_counter() - t1
    root, tree = job.mmcs.commit([job.lde])
    leaves = tree.digest_layers()[0][: 1 << 12]
    tree.free()
    same_d = bool(np.array_equal(leaves, digs))
    H = job.h << job.log_blowup
    est = t_lde * job.width / len(cols) + t_hash * H / (1 << 12)
    return {"value": 1.0 / est, "unit": "commits/s", "cores": 1, "kind": "port", "build_flags": o.build_flags(),
            "sample": "coset LDE of %d of the %d columns (%.2f s) + leaf hashes of 2^12 of the 2^%d rows (%.2f s), scaled "
                      "linearly to the whole matrix (compression layers, < 1 %% of the hashing, not included)" % (
                          len(cols), job.width, t_lde, job.log_height + job.log_blowup, t_hash),
            "seconds_estimated_whole_job": est, "lde_columns_equal_to_gpu": same, "leaf_digests_equal_to_gpu": same_d}


HOST_CORES, PINNING = [], "not evaluated"
PG_NOTE = ""


def other_workloads(budget_s=240.0, child_timeout_s=120.0):
    """The default run (what the driver records) also carries SHORT runs of the other single-GPU BASELINE configs and of the
    reference's own configuration, so that their numbers are in a driver-run record and not only in builder-run profiles:
    each is this same script in a child process (`--no-cpu-baseline --no-extras`, few steps), its line condensed to a few
    numbers (the entries are the LAST thing in the line, the two BASELINE configs last of all, so that a reader who keeps only
    the tail of the line still sees them).  A child gets at most `child_timeout_s`; once `budget_s` is spent the remaining ones
    are recorded as skipped.  A child that fails leaves its error text; the main line does not depend on them."""
    import subprocess
    runs = [("cfg2 under the reference's Keccak hashes", ["--hash", "keccak", "--steps", "6", "--warmup", "1"]),
            ("the reference's own configuration (Keccak + hiding), 2^20 rows", ["--hash", "keccak", "--hiding", "--steps", "6", "--warmup", "2"]),
            ("cfg3", ["--workload", "cfg3", "--steps", "2", "--warmup", "1"]),
            ("cfg5", ["--workload", "cfg5", "--steps", "6", "--warmup", "2"])]
    res = []
    t_all = time.perf_counter()
    for label, extra in runs:
        t0 = time.perf_counter()
        entry = {"label": label, "args": " ".join(extra)}
        left = budget_s - (t0 - t_all)
        if left < 20.0:
            entry["skipped"] = "time budget of the extras spent"
            res.append(entry)
            continue
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + ["--no-cpu-baseline", "--no-extras"],
                               capture_output=True, text=True, timeout=min(child_timeout_s, left))
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                entry["error"] = "exit code %d: %s" % (p.returncode, p.stderr.strip()[-200:])
            else:
                d = json.loads(lines[-1])
                r, v = d.get("roofline") or {}, d.get("valu_roofline") or {}
                rnd = lambda x, n=4: round(x, n) if isinstance(x, (int, float)) else x  # noqa: E731
                entry.update({"value": rnd(d.get("value"), 2), "unit": d.get("unit"), "steps": d.get("steps"),
                              "lde_us": rnd(r.get("avg_us"), 1), "lde_gbps": rnd(r.get("achieved"), 1), "lde_frac": rnd(r.get("frac")),
                              "lde_valu_frac": rnd(r.get("valu_frac")), "lde_traffic": r.get("traffic")})
                if v:
                    entry.update({"proof_valu_issue_frac": rnd(v.get("frac")), "valu_M_instr_per_proof": rnd((v.get("valu_wave_instr_per_proof") or 0) / 1e6, 1) or None,
                                  "sustained_gperm_s": rnd(v.get("sustained_gperm_s"), 3),
                                  "sustained_frac_of_kernel_ceiling": rnd(v.get("sustained_frac_of_kernel_ceiling"))})
                if "commit" in r:
                    entry["commit_gperm_s"] = rnd(r["commit"].get("gperm_s"), 3)
        except subprocess.TimeoutExpired:
            entry["error"] = "timed out after %.0f s" % min(child_timeout_s, left)
        except Exception as e:  # noqa: BLE001
            entry["error"] = repr(e)[:200]
        entry["wall_s"] = round(time.perf_counter() - t0, 1)
        res.append(entry)
    return res


def parse_cpulist(text):
    """'0-15,64-79' -> {0..15, 64..79} (the kernel's cpulist format)."""
    cores = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cores.update(range(int(lo), int(hi or lo) + 1))
    return cores


def pin_to_gpu_local_cores(torch, device_index, world, sysfs="/sys/bus/pci/devices"):
    """Multi-rank runs: restrict this rank (and every thread it starts afterwards) to the host cores local to its GPU, read from
    /sys/bus/pci/devices/<bdf>/local_cpulist, intersected with the affinity mask the launcher gave us.  Anything unreadable or an
    empty intersection: nothing is changed.  Returns (cores in the mask afterwards, description for the JSON line)."""
    have = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    if world <= 1 and os.environ.get("P3HIP_BENCH_PIN") != "1":
        return have, "none (single rank: the launcher's mask is kept)"
    if os.environ.get("P3HIP_BENCH_PIN") == "0":
        return have, "none (P3HIP_BENCH_PIN=0)"
    try:
        if os.environ.get("P3HIP_BENCH_PIN_CPULIST"):  # tests: stands in for the sysfs file
            local, src = parse_cpulist(os.environ["P3HIP_BENCH_PIN_CPULIST"]), "P3HIP_BENCH_PIN_CPULIST"
        else:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            src = os.path.join(sysfs, bdf, "local_cpulist")
            with open(src) as f:
                local = parse_cpulist(f.read())
        want = sorted(local & set(have))
        if not want:
            return have, "none (%s has no core inside this process's mask)" % src
        os.sched_setaffinity(0, want)
        return want, "pinned to the %d core(s) of %s inside the launcher's mask" % (len(want), src)
    except Exception as e:  # noqa: BLE001 - unreadable sysfs, no such attribute, not permitted: leave the mask alone
        return have, "none (%s: %s)" % (type(e).__name__, e)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes (one per GPU) with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's JSON line, return the worst exit code.  This parent runs no
    kernel and never execs (it only counts devices; the ranks are fresh child processes either way)."""
    import socket
    import subprocess
    backend = os.environ.get("P3HIP_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d but this node shows %d GPU(s); refusing to fake an N-GPU line "
                  "(P3HIP_BENCH_BACKEND=gloo rehearses the N-rank path on fewer GPUs)" % (n, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), P3HIP_BENCH_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that dies before the rendezvous would leave the others waiting for it: once one has failed, the rest get 20 s
    deadline = None
    while any(p.poll() is None for p in procs):
        if deadline is None and any(p.poll() not in (None, 0) for p in procs):
            deadline = time.monotonic() + 20.0
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()  # exactly the children started above
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(5.0)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    if bad:
        print("bench.py: rank exit codes %r" % (rcs,), file=sys.stderr)
        return max(abs(rc) for rc in bad) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="BASELINE configs[1] (default, the headline), configs[2], configs[3] (64 proofs in total per step over the "
                         "ranks) or configs[4]")
    ap.add_argument("--log-height", type=int, default=None)
    ap.add_argument("--log-blowup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="independent proofs per rank per step")
    ap.add_argument("--threads", type=int, default=None, help="concurrent provers (host threads/streams) per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="default run only: skip the short runs of the other BASELINE configs that are added to the line as `other_workloads`")
    ap.add_argument("--hash", choices=["poseidon2", "keccak"], default="poseidon2",
                    help="poseidon2 = BASELINE.json's configuration (default); keccak = the hashes the reference itself wires")
    ap.add_argument("--hiding", action="store_true",
                    help="the reference's hiding configuration (MerkleTreeHidingMmcs + HidingFriPcs, fib_air.rs:40-65); with --hash keccak "
                         "this is exactly what the reference runs")
    args = ap.parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and args.gpus > 1 and int(env_world) != args.gpus:
        print("bench.py: --gpus %d contradicts WORLD_SIZE=%s" % (args.gpus, env_world), file=sys.stderr)
        sys.exit(2)
    defaults = {"cfg2": dict(log_height=20, log_blowup=1, batch=32, threads=4, steps=40, warmup=2),
                "cfg4": dict(log_height=20, log_blowup=1, batch=None, threads=4, steps=10, warmup=1),
                "cfg3": dict(log_height=24, log_blowup=2, batch=4, threads=2, steps=3, warmup=1),
                "cfg5": dict(log_height=16, log_blowup=1, batch=1, threads=1, steps=10, warmup=2)}[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    CFG4_TOTAL = 64  # BASELINE configs[3]: "batch of 64 independent fib_air 2^20 proofs"
    if args.workload == "cfg4" and args.batch is None:
        if CFG4_TOTAL % world:
            print("bench.py: cfg4 splits 64 proofs over the ranks; %d ranks do not divide 64" % world, file=sys.stderr)
            sys.exit(2)
        args.batch = CFG4_TOTAL // world
    for k, v in defaults.items():
        if getattr(args, k) is None:
            setattr(args, k, v)

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    p3 = load_package()

    # P3HIP_BENCH_STUB=1 (tests only): the CLI, the rank launcher and the scatter / gather loop with a job that proves
    # nothing and touches no GPU; its line says so in `metric` and `data` and can not be mistaken for a measurement.
    stub = os.environ.get("P3HIP_BENCH_STUB") == "1"
    n_dev = torch.cuda.device_count()
    if not stub:
        torch.cuda.set_device(local_rank % max(n_dev, 1))
    # before any worker thread exists: threads inherit the mask
    global HOST_CORES, PINNING
    HOST_CORES, PINNING = pin_to_gpu_local_cores(torch, local_rank % max(n_dev, 1), world)
    # backend "nccl" IS RCCL on ROCm.  P3HIP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with
    # fewer GPUs than ranks (collectives then move host tensors).
    backend = os.environ.get("P3HIP_BENCH_BACKEND", "nccl")
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    # P3HIP_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, scatter, gather, all_reduce) even with
    # one rank — the only way to put RCCL itself through this code on a one-GPU box.
    use_dist = world > 1 or os.environ.get("P3HIP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        from plonky3_mobile_amd import batch as _pbatch  # p3 (load_package above) registered the package
        global PG_NOTE
        PG_NOTE = _pbatch.init_process_group_for_batches(backend, local_rank)
    if stub:
        if backend == "nccl":
            raise RuntimeError("P3HIP_BENCH_STUB=1 needs P3HIP_BENCH_BACKEND=gloo")
        if os.environ.get("P3HIP_BENCH_STUB_FAIL") == "1" and rank == world - 1:
            raise RuntimeError("P3HIP_BENCH_STUB_FAIL=1: this rank fails on purpose (launcher test)")
        job = StubJob(args.batch, first_instance=rank * args.batch)
        return run(args, job, dist, world, rank, local_rank, n_dev, backend, coll_dev, use_dist, defaults, stub)
    ok, msg = p3.is_available()
    if not ok:
        raise RuntimeError("no HIP backend, refusing to run a fallback: " + msg)

    from plonky3_mobile_amd import bench_support as bs
    if args.workload == "cfg5":
        job = bs.WideCommitJob(p3, args.log_height, args.log_blowup, hash=args.hash)
    else:
        job = bs.FibAirJob(p3, args.log_height, args.log_blowup, args.batch, first_instance=rank * args.batch,
                           threads=args.threads, hash=args.hash, hiding=args.hiding, config_label={"cfg2": "configs[1]", "cfg3": "configs[2]"}.get(
                               args.workload if (args.log_height, args.log_blowup) == (defaults["log_height"], defaults["log_blowup"]) else ""))
    return run(args, job, dist, world, rank, local_rank, n_dev, backend, coll_dev, use_dist, defaults, stub)


class _DirectSink:
    """Hands the prover threads the staging row of each proof (batch.ProofGatherer): the C prover writes the bytes there itself."""
    direct = True

    def __init__(self, put, rows):
        self.put, self.rows = put, rows

    def buffer(self, i):
        return self.put.row_ptr(self.rows[i])

    def done(self, i, length):
        self.put.set(self.rows[i], i, length)


class StubJob:
    """Stands in for FibAirJob under P3HIP_BENCH_STUB=1 (CPU tests of the CLI / launcher / collectives): a "proof" is 64
    bytes derived from the instance.  Nothing here is a measurement."""

    def __init__(self, batch, first_instance=0):
        self.batch, self.first, self._steps = batch, first_instance, []

    def step_begin(self, instances=None, sink=None):
        todo = instances if instances is not None else [(i, self.first + i) for i in range(self.batch)]
        got = {}
        for i, a in todo:
            pf = (b"STUB" + int(a).to_bytes(8, "little")) * 5 + b"\0" * 4
            if sink is not None:
                sink(i, pf)
            got[i] = pf
        self._steps.append(got)

    def step_end(self):
        return self._steps.pop(0)

    def prove_one(self, a, b):
        return (b"STUB" + int(a).to_bytes(8, "little")) * 5 + b"\0" * 4

    def metric_name(self):
        return "STUB (no proofs were computed)"

    def unit(self):
        return "stub-units/s"

    def config(self):
        return {"batch_per_gpu": self.batch}

    def workload_name(self):
        return "STUB job (P3HIP_BENCH_STUB=1): CLI / launcher / scatter-gather rehearsal without a GPU"

    def roofline(self):
        return None

    def extra_report(self):
        return {}

    def close(self):
        pass


def run(args, job, dist, world, rank, local_rank, n_dev, backend, coll_dev, use_dist, defaults, stub):
    import torch

    def gpu_sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        gpu_sync()
        if use_dist:
            dist.barrier()
        gpu_sync()

    from plonky3_mobile_amd import batch as pbatch
    n_total = args.batch * world
    sharded = args.workload != "cfg5" and use_dist
    allow_local = os.environ.get("P3HIP_BENCH_ALLOW_LOCAL") == "1"
    coll_state = {"mode": ("rccl scatter/gather" if backend == "nccl" else backend + " scatter/gather (rehearsal, not RCCL)")
                  if sharded else ("none (single rank)" if world == 1 else "none (replicas only: one matrix per rank)")}
    if sharded and os.environ.get("P3HIP_BENCH_NO_GATHER"):
        coll_state["mode"] = "disabled by P3HIP_BENCH_NO_GATHER: local sharding only"

    # where a rank's wall time goes besides proving (seconds, timed region only): blocked in the scatter of the next step's
    # descriptors, blocked in retire() on the previous step's gather, blocked waiting for its own provers to finish a step
    waits = {"scatter_wait_s": 0.0, "gather_wait_s": 0.0, "prover_join_s": 0.0, "issue_host_s": 0.0, "gather_launch_s": 0.0}
    timing = {"on": False}

    def timed(key, fn):
        t_in = time.perf_counter()
        r = fn()
        if timing["on"]:
            waits[key] += time.perf_counter() - t_in
        return r

    def scatter_run(first, count):
        """ONE scatter for a whole run of `count` steps (one scatter in): every step's descriptors are known when the
        run starts.  It is issued inside the timed region (run_steps), on the collective stream, and nothing blocks on it: the first
        issue() waits for the event behind its pinned landing copy."""
        steps = [[(k * n_total + i, k * n_total + i + 1) for i in range(n_total)] for k in range(first, first + count)] if rank == 0 else []
        coll_state["descriptors"] = timed("scatter_wait_s", lambda: coll_state["scatterer"].run(steps))
        coll_state["descriptors_first"] = first
        coll_state["scatters"] = coll_state.get("scatters", 0) + 1

    def descriptors(k):
        return timed("scatter_wait_s", lambda: coll_state["descriptors"].step(k - coll_state["descriptors_first"]))

    # Steps are PIPELINED (depth 1): step k + 1 is issued — its instances dealt to the prover threads' queue — before
    # step k retires, so a prover that has finished its share of step k starts on step k + 1 at once instead of
    # idling until the slowest proof of the step is done (the per-step join cost ~8 % at 32 proofs per step).  All K
    # steps complete inside the timed region (the last one is retired and its proofs collected before the closing barrier).
    # Multi-rank (BASELINE configs[3]): rank 0 scatters the instance descriptors, every rank proves its shard (instance i
    # -> rank i mod world) with the prover threads writing each proof straight into the pinned staging row of the step's
    # gather, the proof bytes are gathered back on rank 0.  Order of collectives on every rank: scatter(k+1), gather(k);
    # the previous step's proofs are collected WHILE the next one is being proved.  No other collective.
    use_gather = sharded and "scatter/gather" in coll_state["mode"]
    if use_gather:
        try:
            width = torch.tensor([len(job.prove_one(0, 1))], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(width, op=dist.ReduceOp.MAX)  # proofs of one parameter set have one length: the slot width
            coll_state["width"] = int(width.item())
            coll_state["gatherer"] = pbatch.ProofGatherer(n_total, coll_dev, width=coll_state["width"])
            # every buffer of the run's one scatter exists before the timed region starts
            coll_state["scatterer"] = pbatch.DescriptorScatter(max(args.steps, args.warmup, 1), n_total, coll_dev)
        except Exception as e:
            print("bench.py: collective setup failed on rank %d: %r" % (rank, e), file=sys.stderr)
            if not allow_local:
                sys.stderr.flush()
                os._exit(3)
            use_gather = False
            coll_state["mode"] = "FALLBACK to local sharding (collective failed: %s; P3HIP_BENCH_ALLOW_LOCAL=1)" % type(e).__name__

    def issue(k):
        if not sharded:
            job.step_begin()
            return None
        if not use_gather:
            job.step_begin([(i, k * n_total + i) for i in pbatch.shard_instances(n_total, rank, world)])
            return None
        mine = descriptors(k)

        def begin():
            rows = {i: r for r, (i, _, _) in enumerate(mine)}
            put, slot = coll_state["gatherer"].open(len(mine), coll_state["width"])
            if stub or os.environ.get("P3HIP_BENCH_DIRECT_SINK", "1") != "1":
                job.step_begin([(i, a) for i, a, _ in mine], lambda i, pf: put(rows[i], i, pf))
            else:
                job.step_begin([(i, a) for i, a, _ in mine], _DirectSink(put, rows))
            return slot
        return timed("issue_host_s", begin)

    def retire(slot):
        got = timed("prover_join_s", job.step_end)
        if slot is None:
            return got

        prev, coll_state["pending"] = coll_state.get("pending"), timed("gather_launch_s", lambda: coll_state["gatherer"].launch(slot))
        return timed("gather_wait_s", lambda: prev.wait(copy=False) if prev is not None else None)

    def run_steps(first, count):
        # A failed collective is FATAL: an N-GPU line must never be printed without RCCL having moved the batch.
        try:
            if count <= 0:
                return
            if use_gather:
                scatter_run(first, count)
            inflight = [issue(first)]
            for k in range(first + 1, first + count):
                inflight.append(issue(k))
                retire(inflight.pop(0))
            retire(inflight.pop(0))
            pend = coll_state.pop("pending", None)
            if pend is not None:
                timed("gather_wait_s", lambda: pend.wait(copy=False))  # the last step's proofs must be on rank 0 inside the timed region
        except Exception as e:
            if not use_gather:
                raise
            print("bench.py: scatter/gather failed on rank %d: %r" % (rank, e), file=sys.stderr)
            sys.stderr.flush()
            os._exit(3)

    run_steps(0, args.warmup)
    # The issuing thread must not stop for the interpreter's cyclic garbage collector inside the timed region: with torch imported a
    # full collection walks ~10^6 objects (tens of milliseconds), and the multi-rank path allocates enough small objects per step
    # (descriptor tuples, views of the gathered rows) to trigger one where the single-rank path does not — it showed as ~35 ms of a
    # 2.2 s run that no timer of the loop owned.  Collect now, then keep the collector off until the run is over.
    import gc
    gc.collect()
    gc.disable()
    barrier()
    timing["on"] = True
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    own_wall = time.perf_counter() - t0  # this rank's own steps, before it waits for the slowest rank
    barrier()
    elapsed = time.perf_counter() - t0
    timing["on"] = False
    gc.enable()
    cur_dev = -1 if stub else torch.cuda.current_device()
    # one row per rank, so that an N-rank line explains itself: what the rank proved, how long its own steps took, where it waited
    # (a scaling loss shows up as one rank's prove_wall_s, or as gather / scatter waits), and which host cores it ran on
    int_keys = ("rank", "local_rank", "device_count", "device", "proofs", "host_cores", "pinned")
    flt_keys = ("prove_wall_s", "scatter_wait_s", "gather_wait_s", "prover_join_s", "issue_host_s", "gather_launch_s")
    mine = [rank, local_rank, n_dev, cur_dev, args.batch * args.steps, len(HOST_CORES), 1 if PINNING.startswith("pinned") else 0,
            own_wall, waits["scatter_wait_s"], waits["gather_wait_s"], waits["prover_join_s"], waits["issue_host_s"], waits["gather_launch_s"]]

    def rank_row(vals):
        d = {k: int(v) for k, v in zip(int_keys, vals)}
        d.update({k: round(float(v), 4) for k, v in zip(flt_keys, vals[len(int_keys):])})
        d["proofs_per_s"] = round(d["proofs"] / d["prove_wall_s"], 2) if d["prove_wall_s"] > 0 else None
        return d
    ranks_info = [rank_row(mine)]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        info = torch.tensor(mine, dtype=torch.float64, device=coll_dev)
        infos = [torch.empty_like(info) for _ in range(world)]
        dist.all_gather(infos, info)
        ranks_info = [rank_row(i.cpu().tolist()) for i in infos]
    ranks_info[rank if use_dist else 0]["host_core_list"] = ",".join(map(str, HOST_CORES[:64])) + ("..." if len(HOST_CORES) > 64 else "")
    ranks_info[rank if use_dist else 0]["pinning"] = PINNING
    units = args.batch * args.steps * world
    value = units / elapsed

    roof = job.roofline()
    out = {
        "metric": job.metric_name(),
        "value": value,
        "unit": job.unit(),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if args.workload == "cfg4" else "weak",
        "vs_baseline": None,
        "dtype": "u32 (BabyBear Montgomery, 31-bit modular)",
        "data": "synthetic" if not stub else "STUB: nothing was proved, no GPU was used (P3HIP_BENCH_STUB=1)",
        "config": dict(job.config(), workload=job.workload_name() + (
                           "; BASELINE configs[3]: %d proofs in total per step, instance i -> rank i mod %d" % (n_total, world)
                           if args.workload == "cfg4" else ""), hash=args.hash, total_batch_per_step=n_total,
                       step_pipelining="depth 1: step k+1's instances are dealt to the prover threads before step k's last proofs "
                                       "finish; all K steps complete inside the timed region",
                       parallelism=("independent proofs, instance i -> rank i mod N; RCCL only scatters descriptors / gathers proof bytes"
                                    if args.workload != "cfg5" else "replicas only (one matrix per rank, no collective)")),
        "roofline": roof,
        "collectives": coll_state["mode"] + ((" [%s; one descriptor scatter per run (%d in the timed region), one gather per step on a "
                                               "highest-priority stream of their own]" % (PG_NOTE, 1)) if use_gather else ""),
        "world_size": world,
        "dist_backend": (backend + (" (= RCCL on ROCm)" if backend == "nccl" else "")) if use_dist else None,
        "ranks": ranks_info,
        "ranks_note": "per rank: proofs and prove_wall_s of its own timed steps (before the closing barrier), seconds blocked in the "
                      "descriptor scatter / in retire() on the previous step's proof gather / joining its own provers, host time spent opening a staging slot and dealing a "
                      "step to the provers (issue_host_s) and enqueuing a step's gather (gather_launch_s), host cores in its affinity mask; "
                      "host_core_list and pinning are rank 0's (every rank applies the same rule to its own GPU)",
        "gather_bytes_per_step": (n_total * coll_state["width"]) if use_gather and "width" in coll_state else 0,
        "descriptor_scatters_in_timed_region": (1 if args.steps > 0 else 0) if use_gather else 0,
        "descriptor_scatter_host_ms": ({k: round(v, 3) for k, v in coll_state["scatterer"].last_ms.items()}
                                       if use_gather and "scatterer" in coll_state else None),
        "parity": "proof bytes / digests are compared with the repo's C oracle (a restatement of upstream Plonky3 from "
                  "recall; pinned by reference code only for the DFT): self-consistent, upstream parity UNPINNED",
    }
    out.update(job.extra_report())
    vr = out.get("valu_roofline")
    if isinstance(vr, dict) and vr.get("valu_wave_instr_per_proof"):
        # the ruler that binds: what the timed region issued, in VALU wave-instructions per second, against the chip's measured issue peak
        vr["achieved"] = value * vr["valu_wave_instr_per_proof"] / 1e9 / world
        vr["frac"] = vr["achieved"] / vr["peak"]
        vr["proof_valu_issue_frac"] = vr["frac"]
        vr["achieved_is"] = "proofs/s of the timed region x VALU wave-instructions per proof (PMC), per GPU"
    if isinstance(vr, dict) and vr.get("permutations_per_proof"):
        # the same in the hash kernels' own unit: proofs/s x permutations per proof (counts the hash layers only: 78-97 % of the instructions)
        vr["sustained_gperm_s"] = value * vr["permutations_per_proof"] / 1e9
        ceiling = vr.get("kernel_ceiling_gperm_s")
        if ceiling:
            vr["sustained_frac_of_kernel_ceiling"] = vr["sustained_gperm_s"] / (ceiling * world)
            vr["kernel_ceiling"] = ("permutations / time of one 2^24 x 2 commit timed in this run (layers of >= 2^20 lanes fill the chip for many "
                                    "workgroup generations: the hash kernels at their own VALU-bound rate), per GPU")
    if world > 1:
        out["cpu_baseline"] = "omitted for world > 1 (the CPU leg is timed by the N=1 run only)"
    elif stub or args.no_cpu_baseline:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not stub:
        if args.workload == "cfg5":
            out["cpu_baseline"] = cpu_baseline_wide(job)
        else:
            out["cpu_baseline"] = cpu_baseline_fib(args.log_height, job, 1 if args.hash == "keccak" else 0, hiding=args.hiding)
    job.close()
    plain_default = (args.workload == "cfg2" and args.hash == "poseidon2" and not args.hiding and
                     (args.log_height, args.log_blowup) == (defaults["log_height"], defaults["log_blowup"]))
    if rank == 0 and world == 1 and not stub and plain_default and not args.no_extras:
        out["other_workloads"] = other_workloads()
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
