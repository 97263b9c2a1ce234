#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): fib_air proofs/sec + LDE achieved-HBM GB/s, BabyBear 2^20-row trace,
blowup 2, on N MI355X (one process per GPU; proofs are independent, so ranks shard the batch with no
data-path collective — weak scaling).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" proves `--batch` independent fib_air instances (a, b) = (i, i+1) per rank with all inputs
generated in HBM.  Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: `roofline` (the coset-LDE unit, algorithmic bytes 4*h*w*(1+blowup) per SURVEY.md §8d, timed
with HIP events on the launch stream) and `cpu_baseline` (the C oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# concurrent provers each own a stream; give them distinct hardware queues (ROCm default is 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(log_height, log_blowup, job, hash_kind=0):
    """The reference CPU prover (Rust + Plonky3) cannot be built here, so the baseline is the
    repo's C restatement (oracle/, kind "port"), single-threaded like the reference build
    (native/Cargo.toml:32-43 enables no `parallel` feature), timed on this host on the SAME instance.
    Bounded sample: one full proof at the bench size (about 20-30 s of CPU at 2^20)."""
    from oracle import oracle as o
    o.build()
    fp = o.FriParams(*[getattr(job.params, k) for k in ("log_blowup", "log_final_poly_len", "num_queries",
                                                         "proof_of_work_bits")])
    o.set_threads(1)
    t0 = time.perf_counter()
    proof = o.prove_fib_air(0, 1, log_height, fp, hash=hash_kind)
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
    proof_mt = o.prove_fib_air(0, 1, log_height, fp, hash=hash_kind)
    dt_mt = time.perf_counter() - t1
    o.set_threads(1)
    gpu = job.prove_one(0, 1)
    ok = o.verify_fib_air(gpu, 0, 1, o.fib_public_x(0, 1, 1 << log_height), log_height, fp, hash=hash_kind) == 0
    return {"value": 1.0 / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": "1 full fib_air proof, instance (a,b)=(0,1), 2^%d rows, same FRI parameters" % log_height,
            "seconds": dt, "proof_bytes_equal_to_gpu": bool(gpu == proof), "oracle_verifier_accepts_gpu_proof": bool(ok),
            "proof_bytes": len(proof),
            "all_cores": {"value": 1.0 / dt_mt, "unit": "proofs/s", "cores": cores, "seconds": dt_mt,
                          "same_bytes": bool(proof_mt == proof),
                          "note": "same C port, OpenMP over the hashing/opening/folding loops; transforms and transcript serial"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-height", type=int, default=20)
    ap.add_argument("--log-blowup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="independent proofs per rank per step")
    ap.add_argument("--threads", type=int, default=8, help="concurrent provers (host threads/streams) per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hash", choices=["poseidon2", "keccak"], default="poseidon2",
                    help="poseidon2 = BASELINE.json's configuration (default); keccak = the hashes the reference itself wires")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    p3 = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(n_dev, 1))
    # backend "nccl" IS RCCL on ROCm.  P3HIP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with
    # fewer GPUs than ranks (collectives then move host tensors).
    backend = os.environ.get("P3HIP_BENCH_BACKEND", "nccl")
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    ok, msg = p3.is_available()
    if not ok:
        raise RuntimeError("no HIP backend, refusing to run a fallback: " + msg)

    n = 1 << args.log_height
    from plonky3_mobile_amd import bench_support as bs
    job = bs.FibAirJob(p3, args.log_height, args.log_blowup, args.batch, first_instance=rank * args.batch,
                       threads=args.threads, hash=args.hash)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from plonky3_mobile_amd import batch as pbatch
    n_total = args.batch * world

    coll_state = {"mode": ("rccl scatter/gather" if backend == "nccl" else backend + " scatter/gather (rehearsal)") if world > 1 else "none (single rank)"}
    if os.environ.get("P3HIP_BENCH_NO_GATHER"):
        coll_state["mode"] = "disabled by P3HIP_BENCH_NO_GATHER: local sharding only"

    def one_step(k):
        if world == 1:
            return job.step()
        # BASELINE configs[3]: rank 0 scatters the instance descriptors, every rank proves its shard
        # (instance i -> rank i mod world), the proof bytes are gathered back on rank 0.  No other collective.
        if "scatter/gather" in coll_state["mode"] and not coll_state["mode"].startswith("fallback"):
            try:
                inst = [(k * n_total + i, k * n_total + i + 1) for i in range(n_total)] if rank == 0 else []
                mine = pbatch.scatter_descriptors(inst, device=coll_dev)
                got = job.step([(i, a) for i, a, _ in mine])
                # the gather of this step's proofs overlaps the next step's proving; the previous one is collected now
                prev, coll_state["pending"] = coll_state.get("pending"), pbatch.gather_proofs_async(
                    sorted(got.items()), n_total, device=coll_dev)
                return prev.wait() if prev is not None else None
            except Exception as e:  # keep the scaling run alive: the sharding itself needs no collective
                coll_state["mode"] = "fallback to local sharding (collective failed: %s)" % type(e).__name__
                print("bench.py: scatter/gather failed on rank %d: %r" % (rank, e), file=sys.stderr)
        mine = [(i, k * n_total + i) for i in pbatch.shard_instances(n_total, rank, world)]
        return job.step(mine)

    def drain():
        pend = coll_state.pop("pending", None)
        if pend is not None:
            pend.wait()

    for k in range(args.warmup):
        one_step(k)
    drain()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    drain()  # the last step's proofs must be on rank 0 inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    units = args.batch * args.steps * world
    value = units / elapsed

    # ---- roofline of the dominant HBM-bound unit: the coset LDE, HIP events on the launch stream ----
    roof = job.lde_roofline(reps=20)
    traffic, traffic_src = None, None
    try:  # HBM-side bytes of the same LDE unit from the committed PMC passes (not collected live)
        with open(os.path.join(ROOT, "profiles", "r01_pmc_lde_v2.json")) as f:
            pmc = json.load(f)
        key = {(20, 1): "cfg2_lde_2^20x2_blowup2", (24, 2): "cfg3_lde_2^24x2_blowup4"}.get((args.log_height, args.log_blowup))
        if key:
            traffic = pmc[key]["total_bytes"]
            traffic_src = "profiles/r01_pmc_lde_v2.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, summed over the unit's three launches (tools/pmc_probe.py, tools/pmc_summarize.py)"
    except Exception:
        pass
    out = {
        "metric": job.metric_name(),
        "value": value,
        "unit": job.unit(),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 (BabyBear Montgomery, 31-bit modular)",
        "data": "synthetic",
        "config": {"workload": job.workload_name(), "log_height": args.log_height, "width": 2,
                   "log_blowup": args.log_blowup, "hash": args.hash, "batch_per_gpu": args.batch,
                   "concurrent_provers_per_gpu": job.threads,
                   "fri": {"log_final_poly_len": job.params.log_final_poly_len, "num_queries": job.params.num_queries,
                           "proof_of_work_bits": job.params.proof_of_work_bits},
                   "parallelism": "independent proofs, instance i -> rank i mod N; RCCL only scatters descriptors / gathers proof bytes"},
        "roofline": {"bound": "hbm", "achieved": roof["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": roof["gbps"] / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "coset_lde_batch = narrow_inv1_kernel + narrow_mid_kernel + narrow_fwd2_kernel (one unit, three launches)", "algorithmic_bytes": roof["bytes"],
                     "avg_us": roof["avg_us"], "concurrent_gbps": roof.get("concurrent_gbps"),
                     "concurrent_streams": roof.get("concurrent_streams")},
        # The kernel that dominates a proof BY TIME is Poseidon2 (12.6 M permutations per 2^20 proof, ~80 % of the GPU
        # time) and it is integer-VALU-bound, which the contract's hbm|mfma roofline cannot express: reported here
        # against the issue ceiling derived from the measured per-instruction rates.
        "valu_roofline": {"kernel": "Poseidon2 leaf/compress (one state per lane, fp64 integer arithmetic)",
                          "achieved": job.poseidon2_rate() / 1e9,
                          "peak": 7.26, "unit": "Gperm/s", "instructions_per_permutation": 5156,
                          "peak_basis": "5.16k wave-instructions per permutation (ISA count of compress_layer_f64_kernel), "
                                        "nearly all fp64 VALU at 4.2 cycles each, on 1024 SIMDs at 2.4 GHz "
                                        "(profiles/r01_microbench2_valu_issue_rates.txt)",
                          # sustained: tools/clock_probe.hip (profiles/r01_v5_clock_probe_sustained_valu.txt) — under a
                          # chip-wide fp64 FMA load the shader clock settles at ~1.93 GHz and a SIMD retires one fp64
                          # wave-instruction per 4.55 of those cycles: 434 G wave-instructions/s chip-wide
                          "sustained_peak": 434.0e9 / 4850 * 64 / 1e9,
                          "sustained_basis": "434 G fp64 wave-instructions/s measured chip-wide under sustained load "
                                             "(1.93 GHz x 1024 SIMDs / 4.55 cycles) / 4.85k VALU instructions per permutation"},
        "stages_ms": job.stage_breakdown(),
        "collectives": coll_state["mode"],
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["valu_roofline"]["frac"] = out["valu_roofline"]["achieved"] / out["valu_roofline"]["peak"]
        out["valu_roofline"]["frac_of_sustained"] = out["valu_roofline"]["achieved"] / out["valu_roofline"]["sustained_peak"]
        out["cpu_baseline"] = cpu_baseline(args.log_height, args.log_blowup, job, 1 if args.hash == "keccak" else 0)
    job.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
