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

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(log_height, log_blowup, job):
    """The reference CPU prover (Rust + Plonky3) cannot be built here, so the baseline is the
    repo's C restatement (oracle/, kind "port"), single-threaded like the reference build
    (native/Cargo.toml:32-43 enables no `parallel` feature), timed on this host on the same workload."""
    import numpy as np
    from oracle import oracle as o
    o.build()
    n = 1 << log_height
    t0 = time.perf_counter()
    if hasattr(o, "prove_fib_air") and job.prover is not None:
        proof = o.prove_fib_air(0, 1, log_height, log_blowup)
        dt = time.perf_counter() - t0
        same = job.prover.proof_bytes(job.prover.prove(0, 1)) == proof
        return {"value": 1.0 / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
                "sample": "1 full fib_air proof of the same instance (a,b)=(0,1), 2^%d rows" % log_height,
                "seconds": dt, "proof_bytes_equal_to_gpu": bool(same)}
    trace = o.generate_trace_rows(0, 1, n)
    lde = o.coset_lde_batch(trace, log_blowup, (31 << 32) % 0x78000001, True)
    t1 = time.perf_counter()
    root, _ = o.mmcs_commit([lde])
    t2 = time.perf_counter()
    gpu_root = job.step()[0] if job.first == 0 else None
    return {"value": 1.0 / (t2 - t0), "unit": "commitments/s", "cores": 1, "kind": "port",
            "sample": "1 trace commitment (trace gen + coset LDE + Poseidon2 Merkle tree), 2^%d rows" % log_height,
            "seconds": t2 - t0, "lde_seconds": t1 - t0, "commit_seconds": t2 - t1,
            "root_equal_to_gpu": bool(gpu_root is not None and np.array_equal(gpu_root, root))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-height", type=int, default=20)
    ap.add_argument("--log-blowup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="independent proofs per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    p3 = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ok, msg = p3.is_available()
    if not ok:
        raise RuntimeError("no HIP backend, refusing to run a fallback: " + msg)

    n = 1 << args.log_height
    from plonky3_mobile_amd import bench_support as bs
    job = bs.FibAirJob(p3, args.log_height, args.log_blowup, args.batch, first_instance=rank * args.batch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    units = args.batch * args.steps * world
    value = units / elapsed

    # ---- roofline of the dominant HBM-bound unit: the coset LDE, HIP events on the launch stream ----
    roof = job.lde_roofline(reps=20)
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
                   "log_blowup": args.log_blowup, "batch_per_gpu": args.batch,
                   "parallelism": "independent proofs sharded across ranks, no collective"},
        "roofline": {"bound": "hbm", "achieved": roof["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": roof["gbps"] / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "coset_lde_batch (ntt_pass_kernel launches)", "algorithmic_bytes": roof["bytes"],
                     "avg_us": roof["avg_us"], "batched_gbps": roof.get("batched_gbps")},
        "stages_ms": job.stage_breakdown(),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.log_height, args.log_blowup, job)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
