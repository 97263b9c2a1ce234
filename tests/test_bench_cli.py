"""bench.py's command line as the driver uses it: `python bench.py --gpus N` must really start N ranks (round 2's
bench parsed --gpus and ignored it).  CPU only: the job is the labelled stub (P3HIP_BENCH_STUB=1), the collectives run
over gloo; what is under test is the launcher, the rank environment, the scatter / gather loop and the JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def _line(res):
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (res.stdout, res.stderr)
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_gathers_on_rank_0():
    res = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"P3HIP_BENCH_STUB": "1", "P3HIP_BENCH_BACKEND": "gloo"})
    assert res.returncode == 0, res.stderr
    out = _line(res)
    assert out["n_gpus"] == 2 and out["world_size"] == 2
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and [r["local_rank"] for r in out["ranks"]] == [0, 1]
    assert out["scaling"] == "weak" and out["config"]["total_batch_per_step"] == 64 and out["config"]["batch_per_gpu"] == 32
    assert out["collectives"].startswith("gloo scatter/gather")
    assert out["cpu_baseline"].startswith("omitted for world > 1")
    assert "STUB" in out["metric"] and "STUB" in out["data"]  # a stub line can not pass for a measurement


def test_cfg4_is_64_proofs_in_total_split_over_the_ranks():
    res = _bench(["--gpus", "2", "--workload", "cfg4", "--steps", "2", "--warmup", "0"],
                 {"P3HIP_BENCH_STUB": "1", "P3HIP_BENCH_BACKEND": "gloo"})
    assert res.returncode == 0, res.stderr
    out = _line(res)
    assert out["scaling"] == "strong" and out["config"]["total_batch_per_step"] == 64 and out["config"]["batch_per_gpu"] == 32
    assert "configs[3]" in out["config"]["workload"]


def test_single_rank_line_is_not_distributed():
    res = _bench(["--steps", "2", "--warmup", "0", "--workload", "cfg4"], {"P3HIP_BENCH_STUB": "1", "P3HIP_BENCH_BACKEND": "gloo"})
    assert res.returncode == 0, res.stderr
    out = _line(res)
    assert out["n_gpus"] == 1 and out["world_size"] == 1 and out["dist_backend"] is None and len(out["ranks"]) == 1
    assert out["config"]["batch_per_gpu"] == 64


def test_refuses_more_ranks_than_gpus_on_the_rccl_backend():
    res = _bench(["--gpus", "2", "--steps", "1"], {})  # this container has no GPU: an N-GPU line must not be faked
    assert res.returncode == 2 and "refusing" in res.stderr and not res.stdout.strip()


def test_gpus_flag_must_agree_with_the_launcher_environment():
    res = _bench(["--gpus", "4", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "P3HIP_BENCH_STUB": "1", "P3HIP_BENCH_BACKEND": "gloo"})
    assert res.returncode == 2 and "contradicts" in res.stderr


def test_a_failing_rank_fails_the_launcher():
    # the stub refuses the RCCL backend: every rank raises, the parent must report a non-zero exit and print no line
    res = _bench(["--gpus", "2", "--steps", "1"], {"P3HIP_BENCH_STUB": "1", "P3HIP_BENCH_BACKEND": "gloo", "P3HIP_BENCH_STUB_FAIL": "1"})
    assert res.returncode != 0 and not [l for l in res.stdout.splitlines() if l.startswith("{")]
