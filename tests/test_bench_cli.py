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
    # the line explains itself per rank (round-3 review, item 4): work done, own wall time, where it waited, host cores
    for r in out["ranks"]:
        assert r["proofs"] == 32 * 3 and r["prove_wall_s"] > 0 and r["proofs_per_s"] > 0
        for k in ("scatter_wait_s", "gather_wait_s", "prover_join_s"):
            assert r[k] >= 0.0
        assert r["host_cores"] >= 1 and r["pinned"] in (0, 1)
    assert out["ranks"][0]["pinning"] and out["ranks"][0]["host_core_list"]
    assert out["gather_bytes_per_step"] == 64 * 64  # 64 stub proofs of 64 bytes per step
    assert out["descriptor_scatters_in_timed_region"] == 1 and "one descriptor scatter per run" in out["collectives"]


def test_ranks_pin_to_the_cores_local_to_their_gpu_when_the_list_is_readable():
    """P3HIP_BENCH_PIN_CPULIST stands in for /sys/bus/pci/devices/<bdf>/local_cpulist: every rank narrows its affinity mask to the
    listed cores that are inside the launcher's mask and says so; an unreadable / disjoint list changes nothing."""
    have = sorted(os.sched_getaffinity(0))
    res = _bench(["--gpus", "2", "--steps", "2", "--warmup", "0"],
                 {"P3HIP_BENCH_STUB": "1", "P3HIP_BENCH_BACKEND": "gloo", "P3HIP_BENCH_PIN_CPULIST": "%d,100000-100003" % have[0]})
    assert res.returncode == 0, res.stderr
    out = _line(res)
    assert [r["host_cores"] for r in out["ranks"]] == [1, 1] and [r["pinned"] for r in out["ranks"]] == [1, 1]
    assert out["ranks"][0]["pinning"].startswith("pinned to the 1 core") and out["ranks"][0]["host_core_list"] == str(have[0])
    res = _bench(["--gpus", "2", "--steps", "2", "--warmup", "0"],
                 {"P3HIP_BENCH_STUB": "1", "P3HIP_BENCH_BACKEND": "gloo", "P3HIP_BENCH_PIN_CPULIST": "100000-100003"})
    out = _line(res)
    assert [r["pinned"] for r in out["ranks"]] == [0, 0] and out["ranks"][0]["host_cores"] == len(have)
    assert out["ranks"][0]["pinning"].startswith("none (")


def test_cpulist_parser():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert bench.parse_cpulist("") == set()


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
    r = out["ranks"][0]
    assert r["proofs"] == 128 and r["prove_wall_s"] > 0 and r["gather_wait_s"] == 0.0 and r["scatter_wait_s"] == 0.0
    assert r["pinning"].startswith("none (single rank") and out["gather_bytes_per_step"] == 0


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


def test_other_workloads_condenses_child_lines_and_survives_a_failing_child(monkeypatch):
    """bench.py's default line carries short runs of the other configs (`other_workloads`), each this script in a child process.
    The condensation and the failure handling are checked here with canned children: a child that fails leaves its error text and
    the remaining ones still run."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    good = {"metric": "fib_air proofs/sec", "value": 19.1, "unit": "proofs/s", "steps": 2, "warmup": 1, "ms_per_step": 209.0,
            "config": {"workload": "fib_air 2^24-row trace"},
            "roofline": {"bound": "hbm", "achieved": 735.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.092, "traffic": 2.03e9, "avg_us": 912.0,
                         "algorithmic_bytes": 671088640, "kernel": "coset_lde_batch", "commit": {"gperm_s": 7.5}},
            "valu_roofline": {"hash": "Poseidon2-BabyBear-16", "achieved": 545.6, "frac": 0.97, "sustained_gperm_s": 7.7, "valu_wave_instr_per_proof": 28.6e9}}
    calls = []

    def fake_run(cmd, **kw):
        calls.append(cmd)
        if "cfg5" in cmd:
            return types.SimpleNamespace(returncode=3, stdout="", stderr="boom: no HIP backend")
        return types.SimpleNamespace(returncode=0, stdout="noise\n" + json.dumps(good) + "\n", stderr="")
    monkeypatch.setattr("subprocess.run", fake_run)
    res = bench.other_workloads()
    assert len(res) == 4 and len(calls) == 4
    assert all("--no-cpu-baseline" in c and "--no-extras" in c for c in calls)  # a child never recurses into extras
    assert [e["label"] for e in res][-2:] == ["cfg3", "cfg5"]  # the BASELINE configs last: they must survive a tail of the line
    ok = [e for e in res if "error" not in e]
    bad = [e for e in res if "error" in e]
    assert len(ok) == 3 and len(bad) == 1 and "boom" in bad[0]["error"] and "cfg5" in bad[0]["args"]
    e = ok[0]
    assert e["value"] == 19.1 and e["lde_frac"] == 0.092 and e["commit_gperm_s"] == 7.5 and e["lde_us"] == 912.0
    assert e["proof_valu_issue_frac"] == 0.97 and e["sustained_gperm_s"] == 7.7 and e["valu_M_instr_per_proof"] == 28600.0
    assert len(json.dumps(res)) < 1900  # all four entries fit the last 2000 bytes of the line
    # a spent budget records the rest as skipped instead of starting children
    res = bench.other_workloads(budget_s=0.0)
    assert all("skipped" in e for e in res) and len(calls) == 4


def test_pmc_profiles_are_keyed_by_the_build_and_staleness_is_said(tmp_path, monkeypatch):
    """Round-4 advisor finding: a checked-in PMC count divided by a time measured in this run was presented as a measurement of this run
    whatever the code had become.  Every round-5 PMC summary records the hash of the library's SOURCES (and of the binary); the bench
    support says `same build` only when the tree's sources hash to the same value, `STALE` otherwise, and a profile without an identity
    (round 4 and earlier) is named as such."""
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    p3 = load_package()
    from plonky3_mobile_amd import bench_support as bs
    name, pmc, stale = bs._profile_json(("r05_pmc_proofs.json",))
    assert name == "r05_pmc_proofs.json" and set(pmc["workloads"]) >= {"cfg2", "cfg2_keccak", "cfg2_keccak_hiding", "cfg3"}
    assert len(pmc["src_sha256"]) == 64 and len(pmc["lib_sha256"]) == 64
    assert stale == (pmc["src_sha256"] != p3._lib.src_sha256())  # whichever it is, it is what the hashes say
    view = bs.proof_valu_view("cfg2")
    assert view["valu_wave_instr_per_proof"] > 5e8 and view["top_kernels"][0][0].startswith(("compress_layer_f64", "leaf_hash_f64"))
    assert ("STALE" in view["profile_build"]) == bool(stale)
    monkeypatch.setattr(p3._lib, "_src_sha", "0" * 64)  # another tree: the same profile must now read as stale
    assert bs._profile_json(("r05_pmc_proofs.json",))[2] is True and "STALE" in bs.proof_valu_view("cfg2")["profile_build"]
    monkeypatch.setattr(p3._lib, "_src_sha", None)
    assert bs._profile_json(("r04_pmc_lde_valu.json",))[2] is None  # no identity recorded: said so, not assumed fresh
    assert "predates" in bs._stale_note(None)
