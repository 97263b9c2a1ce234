#!/usr/bin/env python3
"""Writes tests/golden/*.json.  Fixtures are DATA (inputs + expected outputs in CANONICAL
integers), produced by tests/pyref.py (big-int O(n^2) mathematics, independent of the C oracle
and of the HIP kernels) — except `poseidon2_bb16_kat.json`, whose input/expected pair is the
known-answer vector of upstream Plonky3's own test `test_poseidon2_width_16_random`
(p3-baby-bear, constants from Xoroshiro128Plus::seed_from_u64(1)); upstream source is absent
from the container, the vector was supplied from memory and is reproduced exactly by pyref.

Usage: python tests/golden/make_golden.py      (rewrites the json files next to it)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import pyref as R  # noqa: E402


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")


def main():
    import random
    rnd = random.Random(20261004)
    # field ops
    pairs = [(0, 0), (1, 1), (R.P - 1, R.P - 1), (R.P - 1, 1), (2, R.inv(2))] + \
            [(rnd.randrange(R.P), rnd.randrange(R.P)) for _ in range(64)]
    dump("field_ops.json", {
        "p": R.P, "monty_r_mod_p": R.R % R.P, "monty_r2_mod_p": R.R * R.R % R.P,
        "two_adic_generator": {str(b): R.two_adic_generator(b) for b in range(0, 28)},
        "cases": [{"a": a, "b": b, "add": (a + b) % R.P, "sub": (a - b) % R.P, "mul": a * b % R.P,
                   "inv_a": R.inv(a) if a else 0, "monty_a": R.to_monty(a)} for a, b in pairs]})
    # DFT: reference benchmark_input (fib_air.rs:77-86) on small shapes + random
    dft = []
    for h, w in [(1, 3), (2, 2), (4, 1), (8, 2), (16, 5), (64, 3), (256, 8), (512, 2)]:
        m = R.benchmark_input(h, w)
        dft.append({"h": h, "w": w, "input": "benchmark_input", "out": R.naive_dft(m)})
    m = [[rnd.randrange(R.P) for _ in range(4)] for _ in range(128)]
    dft.append({"h": 128, "w": 4, "input": m, "out": R.naive_dft(m)})
    dump("dft.json", dft)
    # coset LDE (natural order), shift = generator 31 and another shift
    lde = []
    for h, w, ab, shift in [(1, 2, 1, 31), (2, 2, 1, 31), (8, 2, 1, 31), (8, 2, 2, 31), (32, 3, 1, 31),
                            (64, 2, 2, 31), (16, 4, 1, 1), (16, 1, 3, 1234567)]:
        m = R.fib_trace(0, 1, h) if w == 2 else R.benchmark_input(h, w)
        lde.append({"h": h, "w": w, "added_bits": ab, "shift": shift,
                    "input": "fib_trace" if w == 2 else "benchmark_input",
                    "out": R.coset_lde(m, ab, shift)})
    dump("coset_lde.json", lde)
    # fib trace
    dump("fib_trace.json", {"a": 0, "b": 1, "n": 64, "rows": R.fib_trace(0, 1, 64),
                            "n8_last_right": R.fib_trace(0, 1, 8)[-1][1]})
    # Poseidon2
    kat_in = [894848333, 1437655012, 1200606629, 1690012884, 71131202, 1749206695, 1717947831,
              120589055, 19776022, 42382981, 1831865506, 724844064, 171220207, 1299207443,
              227047920, 1783754913]
    kat_out = [1255099308, 941729227, 93609187, 112406640, 492658670, 1824768948, 812517469,
               1055381989, 670973674, 1407235524, 891397172, 1003245378, 1381303998, 1564172645,
               1399931635, 1005462965]
    ei, it, ef = R.rng_rc(1)
    assert R.poseidon2(kat_in, (ei, it, ef)) == kat_out, "upstream KAT not reproduced"
    dump("poseidon2_bb16_kat.json", {
        "provenance": "Plonky3 p3-baby-bear test_poseidon2_width_16_random (upstream, recalled); "
                      "constants = Poseidon2::new_from_rng_128(Xoroshiro128Plus::seed_from_u64(1))",
        "rc_ext_init": ei, "rc_internal": it, "rc_ext_final": ef, "input": kat_in, "expected": kat_out})
    dei, dit, def_ = R.DEFAULT_RC
    vecs = [[0] * 16, list(range(16)), kat_in, [R.P - 1] * 16]
    dump("poseidon2_bb16_default.json", {
        "provenance": "round constants: published Grain-LFSR procedure (tools/gen_poseidon2_rc.py), "
                      "first row == recalled BABYBEAR_RC16_EXTERNAL_INITIAL[0]; outputs by tests/pyref.py",
        "rc_ext_init_row0": dei[0], "rc_internal": dit, "rc_ext_final_row3": def_[3],
        "cases": [{"input": v, "expected": R.poseidon2(v)} for v in vecs]})
    # sponge / compress / merkle
    rows = [[rnd.randrange(R.P) for _ in range(n)] for n in (0, 1, 2, 7, 8, 9, 16, 17, 24)]
    mm = []
    for dims in ([(8, 2)], [(4, 3), (4, 5)], [(8, 2), (4, 4), (1, 9)], [(1, 2)], [(2, 8), (2, 8)], [(16, 4), (2, 1)]):
        mats = [[[rnd.randrange(R.P) for _ in range(w)] for _ in range(h)] for h, w in dims]
        mm.append({"dims": dims, "mats": mats, "layers": R.merkle_layers(mats)})
    dump("mmcs.json", {"hash_row": [{"items": r, "digest": R.hash_row(r)} for r in rows],
                       "compress": [{"l": rows[4], "r": rows[4][::-1], "digest": R.compress(rows[4], rows[4][::-1])}],
                       "trees": mm})


if __name__ == "__main__":
    main()
