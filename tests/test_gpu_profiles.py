"""The kernel-path switches (environment variables read once per process) select code that the default configuration no
longer reaches — e.g. the 16-lane cooperative Poseidon2 for layers of 2^12..2^15 digests, the one-state-per-lane Keccak
levels everywhere, large cooperative chunks.  Each variant runs tests/variant_check.py in a child process: trees layer by
layer and proofs byte for byte against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    {"P3HIP_COOP_MAX_LOG": "15", "P3HIP_COOP_CHUNK_LOG": "7"},          # round-1 shape: cooperative below 2^15, 128-digest chunks
    {"P3HIP_COOP_MAX_LOG": "8", "P3HIP_COOP_CHUNK_LOG": "3"},
    {"P3HIP_KECCAK_COOP_MAX_LOG": "0", "P3HIP_KECCAK_LANE_CHUNK_LOG": "11"},  # no cooperative Keccak, 2048-digest workgroups
    {"P3HIP_KECCAK_COOP_MAX_LOG": "15", "P3HIP_KECCAK_COOP_CHUNK_LOG": "5"},
    {"P3HIP_KECCAK_COOP_MAX_LOG": "12"},                                     # round 2's default
    {"P3HIP_NTT_NARROW_COSSPLIT": "0", "P3HIP_NTT_FUSED": "0", "P3HIP_HIDING_PIECEWISE": "1"},
    {"P3HIP_NTT_NARROW_F64": "0"},                                        # integer butterflies at every size
    {"P3HIP_NTT_NARROW_F64": "7", "P3HIP_VARIANT_BIG_LDE": "1"},          # fp64 butterflies at every size, incl. 12-stage digits
    {"P3HIP_NTT_NARROW_F64": "7", "P3HIP_NTT_NARROW_F64_TILES": "1", "P3HIP_NTT_NARROW_VW": "2"},
    {"P3HIP_NTT_NARROW_F64": "5", "P3HIP_NTT_NARROW_F64_TILES": "2", "P3HIP_NTT_NARROW_VW": "1", "P3HIP_NTT_NARROW_COSSPLIT": "1"},
    {"P3HIP_NTT_NARROW_F64": "7", "P3HIP_NTT_NARROW_F64_XW": "1", "P3HIP_VARIANT_BIG_LDE": "1"},  # fp64 rounds, hand-overs on words
    {"P3HIP_RNG_TWO_PASS": "1", "P3HIP_LEAF_WIDE": "0", "P3HIP_HIDING_BARY_SPLIT": "1"},                    # first forms of the RNG fill and of the wide-row leaf kernel
    {"P3HIP_NTT_NARROW_BLOCKED12": "0", "P3HIP_NTT_NARROW_K3_LQ1": "1", "P3HIP_NTT_NARROW_WIDE": "0", "P3HIP_VARIANT_BIG_LDE": "1"},
    {"P3HIP_NTT_NARROW_BLOCKED12": "0", "P3HIP_VARIANT_CFG3_LDE": "1"},   # cfg3's LDE shape on the row-major intermediates
    # round 4's switches: generic row-set leaf kernel for the salted leaves, four-launch RNG fill for small fills too, fills on the
    # prover's own stream; RNG generation with four / two chunks per lane at sizes whose waves are then only partly filled
    {"P3HIP_LEAF_SALTED": "0", "P3HIP_RNG_SMALL": "0", "P3HIP_HIDING_RNG_SIDE": "0"},
    {"P3HIP_RNG_SUB_LOG": "2", "P3HIP_RNG_SMALL": "0"},
    {"P3HIP_RNG_SUB_LOG": "1", "P3HIP_RNG_SMALL": "0"},
    # the latency switches together: the FRI tail (layers of <= 2^7 rows) in one single-workgroup launch, the hiding prover's randomization
    # commitment on a second side stream, one Poseidon2 state per DPP quad for layers of 2^12..2^15 digests
    {"P3HIP_FRI_TAIL": "1", "P3HIP_HIDING_R_SIDE": "1", "P3HIP_Q4_MAX_LOG": "15"},
    {"P3HIP_Q4_MAX_LOG": "14", "P3HIP_Q4_PRIO": "0", "P3HIP_COOP_MAX_LOG": "9"},
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_switch_variant_matches_oracle(env):
    child_env = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py")], env=child_env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
