"""CPU tests of the oracle's Keccak restatement (oracle/keccak.c) — the hash configuration the reference itself
wires into its MMCS (native/src/fib_air.rs:28-38).  The permutation is PINNED: SHA3-256 and Keccak-256 built in
this file on top of oracle.keccak_f must equal python's hashlib (FIPS 202) / the well-known Keccak-256 digests.
The sponge / serialisation / compression conventions are [UPSTREAM-RECALL] (p3-symmetric 0.4.2 is absent)."""
import hashlib

import numpy as np
import pytest

P = 0x78000001


def _sponge_bytes(o, msg, rate, suffix, outlen):
    """byte-level Keccak sponge on top of the oracle permutation (pad10*1 with a domain suffix)."""
    st = np.zeros(25, dtype=np.uint64)
    m = bytearray(msg) + bytes([suffix])
    while len(m) % rate:
        m.append(0)
    m[-1] |= 0x80
    for off in range(0, len(m), rate):
        blk = np.frombuffer(bytes(m[off:off + rate]), dtype="<u8")
        st[: rate // 8] ^= blk
        st = o.keccak_f(st)
    return st.tobytes()[:outlen]


def test_permutation_pinned_by_sha3_and_keccak256(oracle):
    rng = np.random.default_rng(1600)
    for n in [0, 1, 3, 135, 136, 137, 271, 272, 1000]:
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert _sponge_bytes(oracle, msg, 136, 0x06, 32) == hashlib.sha3_256(msg).digest(), n
        assert _sponge_bytes(oracle, msg, 72, 0x06, 64) == hashlib.sha3_512(msg).digest(), n
    # original Keccak padding (what p3-keccak's Keccak256Hash uses): digest of the empty string
    assert _sponge_bytes(oracle, b"", 136, 0x01, 32).hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert _sponge_bytes(oracle, b"abc", 136, 0x01, 32).hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_padding_free_sponge_and_compress_conventions(oracle):
    # empty input: no permutation, all-zero digest (PaddingFreeSponge::hash_iter)
    assert not oracle.keccak_hash_row(np.zeros(0, dtype=np.uint32)).any()
    # one element: low half of lane 0, one permutation
    st = np.zeros(25, dtype=np.uint64)
    st[0] = 0x12345
    exp = oracle.keccak_f(st)[:4].view(np.uint32)
    assert np.array_equal(oracle.keccak_hash_row(np.array([0x12345], dtype=np.uint32)), exp)
    # 34 elements = exactly one full block of 17 lanes: one permutation; 35 elements: two
    rng = np.random.default_rng(5)
    row = rng.integers(0, P, 35, dtype=np.uint32)
    st = np.zeros(25, dtype=np.uint64)
    st[:17] = row[:34].astype(np.uint64)[0::2] | (row[:34].astype(np.uint64)[1::2] << np.uint64(32))
    one = oracle.keccak_f(st)
    assert np.array_equal(oracle.keccak_hash_row(row[:34]), one[:4].view(np.uint32))
    two = one.copy()
    two[0] = np.uint64(row[34])  # overwrite mode: lanes 1.. keep the previous state
    assert np.array_equal(oracle.keccak_hash_row(row), oracle.keccak_f(two)[:4].view(np.uint32))
    # compress = hash of the 8 lanes of the two digests
    l, r = rng.integers(0, 2**32, 8, dtype=np.uint32), rng.integers(0, 2**32, 8, dtype=np.uint32)
    st = np.zeros(25, dtype=np.uint64)
    st[:4] = l.view(np.uint64)
    st[4:8] = r.view(np.uint64)
    assert np.array_equal(oracle.keccak_compress(l, r), oracle.keccak_f(st)[:4].view(np.uint32))


@pytest.mark.parametrize("dims", [[(8, 2)], [(16, 3), (16, 36)], [(32, 2), (8, 5), (1, 4)], [(1, 7)]])
def test_keccak_tree_open_verify(oracle, dims):
    rng = np.random.default_rng(len(dims) * 31 + dims[0][0])
    mats = [rng.integers(0, P, d, dtype=np.uint32) for d in dims]
    root, tree = oracle.mmcs_commit(mats, oracle.HASH_KECCAK)
    proot, _ = oracle.mmcs_commit(mats)
    assert not np.array_equal(root, proot)
    maxh = max(d[0] for d in dims)
    for index in {0, maxh - 1, maxh // 2}:
        rows, path = tree.open_batch(index)
        assert oracle.mmcs_verify_batch(root, dims, index, rows, path, oracle.HASH_KECCAK)
        assert not oracle.mmcs_verify_batch(root, dims, index, rows, path)  # wrong hash configuration
        if rows.size:
            bad = rows.copy()
            bad[0] ^= 1
            assert not oracle.mmcs_verify_batch(root, dims, index, bad, path, oracle.HASH_KECCAK)


def test_keccak256_bytes(oracle):
    rng = np.random.default_rng(256)
    assert oracle.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    for n in [1, 31, 135, 136, 137, 272, 500]:
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.keccak256(msg) == _sponge_bytes(oracle, msg, 136, 0x01, 32), n


@pytest.mark.parametrize("log_n", [1, 3, 6, 9])
def test_keccak_config_prove_then_verify(oracle, log_n):
    """fib_air under the reference's own hashes (Keccak MMCS + SerializingChallenger32 / HashChallenger<Keccak256>,
    fib_air.rs:28-53; non-hiding).  Self-consistency only: the conventions of p3-challenger are recalled, not pinned."""
    fp = oracle.FriParams(1, 0, 12, 6)
    K = oracle.HASH_KECCAK
    proof = oracle.prove_fib_air(0, 1, log_n, fp, hash=K)
    x = oracle.fib_public_x(0, 1, 1 << log_n)
    assert oracle.verify_fib_air(proof, 0, 1, x, log_n, fp, hash=K) == 0
    assert oracle.prove_fib_air(0, 1, log_n, fp, hash=K) == proof
    assert oracle.verify_fib_air(proof, 0, 1, x + 1, log_n, fp, hash=K) != 0
    assert oracle.verify_fib_air(proof, 0, 1, x, log_n, fp) != 0          # Poseidon2 verifier on a Keccak proof
    assert proof != oracle.prove_fib_air(0, 1, log_n, fp)
    # the reference's instance: n = 8, x = 21 (fib_air.rs:56-57)
    if log_n == 3:
        assert x == 21


def test_keccak_config_tampering_and_parameters(oracle):
    K = oracle.HASH_KECCAK
    fp = oracle.FriParams(1, 0, 3, 4)
    proof = oracle.prove_fib_air(0, 1, 4, fp, hash=K)
    x = oracle.fib_public_x(0, 1, 16)
    words = np.frombuffer(proof, dtype=np.uint32)
    rng = np.random.default_rng(1)
    for pos in rng.choice(len(words), size=60, replace=False):
        bad = words.copy()
        bad[pos] = (int(bad[pos]) + 1) % 0x78000001
        assert oracle.verify_fib_air(bad.tobytes(), 0, 1, x, 4, fp, hash=K) != 0, pos
    for t in [(2, 0, 5, 3), (2, 2, 5, 4), (1, 3, 7, 8)]:
        fp = oracle.FriParams(*t)
        proof = oracle.prove_fib_air(5, 8, 7, fp, hash=K)
        assert oracle.verify_fib_air(proof, 5, 8, oracle.fib_public_x(5, 8, 128), 7, fp, hash=K) == 0
