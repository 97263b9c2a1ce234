"""Host-side mirror of Plonky3's Mmcs contract: MerkleTreeMmcs over either hash configuration —
"poseidon2" (north_star: Poseidon2 sponge 16/8/8 + TruncatedPermutation, digest 8 field elements) or "keccak"
(what the reference itself wires at native/src/fib_air.rs:28-51: PaddingFreeSponge<KeccakF,25,17,4> behind
SerializingHasher + CompressionFunctionFromHasher, digest [u64;4]; non-hiding).  commit / open_batch /
get_matrices keep everything device-resident; verify_batch is the verifier's job and lives in the test oracle."""
import ctypes as C

import numpy as np

from . import _lib
from .gpu_dft import _is_torch, _stream_ptr, dev_u32


def poseidon2_permute(states):
    """n x 16 states (torch device tensor in place, or numpy -> new array)."""
    L = _lib.lib()
    if _is_torch(states):
        assert states.is_cuda and states.is_contiguous() and states.shape[-1] == 16
        _lib.check(L.p3hip_poseidon2_permute_dev(C.c_void_p(states.data_ptr()), states.numel() // 16, _stream_ptr()))
        return states
    a = np.ascontiguousarray(states, dtype=np.uint32).copy()
    assert a.shape[-1] == 16
    _lib.check(L.p3hip_poseidon2_permute(a.ctypes.data_as(C.c_void_p), a.size // 16))
    return a


HASH_POSEIDON2, HASH_KECCAK = 0, 1


def keccak_f(states):
    """KeccakF::permute_mut on n x 25 u64 states (torch device int64 tensor in place, or numpy uint64 -> new array)."""
    import torch
    L = _lib.lib()
    if _is_torch(states):
        assert states.is_cuda and states.is_contiguous() and states.shape[-1] == 25 and states.element_size() == 8
        _lib.check(L.p3hip_keccak_f_dev(C.c_void_p(states.data_ptr()), states.numel() // 25, _stream_ptr()))
        return states
    a = np.ascontiguousarray(states, dtype=np.uint64)
    assert a.shape[-1] == 25
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    _lib.check(L.p3hip_keccak_f_dev(C.c_void_p(d.data_ptr()), a.size // 25, _stream_ptr()))
    return d.cpu().numpy().view(np.uint64).reshape(a.shape)


class MerkleTree:
    """Prover data of one commitment: device digest layers + the committed matrices (kept alive)."""

    def __init__(self, handle, mats, root):
        self._h = handle
        self.mats = mats
        self.root = root
        self.log_max_height = _lib.lib().p3hip_mmcs_log_max_height(handle)

    def digest_layers(self):
        """All digest layers as numpy arrays (len, 8) — test/inspection helper (downloads)."""
        L = _lib.lib()
        out = []
        for l in range(L.p3hip_mmcs_num_layers(self._h)):
            n = C.c_size_t()
            p = L.p3hip_mmcs_layer_dev(self._h, l, C.byref(n))
            a = np.zeros((n.value, 8), dtype=np.uint32)
            _lib.check(L.p3hip_download(a.ctypes.data_as(C.c_void_p), C.c_void_p(p), a.nbytes))
            out.append(a)
        return out

    def free(self):
        if self._h:
            _lib.lib().p3hip_mmcs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MerkleTreeMmcs:
    def __init__(self, hash="poseidon2"):
        kinds = {"poseidon2": HASH_POSEIDON2, "keccak": HASH_KECCAK}
        if hash not in kinds:
            raise ValueError("unknown hash configuration %r" % (hash,))
        self.hash, self._kind = hash, kinds[hash]

    def commit(self, mats):
        """Mmcs::commit.  mats: list of 2-D matrices (torch device tensors stay resident; numpy arrays are
        uploaded).  Returns (root as numpy uint32[8], MerkleTree)."""
        import torch
        L = _lib.lib()
        dmats = [m.contiguous() if _is_torch(m) else dev_u32(m) for m in mats]
        n = len(dmats)
        ptrs = (C.c_void_p * n)(*[m.data_ptr() for m in dmats])
        hs = (C.c_size_t * n)(*[m.shape[0] for m in dmats])
        ws = (C.c_size_t * n)(*[m.shape[1] for m in dmats])
        root = np.zeros(8, dtype=np.uint32)
        handle = C.c_void_p()
        torch.cuda.current_stream()  # make sure a context exists
        _lib.check(L.p3hip_mmcs_commit_hash_dev(self._kind, ptrs, hs, ws, n, root.ctypes.data_as(C.c_void_p),
                                                C.byref(handle), _stream_ptr()))
        return root, MerkleTree(handle, dmats, root)

    def commit_matrix(self, mat):
        return self.commit([mat])

    def get_matrices(self, tree):
        return tree.mats

    def get_max_height(self, tree):
        return 1 << tree.log_max_height

    def open_batch(self, index, tree):
        """Mmcs::open_batch -> (list of opened rows per matrix, sibling path (log_max_height, 8))."""
        tot = sum(m.shape[1] for m in tree.mats)
        rows = np.zeros(max(tot, 1), dtype=np.uint32)
        path = np.zeros((max(tree.log_max_height, 1), 8), dtype=np.uint32)
        _lib.check(_lib.lib().p3hip_mmcs_open_batch(tree._h, index, rows.ctypes.data_as(C.c_void_p),
                                                    path.ctypes.data_as(C.c_void_p), _stream_ptr()))
        out, off = [], 0
        for m in tree.mats:
            out.append(rows[off:off + m.shape[1]].copy())
            off += m.shape[1]
        return out, path[: tree.log_max_height].copy()


class MerkleTreeHidingMmcs(MerkleTreeMmcs):
    """MerkleTreeHidingMmcs<.., SmallRng, .., SALT_ELEMS 4> (native/src/fib_air.rs:40-51): commit salts every matrix with
    draws from the MMCS's own rng (a DeviceRng: the stream lives in HBM); open_batch returns (values, (salts, siblings))."""
    SALT_ELEMS = 4

    def __init__(self, hash="keccak", rng=None, seed=1):
        super().__init__(hash)
        from .fib_air import DeviceRng
        self.rng = rng or DeviceRng(seed)

    def commit(self, mats):
        import torch
        L = _lib.lib()
        dmats = [m.contiguous() if _is_torch(m) else dev_u32(m) for m in mats]
        n = len(dmats)
        ptrs = (C.c_void_p * n)(*[m.data_ptr() for m in dmats])
        hs = (C.c_size_t * n)(*[m.shape[0] for m in dmats])
        ws = (C.c_size_t * n)(*[m.shape[1] for m in dmats])
        root = np.zeros(8, dtype=np.uint32)
        handle = C.c_void_p()
        torch.cuda.current_stream()
        _lib.check(L.p3hip_mmcs_commit_hiding_dev(self._kind, ptrs, hs, ws, n, self.rng._h, root.ctypes.data_as(C.c_void_p),
                                                  C.byref(handle), _stream_ptr()))
        return root, MerkleTree(handle, dmats, root)

    def open_batch(self, index, tree):
        n = len(tree.mats)
        tot = sum(m.shape[1] + self.SALT_ELEMS for m in tree.mats)
        rows = np.zeros(tot, dtype=np.uint32)
        path = np.zeros((max(tree.log_max_height, 1), 8), dtype=np.uint32)
        _lib.check(_lib.lib().p3hip_mmcs_open_batch(tree._h, index, rows.ctypes.data_as(C.c_void_p),
                                                    path.ctypes.data_as(C.c_void_p), _stream_ptr()))
        vals, salts, off = [], [], 0
        for m in tree.mats:
            w = m.shape[1]
            vals.append(rows[off:off + w].copy())
            salts.append(rows[off + w:off + w + self.SALT_ELEMS].copy())
            off += w + self.SALT_ELEMS
        assert len(vals) == n
        return vals, (salts, path[: tree.log_max_height].copy())
