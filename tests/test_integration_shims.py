"""integration/ holds the Rust side of the drop-in (never compiled here: no Rust toolchain).  What CAN be checked
offline: every `extern "C"` function the shims declare exists in include/p3hip.h with the same number of parameters,
every file carries the "NEVER COMPILED HERE" banner, and the selector patch names the reference lines it changes."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")


def _header_functions():
    text = open(os.path.join(ROOT, "include", "p3hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    funcs = {}
    for m in re.finditer(r"\b(p3hip_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        funcs[m.group(1)] = n
    return funcs


def _rust_externs(path):
    text = open(path).read()
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", blk.group(1), flags=re.S):
            args = m.group(2).strip().rstrip(",")
            n = 0 if not args else len([a for a in args.split(",") if a.strip()])
            out[m.group(1)] = n
    return out


def test_rust_shims_bind_the_declared_c_abi():
    header = _header_functions()
    assert "p3hip_dft_batch_bb31" in header and header["p3hip_coset_lde_batch_bb31"] == 7
    seen = set()
    for dirpath, _, files in os.walk(INTEG):
        for f in files:
            if not f.endswith(".rs"):
                continue
            for name, n in _rust_externs(os.path.join(dirpath, f)).items():
                assert name in header, "%s: %s is not declared in include/p3hip.h" % (f, name)
                assert header[name] == n, "%s: %s takes %d parameters in include/p3hip.h, %d in the shim" % (f, name, header[name], n)
                seen.add(name)
    # the shims cover the boundary the reference needs: the DFT entry, the LDE, the MMCS and the diagnostics
    for must in ("p3hip_dft_batch_bb31", "p3hip_coset_lde_batch_bb31", "p3hip_mmcs_commit_hash", "p3hip_mmcs_open_batch",
                 "p3hip_mmcs_free", "p3hip_take_last_error", "p3hip_is_available"):
        assert must in seen, must


def test_shim_files_say_they_were_never_compiled():
    n = 0
    for dirpath, _, files in os.walk(INTEG):
        for f in files:
            if f.endswith((".rs", ".patch")):
                assert "NEVER COMPILED HERE" in open(os.path.join(dirpath, f)).read(), f
                n += 1
    assert n >= 5


def test_selector_patch_follows_the_reference_enum():
    patch = open(os.path.join(INTEG, "native", "src", "gpu_dft.rs.patch")).read()
    for needle in ("BackendKind::Hip => 4", '"hip" => BackendKind::Hip', "4 => BackendKind::Hip", "backend_hip::dft_batch",
                   "LAST_VULKAN_ERROR", "self.cpu.dft_batch(mat).to_row_major_matrix()"):
        assert needle in patch, needle
    hdr = open(os.path.join(ROOT, "include", "p3hip.h")).read()
    assert "#define P3HIP_BACKEND_HIP 4" in hdr
