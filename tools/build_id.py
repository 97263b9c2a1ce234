#!/usr/bin/env python3
"""Build identity recorded in every PMC summary: {"src_sha256": hash of the library's sources, "lib_sha256": hash of the binary}.
   python3 tools/build_id.py  ->  one JSON object on stdout (bench_support._profile_json compares src_sha256 with the tree's)."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def build_id():
    import importlib.util
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("p3lib_id", os.path.join(root, "plonky3-mobile_amd", "_lib.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    lib = os.path.join(root, "plonky3-mobile_amd", "libp3hip.so")
    return {"src_sha256": m.src_sha256(), "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest()}


if __name__ == "__main__":
    print(json.dumps(build_id()))
