#!/usr/bin/env python3
"""After an edit that changes the sources' hash but NOT the compiled library (a comment, the header's prose): rewrite the `src_sha256` of the
committed round-5 PMC summaries — but only of those whose recorded `lib_sha256` equals the hash of the library this tree builds, i.e. whose
counts provably belong to the same machine code.  Anything else stays as it is (and reads STALE in the bench line until it is re-measured:
tools/collect_profiles_r05.sh).     python3 tools/refresh_build_id.py"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_id import build_id  # noqa: E402

bid = build_id()
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for f in sorted(glob.glob(os.path.join(root, "profiles", "r05_pmc_*.json"))):
    d = json.load(open(f))
    if d.get("lib_sha256") != bid["lib_sha256"]:
        print("%s: measured on another binary (%s...), left alone" % (os.path.basename(f), str(d.get("lib_sha256"))[:12]))
        continue
    if d.get("src_sha256") != bid["src_sha256"]:
        d["src_sha256"] = bid["src_sha256"]
        json.dump(d, open(f, "w"), indent=1)
        print("%s: same binary, src_sha256 refreshed" % os.path.basename(f))
    else:
        print("%s: up to date" % os.path.basename(f))
