#!/usr/bin/env python3
"""One hash over the sources librtmi.so is built from (raytracing_amd/csrc: kernels, host runtime, headers, tables,
Makefile). tools/profile_bench.sh records it next to the counter passes it collects, tools/pmc_summary.py copies it
into profiles/pmc_summary.json, and bench.py compares it with the tree it runs from: counters taken from other
sources than the ones being timed are reported as stale.
usage: tools/source_hash.py            prints the hash"""
import hashlib
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "raytracing_amd" / "csrc"
SUFFIXES = (".hip", ".cpp", ".h", ".inc")


def source_hash(root=CSRC):
    h = hashlib.sha256()
    for f in sorted(p for p in Path(root).iterdir() if p.suffix in SUFFIXES or p.name == "Makefile"):
        h.update(f.name.encode() + b"\0")
        h.update(f.read_bytes())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
