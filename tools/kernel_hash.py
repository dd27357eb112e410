#!/usr/bin/env python3
"""SHA-256 of the sources the device code is compiled from (kernels.hip and the headers it includes).
A measurement under profiles/ that names this hash belongs to THIS device code; bench.py reports a committed
PMC traffic figure only while the hash still matches (tools/collect_profiles.sh records it when it measures,
`tools/collect_profiles.sh --check <round>` fails when the round's profiles are of older kernels).
usage: kernel_hash.py            print the hash of the working tree's sources"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["kernels.hip", "kernels.hpp", "device_types.hpp", "automaton.hpp"]


def kernel_source_sha256(root: str = ROOT) -> str:
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(root, "ahocorasick_rs_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_sha256())
