"""Executable brute-force statement of the reference's observable semantics
(SURVEY.md §8a / Appendix B).  O(patterns x haystack): small cases only.

Works on `bytes` (byte offsets) or `str` (code-point offsets, which is what
`AhoCorasick.find_matches_as_indexes` returns, /root/reference/src/lib.rs:240-246).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

Match = Tuple[int, int, int]

STD, LF, LL = "std", "lf", "ll"
KIND_NAMES = {0: STD, 1: LF, 2: LL}


def occurrences(pats: Sequence, h) -> List[Match]:
    out = []
    for pid, p in enumerate(pats):
        i = h.find(p)
        while i != -1:
            out.append((pid, i, i + len(p)))
            i = h.find(p, i + 1)
    return out


_KEY = {
    STD: lambda m: (m[2], m[1], m[0]),
    LF: lambda m: (m[1], m[0]),
    LL: lambda m: (m[1], -(m[2] - m[1]), m[0]),
}


def spec(pats: Sequence, h, kind: str = STD, overlapping: bool = False) -> List[Match]:
    if isinstance(kind, int):
        kind = KIND_NAMES[kind]
    o = occurrences(pats, h)
    if overlapping:
        if kind != STD:
            raise ValueError("overlapping requires Standard")
        return sorted(o, key=_KEY[STD])
    o.sort(key=_KEY[kind])
    pos, out = 0, []
    for m in o:  # first in key order with start >= pos is the argmin
        if m[1] >= pos:
            out.append(m)
            pos = m[2]
    return out
