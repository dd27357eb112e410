#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Runs ONLY in the build container
(needs `tokenizers` 0.22.2, whose extension statically links the genuine
aho-corasick crate 1.1.4 -- the exact version the reference pins in
Cargo.lock:6-7).  Nothing here is needed on the GPU box; the JSON files are.

  reference_vectors.json  known-answer vectors of the reference's own tests
                          and README (inputs + expected outputs, transcribed
                          as data; citations per entry)
  ll_crate.json           LeftmostLongest results produced by the genuine
                          crate (via tokenizers added-token matching), small
                          cases stored in full
  ll_crate_large.json     same at 10k-pattern scale: seeds + SHA-256 + count
  lf_re.json              LeftmostFirst results produced by Python `re`
                          alternation (independent engine)
  kinds_large.json        10k-pattern sets WITH duplicates and nested patterns over
                          256-512 KB haystacks, every search mode: Standard and
                          overlapping from the brute-force spec (tests/spec.py),
                          LeftmostFirst from `re` alternation (and the spec, which must
                          agree), LeftmostLongest from the spec; seeds + SHA-256 of the
                          canonical (pattern,start,end) u64 stream + count + head

usage: python tests/golden/make_golden.py
"""
import json
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen  # noqa: E402


def dump(name, obj):
    with open(os.path.join(HERE, name), "w", encoding="utf-8") as f:
        json.dump(obj, f, ensure_ascii=False, indent=None, separators=(",", ":"))
        f.write("\n")
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


# ---------------------------------------------------------------- reference
WINTER_H = "This is the winter of my discontent"
WINTER_P = ["content", "disco", "disc", "discontent", "winter"]

REFERENCE = [
    # tests/test_ac.py:39-56, tests/test_ac_bytes.py:35-44,62-71,91-100
    {"cite": "tests/test_ac.py:39-56", "patterns": ["hello", "world"],
     "haystack": "hello, world, hello again", "kind": 0, "overlapping": False,
     "strings": ["hello", "world", "hello"],
     "indexes": [[0, 0, 5], [1, 7, 12], [0, 14, 19]]},
    # README.md:37-49
    {"cite": "README.md:37-49", "patterns": ["hello", "world", "fish"],
     "haystack": "this is my first hello world. hello!", "kind": 0,
     "overlapping": False, "strings": ["hello", "world", "hello"],
     "indexes": [[0, 17, 22], [1, 23, 28], [0, 30, 35]]},
    # README.md:70-79 (bytes)
    {"cite": "README.md:70-79", "patterns": ["hello", "world"],
     "haystack": "hello world", "kind": 0, "overlapping": False,
     "strings": ["hello", "world"], "indexes": [[0, 0, 5], [1, 6, 11]]},
    # tests/test_ac.py:120-132 (code-point indexes over 2/3/4-byte chars)
    {"cite": "tests/test_ac.py:120-132", "patterns": ["d ☃f", "há", "l🤦l"],
     "haystack": "hello, world ☃fishá l🤦l", "kind": 0, "overlapping": False,
     "strings": ["d ☃f", "há", "l🤦l"],
     "indexes": [[0, 11, 15], [1, 17, 19], [2, 20, 23]]},
    # tests/test_ac.py:208-248 / tests/test_ac_bytes.py:204-252
    {"cite": "tests/test_ac.py:218-229", "patterns": WINTER_P, "haystack": WINTER_H,
     "kind": 0, "overlapping": False, "strings": ["winter", "disc"]},
    {"cite": "tests/test_ac.py:232-238", "patterns": WINTER_P, "haystack": WINTER_H,
     "kind": 1, "overlapping": False, "strings": ["winter", "disco"]},
    {"cite": "tests/test_ac.py:241-248", "patterns": WINTER_P, "haystack": WINTER_H,
     "kind": 2, "overlapping": False, "strings": ["winter", "discontent"]},
    # tests/test_ac.py:255-292 / tests/test_ac_bytes.py:259-295
    {"cite": "tests/test_ac.py:277-286", "patterns": WINTER_P, "haystack": WINTER_H,
     "kind": 0, "overlapping": True,
     "strings": ["winter", "disc", "disco", "discontent", "content"]},
    {"cite": "tests/test_ac.py:271-275,291", "patterns": WINTER_P, "haystack": WINTER_H,
     "kind": 1, "overlapping": True, "error": "ValueError"},
    {"cite": "tests/test_ac.py:271-275,292", "patterns": WINTER_P, "haystack": WINTER_H,
     "kind": 2, "overlapping": True, "error": "ValueError"},
    # README.md:100-119
    {"cite": "README.md:102-105", "patterns": ["disco", "disc", "discontent"],
     "haystack": "discontent", "kind": 0, "overlapping": False, "strings": ["disc"]},
    {"cite": "README.md:106-108", "patterns": ["b", "abcd"], "haystack": "abcdef",
     "kind": 0, "overlapping": False, "strings": ["b"]},
    # README.md:125-139
    {"cite": "README.md:127-129", "patterns": ["disco", "disc"],
     "haystack": "discontent", "kind": 1, "overlapping": False, "strings": ["disco"]},
    {"cite": "README.md:130-131", "patterns": ["disc", "disco"],
     "haystack": "discontent", "kind": 1, "overlapping": False, "strings": ["disc"]},
    {"cite": "README.md:137-139", "patterns": ["b", "abcd"], "haystack": "abcdef",
     "kind": 1, "overlapping": False, "strings": ["abcd"]},
    # README.md:144-148
    {"cite": "README.md:146-148", "patterns": ["disco", "disc", "discontent"],
     "haystack": "discontent", "kind": 2, "overlapping": False,
     "strings": ["discontent"]},
    # README.md:155-161
    {"cite": "README.md:157-161", "patterns": ["winter", "onte", "disco", "discontent"],
     "haystack": "discontent", "kind": 0, "overlapping": True,
     "strings": ["disco", "onte", "discontent"]},
]


# ---------------------------------------------------------------- crate (LL)
def crate_ll(patterns, haystack):
    """The genuine aho-corasick 1.1.4 LeftmostLongest find_iter, reached
    through tokenizers' added-token matcher."""
    from tokenizers import AddedToken, Tokenizer, models
    tok = Tokenizer(models.WordLevel({"[UNK]": 0}, unk_token="[UNK]"))
    n = tok.add_tokens([AddedToken(p, single_word=False, lstrip=False, rstrip=False,
                                   normalized=False) for p in patterns])
    assert n == len(patterns), "patterns must be unique"
    enc = tok.encode(haystack, add_special_tokens=False)
    return [[tid - 1, s, e] for tid, (s, e) in zip(enc.ids, enc.offsets) if tid >= 1]


def uniq(seq):
    seen, out = set(), []
    for x in seq:
        if x not in seen:
            seen.add(x)
            out.append(x)
    return out


def make_ll_small():
    rng = random.Random(20260926)
    alphabets = ["ab", "abc", "abcdefgh", "abé☃", "ab🤦c é"]
    cases = []
    for i in range(240):
        alpha = alphabets[i % len(alphabets)]
        npat = rng.choice([1, 2, 3, 5, 8, 20, 60, 150])  # <=100 -> crate DFA, >100 -> contiguous NFA
        pats = uniq("".join(rng.choice(alpha) for _ in range(rng.randint(1, 7)))
                    for _ in range(npat))
        pats = [p for p in pats if p.strip() == p and p]  # tokenizers trims nothing, but keep it simple
        if not pats:
            continue
        hay = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 120)))
        cases.append({"patterns": pats, "haystack": hay, "expected": crate_ll(pats, hay)})
    return cases


def make_ll_large():
    out = []
    for (npat, nchars, pseed, hseed, plant, lo) in [(10000, 200_000, 5, 55, 512, 5), (2000, 100_000, 6, 66, 512, 5),
                                                    (150, 50_000, 7, 77, 512, 5),
                                                    # denser: a planted pattern every 48 characters, short
                                                    # patterns (2-9 characters: nested occurrences)
                                                    (10000, 300_000, 8, 88, 48, 2)]:
        pats = uniq(gen.gen_patterns(npat, lo, lo + 7, gen.AZ_UNI, pseed))
        hay = gen.gen_unicode_textlike(nchars, hseed, pats, plant_every=plant)
        exp = crate_ll(pats, hay)
        out.append({"n_patterns_requested": npat, "n_unique": len(pats), "lo": lo, "hi": lo + 7,
                    "alphabet": "AZ_UNI", "pattern_seed": pseed, "nchars": nchars,
                    "haystack_seed": hseed, "plant_every": plant, "count": len(exp),
                    "sha256": gen.canonical_sha256(exp), "head": exp[:16]})
        print("ll large", npat, "->", len(exp), "matches")
    return out


# ---------------------------------------------------------------- re (LF)
def re_lf(patterns, haystack):
    rx = re.compile("|".join(re.escape(p) for p in patterns))
    first = {}
    for i, p in enumerate(patterns):
        first.setdefault(p, i)
    return [[first[m.group(0)], m.start(), m.end()] for m in rx.finditer(haystack)]


def make_lf():
    rng = random.Random(977)
    alphabets = ["ab", "abc", "abcdefgh", "abé☃"]
    cases = []
    for i in range(240):
        alpha = alphabets[i % len(alphabets)]
        npat = rng.choice([1, 2, 3, 5, 8, 20, 60, 150])
        pats = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 7)))
                for _ in range(npat)]  # duplicates allowed here
        hay = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 120)))
        cases.append({"patterns": pats, "haystack": hay, "expected": re_lf(pats, hay)})
    return cases


# ---------------------------------------------------------------- large, all kinds
LARGE_CASES = [
    # (generator, n_patterns, pattern seed, haystack bytes, haystack seed, plant_every)
    ("nested", 10000, 21, 256 * 1024, 31, 128),   # dense: 2-byte pieces, nested, ~9 % duplicates
    ("names", 10000, 22, 512 * 1024, 32, 256),    # sparse: 5-12 letters, ~5 % duplicates
    ("names", 4244, 6, 1_000_000, 33, 512),       # the cfg1 pattern set
]


def make_kinds_large():
    import spec
    out = []
    for (g, n, pseed, nbytes, hseed, plant) in LARGE_CASES:
        pats, hay = gen.large_case_inputs({"generator": g, "n_patterns": n, "pattern_seed": pseed,
                                           "haystack_bytes": nbytes, "haystack_seed": hseed,
                                           "plant_every": plant})
        res = {}
        for name, kind, ov in (("standard", "std", False), ("overlapping", "std", True),
                               ("leftmost_first", "lf", False), ("leftmost_longest", "ll", False)):
            m = [list(x) for x in spec.spec(pats, hay, kind, overlapping=ov)]
            if name == "leftmost_first":  # independent engine; must agree with the spec
                via_re = re_lf([p.decode() for p in pats], hay.decode())
                assert via_re == m, "re alternation and the spec disagree"
            res[name] = {"count": len(m), "sha256": gen.canonical_sha256(m), "head": m[:16]}
            print("kinds large", g, n, name, len(m))
        out.append({"generator": g, "n_patterns": n, "pattern_seed": pseed, "haystack_bytes": nbytes,
                    "haystack_seed": hseed, "plant_every": plant,
                    "n_duplicates": len(pats) - len(set(pats)), "results": res})
    return out


if __name__ == "__main__":
    dump("reference_vectors.json", REFERENCE)
    dump("ll_crate.json", make_ll_small())
    dump("ll_crate_large.json", make_ll_large())
    dump("lf_re.json", make_lf())
    dump("kinds_large.json", make_kinds_large())
