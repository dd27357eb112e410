"""GPU: the two saturated-table variants of K1b on sets that do not need them (ACX_FILTER_BIG forces the choice; read once per
process, hence the subprocess): 1 -- every position put to both level-1 tests, the survivors' windows gathered (10^5
patterns: BASELINE config 4); 2 -- ... their windows captured from the row staged in LDS, row-by-row compaction (10^6
patterns, `bench.py --config large`).  All kinds + overlapping, a set with 1- and 2-byte patterns (the side test's
survivors go through the same stage), code points, a dense stretch (mid-row batches), a batch -- against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("ahocorasick_rs_amd.capi")
HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point
from ahocorasick_rs_amd import capi

def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)

pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
hay = gen.gen_textlike(3 << 20, 11, pats).copy()
rng = gen.SplitMix64(5)
for k in range(1 << 20, (1 << 20) + (96 << 10), 32):  # a dense stretch: more survivors in a row than a batch holds
    p = np.frombuffer(pats[rng.next() %% len(pats)], dtype=np.uint8)
    hay[k:k + len(p)] = p
hay = hay.tobytes()
for mk in (0, 1, 2):
    a = capi.Automaton(pats, mk, capi.IMPL_DFA)
    o = Oracle(pats, mk, KIND_DFA)
    for ov in ([False, True] if mk == 0 else [False]):
        for view in (hay, hay[3:]):
            assert np.array_equal(cols(a.find(view, overlapping=ov)), o.find_raw(view, overlapping=ov)), (mk, ov)
    a.close()
# short patterns beside the long ones (the side test), > 32 byte classes
mixed = pats[:3000] + [b"qz", b"~", b"ab"]
a = capi.Automaton(mixed, 0)
assert np.array_equal(cols(a.find(hay)), Oracle(mixed, 0, KIND_DFA).find_raw(hay))
a.close()
# code points (the scan counts lead bytes on its way) + anchors
spats = list(dict.fromkeys(gen.gen_patterns(3000, 5, 12, gen.AZ_UNI, 5)))
bpats = [p.encode() for p in spats]
u = gen.gen_unicode_textlike_bytes(1 << 20, 56, spats).tobytes()
b2c = byte_to_code_point(u)
for mk in (0, 2):
    a = capi.Automaton(bpats, mk)
    want = Oracle(bpats, mk, KIND_DFA).find_raw(u)
    got = cols(a.find(u, codepoints=True))
    assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 1], b2c[want[:, 1]]) and np.array_equal(got[:, 2], b2c[want[:, 2]])
    a.close()
# a batch
a = capi.Automaton(pats, 0, capi.IMPL_DFA)
o = Oracle(pats, 0, KIND_DFA)
hays = [gen.gen_textlike(50_000 + 13 * i, 200 + i, pats).tobytes() for i in range(30)]
m, counts = a.find_batch(hays)
at = 0
for i, h in enumerate(hays):
    want = o.find_raw(h)
    assert counts[i] == len(want) and np.array_equal(cols(m[at:at + len(want)]), want), i
    at += len(want)
a.close()
print("OK")
""" % (os.path.dirname(HERE), HERE)


@pytest.mark.parametrize("variant", ["1", "2"])
def test_saturated_table_variants_of_the_scan(variant):
    r = subprocess.run([sys.executable, "-c", SCRIPT], env={**os.environ, "ACX_FILTER_BIG": variant}, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
