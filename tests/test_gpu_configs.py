"""GPU parity at the other BASELINE.json configurations (bounded sizes so that
the oracle finishes in seconds): cfg3 batch shape, cfg4 100k patterns +
overlapping, cfg5 UTF-8 str haystack with code-point indexes."""
import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def test_cfg3_batch_of_8k_haystacks_device_resident():
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    n_hay, L = 4096, 8192  # 32 MiB of the cfg3 stream
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    buf = capi.DeviceBuffer(n_hay * L)
    a.generate(buf.ptr, n_hay * L, 1, 13)
    host = buf.download()
    r = a.find_device(buf.ptr, n_hay * L, n_hay=n_hay, uniform_len=L)
    m, counts = r.matches(), r.counts()
    r.free()
    o = Oracle(pats, 0, KIND_DFA)
    want = [o.find_raw(host[i * L:(i + 1) * L]) for i in range(n_hay)]
    assert counts.tolist() == [len(w) for w in want]
    assert np.array_equal(cols(m), np.concatenate(want))
    a.close()


@pytest.mark.parametrize("alphabet", ["az", "bytes"])
def test_cfg4_100k_patterns_overlapping(alphabet):
    alpha = gen.AZ if alphabet == "az" else gen.ALL_BYTES
    pats = gen.gen_patterns(100000, 5, 12, alpha, 3 if alphabet == "az" else 4)
    hay = gen.gen_uniform(1 << 24, alpha, 12)
    if alphabet == "bytes":  # uniform bytes almost never match: plant some patterns
        for k in range(0, len(hay) - 64, 4099):
            p = np.frombuffer(pats[k % len(pats)], dtype=np.uint8)
            hay[k:k + len(p)] = p
    a = capi.Automaton(pats, 0, capi.IMPL_AUTO)
    i = a.info
    assert i.n_states > 500000
    o = Oracle(pats, 0, KIND_DFA)
    for ov in (True, False):
        got = cols(a.find(hay, overlapping=ov))
        want = o.find_raw(hay, overlapping=ov)
        assert len(got) == len(want) and np.array_equal(got, want)
    a.close()


def test_cfg5_utf8_leftmost_longest_codepoints():
    import ahocorasick_rs_amd as ac
    pats = list(dict.fromkeys(gen.gen_patterns(10000, 5, 12, gen.AZ_UNI, 5)))
    hay = gen.gen_unicode_textlike(400_000, 56, pats)
    a = ac.AhoCorasick(pats, matchkind=ac.MatchKind.LeftmostLongest)
    got = a.find_matches_as_indexes(hay)
    want = Oracle([p.encode() for p in pats], 2, KIND_DFA).find_str(hay)
    assert got == want
    assert [hay[s:e] for (_, s, e) in got[:50]] == [pats[i] for (i, _, _) in got[:50]]
    assert a.find_matches_as_strings(hay) == [pats[i] for (i, _, _) in got]
