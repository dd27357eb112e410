"""GPU: mid-size host haystacks read IN PLACE (round 6, acx_api.cpp acx_find): beyond K0's sizes and up to 1 MiB the calling
thread copies the haystack into pinned host memory and the scan reads it there -- no staging copy of the runtime, no DMA the
scan's launch waits for.  Every kind, bytes and code points, against the oracle; path_stats["in_place"] says the call went
that way.  Reference path: /root/reference/src/lib.rs:422-434 (bytes), 229-249 (str)."""
import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


@pytest.mark.parametrize("mk", [0, 1, 2])
def test_in_place_every_kind_sizes_and_code_points(mk):
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1) + [b"caf\xc3\xa9", b"\xf0\x9f\xa4\xa6x"]
    a = capi.Automaton(pats, mk)
    o = Oracle(pats, mk, KIND_DFA)
    for n in (65537, 100_003, 262_144, 700_001, 1 << 20, (1 << 20) + 1, 3 << 20):
        hay = gen.gen_textlike(n, 17 + n, pats[:3000]).copy()
        hay[5:10] = np.frombuffer(b"caf\xc3\xa9", dtype=np.uint8)
        hay[n - 5:n] = np.frombuffer(b"\xf0\x9f\xa4\xa6x", dtype=np.uint8)  # a match that ends with the haystack
        hay = hay.tobytes()
        b2c = byte_to_code_point(hay)
        for ov in ([False, True] if mk == 0 else [False]):
            want = o.find_raw(hay, overlapping=ov)
            a.path_stats(reset=True)
            got = cols(a.find(hay, overlapping=ov))
            got_cp = cols(a.find(hay, overlapping=ov, codepoints=True))
            st = a.path_stats()
            assert got.shape == want.shape and np.array_equal(got, want), (mk, n, ov)
            assert np.array_equal(got_cp[:, 0], want[:, 0]) and np.array_equal(got_cp[:, 1], b2c[want[:, 1]]) and \
                np.array_equal(got_cp[:, 2], b2c[want[:, 2]]), (mk, n, ov)
            assert st["in_place"] == (2 if n <= (1 << 20) else 0), (n, st)
            assert st["k0"] == 0
    a.close()


def test_in_place_dense_input_and_back():
    """A dense mid-size input read in place (hot groups / the dense path read the pinned bytes too: slower, the same answer);
    while the context expects dense inputs its calls are staged in HBM as before; sparse inputs go back to in place."""
    pats = [b"ab", b"b", b"abc", b"zzzzzz"]
    a = capi.Automaton(pats, 0)
    o = Oracle(pats, 0, KIND_DFA)
    dense = (b"ab" * 150_000)
    sparse = gen.gen_uniform(300_000, b"qrstuvwxy", 5).tobytes()
    half = sparse[:100_000] + b"abcab" * 20_000 + sparse[:100_000]
    for hay in (sparse, dense, dense, half, sparse, sparse, sparse, half, sparse):
        for ov in (False, True):
            want = o.find_raw(hay, overlapping=ov)
            got = cols(a.find(hay, overlapping=ov))
            assert got.shape == want.shape and np.array_equal(got, want), (len(hay), ov)
    a.path_stats(reset=True)
    for _ in range(12):
        a.find(sparse)
    assert a.path_stats()["in_place"] >= 1
    a.close()


def test_in_place_str_api_and_threads():
    import threading
    import ahocorasick_rs_amd as ac
    names = gen.names_like(2000, 6)
    a = ac.AhoCorasick(names, matchkind=ac.MatchKind.LeftmostLongest)
    o = Oracle([p.encode() for p in names], 2, KIND_DFA)
    texts = [gen.names_haystack(names, n, every=5).decode() for n in (70_000, 200_000, 1_000_000)]
    want = [[(int(p), int(s), int(e)) for p, s, e in o.find_raw(t.encode())] for t in texts]  # (ASCII: byte offsets are code points)
    bad = []

    def worker(t):
        for rep in range(6):
            i = (t + rep) % len(texts)
            if a.find_matches_as_indexes(texts[i]) != want[i]:
                bad.append((t, i))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad, bad[:3]
