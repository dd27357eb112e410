"""GPU: the BASELINE.json configurations at their FULL size (1 GiB; 8 GiB for the north-star target
size), checked through size-independent properties -- the oracle would need minutes per case:

  * every reported slice equals its pattern (sampled densely), offsets are in range;
  * non-overlapping results are sorted and do not overlap; overlapping results are sorted by
    (end, start) and are exactly the set of occurrences the non-overlapping result is drawn from;
  * the count equals the oracle's count on a bounded prefix / on sampled windows, and the match
    stream restricted to a window equals the oracle's stream of that window when the window is
    cut at a position no match crosses;
  * the host-memory entry point (acx_find on 1 GiB of host bytes, pipelined staging, pinned
    result) returns exactly the device-resident result.
"""
import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
GIB = 1 << 30


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def check_slices(host, pats, got, step):
    for (p, s, e) in got[::max(1, len(got) // step)]:
        assert host[int(s):int(e)].tobytes() == pats[int(p)]


def check_window_against_oracle(host, got, o, lo, hi, max_len, overlapping):
    """The matches that lie inside [lo, hi) equal the oracle's on host[lo:hi], provided no reported
    match crosses lo (then the reference iterator, restarted at lo, reports the same stream)."""
    inside = got[(got[:, 1] >= lo) & (got[:, 2] <= hi)]
    crossing = got[(got[:, 1] < lo) & (got[:, 2] > lo)]
    if len(crossing):
        return False
    want = o.find_raw(host[lo:hi], overlapping=overlapping)
    want = want[want[:, 2] <= hi - lo - max_len] if len(want) else want  # the window's tail may cut matches short
    inside = inside[inside[:, 2] <= hi - max_len]
    shifted = want.copy()
    shifted[:, 1:] += lo
    assert np.array_equal(inside, shifted), (lo, hi)
    return True


@pytest.mark.parametrize("alphabet", ["az", "bytes"])
def test_cfg4_100k_patterns_overlapping_1gib(alphabet):
    alpha = gen.AZ if alphabet == "az" else gen.ALL_BYTES
    pats = gen.gen_patterns(100000, 5, 12, alpha, 3 if alphabet == "az" else 4)
    a = capi.Automaton(pats, 0, capi.IMPL_AUTO)
    buf = capi.DeviceBuffer(GIB)
    if alphabet == "az":
        a.generate(buf.ptr, GIB, 0, 12)
        host = buf.download()
    else:
        host = gen.gen_uniform(GIB, alpha, 12)
        for k in range(0, GIB - 64, 1 << 20):  # uniform bytes almost never match: plant some patterns
            p = np.frombuffer(pats[(k >> 20) % len(pats)], dtype=np.uint8)
            host[k:k + len(p)] = p
        buf.upload(host)
    o = Oracle(pats, 0, KIND_DFA)
    for ov in (True, False):
        r = a.find_device(buf.ptr, GIB, overlapping=ov)
        got = cols(r.matches())
        r.free()
        assert len(got) > 1000
        assert got[:, 2].max() <= GIB and np.all(got[:, 2] > got[:, 1])
        assert np.all(got[1:, 2] >= got[:-1, 2])  # sorted by end
        if ov:
            same_end = got[1:, 2] == got[:-1, 2]
            assert np.all(got[1:, 1][same_end] >= got[:-1, 1][same_end])  # longest first at an equal end
        else:
            assert np.all(got[1:, 1] >= got[:-1, 2])  # non-overlapping
        check_slices(host, pats, got, 3000)
        checked = 0
        for lo in (0, 123_456_789, 777_000_000, GIB - (1 << 22)):
            checked += check_window_against_oracle(host, got, o, lo, min(GIB, lo + (1 << 22)), 12, ov)
        assert checked >= 2
    a.close()


def test_cfg5_utf8_leftmost_longest_codepoints_1gib():
    spats = list(dict.fromkeys(gen.gen_patterns(10000, 5, 12, gen.AZ_UNI, 5)))
    pats = [p.encode() for p in spats]
    host = gen.gen_unicode_textlike_bytes(int(GIB / 1.12), 56, spats, threads=16)
    n = len(host)
    a = capi.Automaton(pats, 2, capi.IMPL_AUTO)
    buf = capi.DeviceBuffer(n).upload(host)
    r = a.find_device(buf.ptr, n)
    got_b = cols(r.matches())
    r.free()
    r = a.find_device(buf.ptr, n, codepoints=True)
    got_c = cols(r.matches())
    r.free()
    assert len(got_b) == len(got_c) > 10 ** 6
    assert np.all(got_b[1:, 1] >= got_b[:-1, 2])
    check_slices(host, pats, got_b, 3000)
    # code-point indexes: the number of non-continuation bytes in front of the byte offset
    is_lead = (host & 0xC0) != 0x80
    sample = np.linspace(0, len(got_b) - 1, 4000).astype(np.int64)
    pre = np.concatenate([[0], np.cumsum(is_lead, dtype=np.int64)])
    assert np.array_equal(got_c[sample, 0], got_b[sample, 0])
    assert np.array_equal(got_c[sample, 1], pre[got_b[sample, 1].astype(np.int64)].astype(np.uint64))
    assert np.array_equal(got_c[sample, 2], pre[got_b[sample, 2].astype(np.int64)].astype(np.uint64))
    o = Oracle(pats, 2, KIND_DFA)
    checked = 0
    for lo in (0, 400_000_123, n - (1 << 22)):
        while lo and (host[lo] & 0xC0) == 0x80:
            lo += 1
        checked += check_window_against_oracle(host, got_b, o, lo, min(n, lo + (1 << 22)), 48, False)
    assert checked >= 2
    a.close()


def test_8gib_offsets_beyond_2_pow_32():
    """north_star's target size on one GPU: 8 GiB, offsets >= 2^32, > 2 M tiles."""
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    n = 8 * GIB
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    buf = capi.DeviceBuffer(n)
    a.generate(buf.ptr, n, 1, 11)
    r = a.find_device(buf.ptr, n)
    got = cols(r.matches())
    r.free()
    assert len(got) > 8 * 10 ** 6 and int(got[-1, 2]) > (1 << 32) + (1 << 31)
    assert np.all(got[1:, 1] >= got[:-1, 2]) and got[:, 2].max() <= n
    o = Oracle(pats, 0, KIND_DFA)
    # the generator is position-addressed: any window can be re-made on the host and checked
    for lo in (0, (1 << 32) - (1 << 21), 5 * GIB + 12345 * 1024, n - (1 << 22)):  # multiples of 1 KiB
        tmp = capi.DeviceBuffer(1 << 22)
        a.generate(tmp.ptr, 1 << 22, 1, 11, stream_offset=lo)
        host = tmp.download()
        tmp.free()
        inside = got[(got[:, 1] >= lo) & (got[:, 2] <= lo + (1 << 22) - 12)]
        crossing = got[(got[:, 1] < lo) & (got[:, 2] > lo)]
        if len(crossing):
            continue
        want = o.find_raw(host)
        want = want[want[:, 2] <= (1 << 22) - 12]
        want[:, 1:] += lo
        assert np.array_equal(inside, want), lo
        for (p, s, e) in inside[::97]:
            assert host[int(s) - lo:int(e) - lo].tobytes() == pats[int(p)]
    a.close()


def test_host_memory_entry_point_equals_device_resident_1gib():
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    buf = capi.DeviceBuffer(GIB)
    a.generate(buf.ptr, GIB, 1, 11)
    r = a.find_device(buf.ptr, GIB)
    want = cols(r.matches())
    r.free()
    host = buf.download()
    buf.free()
    for _ in range(2):  # the second call reuses the staging ring and the pinned result pool
        got = a.find(host)
        assert np.array_equal(cols(got), want)
        del got
    got = a.find(host[5:GIB // 3])  # an unaligned, shorter view
    sub = want[(want[:, 1] >= 5) & (want[:, 2] <= GIB // 3)]
    crossing = want[(want[:, 1] < 5) & (want[:, 2] > 5)]
    if not len(crossing):
        sub = sub.copy()
        sub[:, 1:] -= 5
        assert np.array_equal(cols(got)[: len(sub)], sub)
    a.close()
