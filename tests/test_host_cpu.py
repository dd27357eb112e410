"""CPU-only checks of the product's host side: the C ABI loads and exports
every declared symbol, the automaton compiler (C++) builds the right tables,
argument validation mirrors the reference, and nothing matches on the CPU."""
import os
import random
import re

import numpy as np
import pytest

import gen
from spec import occurrences

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
capi = pytest.importorskip("ahocorasick_rs_amd.capi")

ID_MASK, FLAG_OUT, FLAG_OWN, NONE = 0x3FFFFFFF, 0x80000000, 0x40000000, 0xFFFFFFFF


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "acx.h")).read()
    names = sorted(set(re.findall(r"\b(acx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    L = capi.lib()
    for n in names:
        assert hasattr(L, n), n
    assert L.acx_version() == capi.ABI_VERSION == int(re.search(r"#define ACX_VERSION (\d+)", hdr).group(1))
    # (the binding's view of acx_path_stats is as long as the header's array)
    assert len(capi.Automaton.PATH_STATS) == int(re.search(r"#define ACX_PATH_STATS (\d+)", hdr).group(1))


def test_result_line_check_is_the_same_function_on_both_sides():
    # the host accepts a polled 64-byte line (K0's result, the sparse path's totals) when its last word equals
    # seq ^ check(words 1 .. 6): ONE inline function in kernels.hpp compiled for the device and for the host.  Its value is
    # pinned here against a restatement, so that an edit on one side of the boundary cannot pass unnoticed.
    src = open(os.path.join(ROOT, "ahocorasick_rs_amd", "csrc", "kernels.hpp")).read()
    body = src[src.index("inline uint64_t k0_line_check"):]
    body = body[:body.index("\n}\n") + 3]
    for needle in ("0x9E3779B97F4A7C15ull", "0xFF51AFD7ED558CCDull", "(h << 6) + (h >> 2)", "h ^ (h >> 32)"):
        assert needle in body, needle
    M = (1 << 64) - 1

    def check(mid, h0):
        h = h0
        for v in mid:
            h ^= (v + 0x9E3779B97F4A7C15 + ((h << 6) & M) + (h >> 2)) & M
            h = (h * 0xFF51AFD7ED558CCD) & M
        return h ^ (h >> 32)

    h0 = int(re.search(r"uint64_t h = (0x[0-9A-Fa-f]+)ull", body).group(1), 16)
    a, b = check([1, 2, 3, 4, 5, 6], h0), check([1, 2, 3, 4, 5, 7], h0)
    assert a != b and check([0] * 6, h0) != 0  # (an all-zero line never validates itself against seq = 0)


def test_shard_range_twin_of_the_distributed_helper():
    """acx_shard_range (the single-process multi-device batch) cuts a batch exactly like
    distributed.shard_range (the multi-process form): contiguous, in order, sizes differ by at most one."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("acx_distributed", os.path.join(ROOT, "ahocorasick_rs_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for n in (0, 1, 2, 7, 8, 9, 1000, 131072, 1048577):
        for world in (1, 2, 3, 8):
            cuts = [capi.shard_range(n, r, world) for r in range(world)]
            assert cuts == [D.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))


def test_count_exchange_behind_the_c_abi_offsets_and_clean_failure():
    """Round 4: RCCL behind the C ABI (acx_comm_*).  The offset arithmetic is host code; without a HIP device
    the communicator entry points fail with a message instead of crashing (there is no CPU path)."""
    assert capi.output_offsets([3, 0, 5]) == [0, 3, 3, 8]
    assert capi.output_offsets([]) == [0]
    assert capi.output_offsets([7]) == [0, 7]
    rng = random.Random(1)
    for _ in range(50):
        c = [rng.randrange(1 << 40) for _ in range(rng.randrange(1, 9))]
        off = capi.output_offsets(c)
        assert off[0] == 0 and off[-1] == sum(c) and all(off[i + 1] - off[i] == c[i] for i in range(len(c)))
    if capi.device_count() == 0:
        with pytest.raises(capi.AcxError) as e:
            capi.Comm.init_all([0])
        assert "no HIP device" in str(e.value)
        with pytest.raises(capi.AcxError):
            capi.comm_unique_id()
    with pytest.raises((ValueError, capi.AcxError)):
        capi.Comm.init_all([])


def walk_all_occurrences(h, hay: bytes):
    """Test-side walker over the product's host tables (NOT a product path)."""
    out, s = [], 0
    for i, b in enumerate(hay):
        e = int(h.table[s, h.classes[b]])
        s = e & ID_MASK
        if e & FLAG_OUT:
            t = s
            while t != NONE:
                for k in range(h.own_off[t], h.own_off[t + 1]):
                    pid = int(h.own_pid[k])
                    out.append((pid, i + 1 - int(h.pattern_len[pid]), i + 1))
                t = int(h.dlink[t])
    return out


def test_compiler_tables_enumerate_every_occurrence():
    rng = random.Random(7)
    for it in range(150):
        alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][it % 4]
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 6)))
                for _ in range(rng.randint(1, 25))]
        hay = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 120)))
        h = capi.HostAutomaton(pats)
        got = walk_all_occurrences(h, hay)
        want = sorted(occurrences(pats, hay), key=lambda m: (m[2], m[1], m[0]))
        # per end position: longest first, duplicates in id order == sorted by (end, start, pid)
        assert got == want
        # structure: BFS ids are depth-monotone, flags consistent
        ls = h.level_start
        assert ls[0] == 0 and ls[-1] == h.n_states and all(ls[i] <= ls[i + 1] for i in range(len(ls) - 1))
        for s in range(h.n_states):
            own = h.own_off[s + 1] > h.own_off[s]
            incoming = h.table[(h.table & ID_MASK) == s]
            if len(incoming):
                assert bool(incoming[0] & FLAG_OWN) == own
                assert bool(incoming[0] & FLAG_OUT) == (own or h.dlink[s] != NONE)
        h.close()


def nfa_step(h, s: int, b: int):
    """Test-side twin of the device's nfa_step over the compressed form (trie edges + failure links)."""
    while True:
        c = 0
        for k in range(int(h.first_child[s]), int(h.first_child[s + 1])):
            if int(h.in_byte[k]) == b:
                c = k
                break
        if c:
            s = c
            break
        if s == 0:
            break
        s = int(h.fail[s])
    return s, int(h.state_flags[s])


def test_compressed_form_is_the_same_automaton(monkeypatch):
    """The compressed form (always built) steps exactly like the dense table, and with
    ACX_DENSE_LIMIT=0 the compiler keeps the compressed form only."""
    rng = random.Random(11)
    for it in range(40):
        alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][it % 4]
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 6))) for _ in range(rng.randint(1, 25))]
        h = capi.HostAutomaton(pats)
        assert h.dense and h.table.shape[0] == h.n_states
        assert h.first_child[0] == 1 or h.n_states == 1
        assert h.first_child[-1] == h.n_states
        for s in range(h.n_states):
            kids = h.in_byte[h.first_child[s]:h.first_child[s + 1]]
            assert all(kids[i] < kids[i + 1] for i in range(len(kids) - 1))
            for b in set(alpha) | {0, 255}:
                e = int(h.table[s, h.classes[b]])
                t, fl = nfa_step(h, s, b)
                assert (e & ID_MASK, e >> 30) == (t, fl), (pats, s, b)
        monkeypatch.setenv("ACX_DENSE_LIMIT", "0")
        c = capi.HostAutomaton(pats)
        monkeypatch.delenv("ACX_DENSE_LIMIT")
        assert not c.dense and c.table.size == 0 and c.n_states == h.n_states
        for name in ("first_child", "in_byte", "fail", "state_flags", "own_off", "own_pid", "dlink"):
            assert np.array_equal(getattr(c, name), getattr(h, name)), name


def test_leftmost_kinds_file_only_the_first_of_identical_patterns(monkeypatch):
    """Of several identical patterns only the first can be reported by a leftmost kind (lowest index
    wins): the later copies stay out of the own lists and of the prefix table's candidate lists;
    Standard keeps them all (an overlapping search reports every one)."""
    monkeypatch.setenv("ACX_NO_ANCHORS", "1")  # (every pattern under its beginning: the windows below start with the patterns)
    pats = [b"abcde", b"xyzzy", b"abcde", b"abcdefgh", b"abcde", b"xyzzy", b"q" * 9, b"abcdefgh"]
    for mk in (0, 1, 2):
        h = capi.HostAutomaton(pats, mk)
        filed = sorted(int(x) for s in range(h.n_states) for x in h.own_pid[h.own_off[s]:h.own_off[s + 1]])
        want = list(range(len(pats))) if mk == 0 else [0, 1, 3, 6]
        assert filed == want, (mk, filed)
        lg, q2 = int(h.t.prefix_table_log2), int(h.t.filter_q2)
        for pid, p_ in enumerate(pats):
            cands = prefix_candidates(h, lg, q2, (p_ + b"\0" * 16)[:16])
            assert (pid in cands) == (pid in want), (mk, pid, cands)
            assert all(c in want for c in cands)


def test_compiler_matches_survey_sizes():
    # 10k a-z patterns len 5-12 seed 1 -> 63 277 states, 28 classes, stride 32
    h = capi.HostAutomaton(gen.gen_patterns(10000, 5, 12, gen.AZ, 1))
    assert (h.n_states, h.t.n_classes, h.stride) == (63277, 28, 32)
    assert list(h.level_start[:4]) == [0, 1, 27, 703]
    assert (h.t.min_pattern_len, h.t.max_pattern_len, h.t.filter_q, h.t.filter_q2) == (5, 12, 5, 5)
    # tie-break rank: (len desc, pid asc) is a permutation
    order = np.argsort(h.rank)
    lens = h.pattern_len[order]
    assert np.all(lens[:-1] >= lens[1:])
    same = lens[:-1] == lens[1:]
    assert np.all(order[:-1][same] < order[1:][same])
    h.close()


def prefix_walk(h, lg, salt, window: bytes, allow_redirect: bool):
    """Python twin of the kernels' prefix_walk: one probe sequence from the home slot of
    window[:salt]; an entry matches when the window starts with its key (K bytes).  A home slot
    holding another key ends the search unless its MORE bit is set.  -> ("code", code),
    ("redirect", N) or None."""
    idx, home = capi.prefix_slot(window, salt, lg), True
    mine = 1 << (8 + ((capi.prefix_hash(window, salt) >> 11) & 15))
    while True:
        lo, hi, meta, code = (int(x) for x in h.prefix_table[idx])
        if meta == 0xFFFFFFFF:
            return None
        K, nxt = meta & 15, (meta >> 4) & 15
        if (hi << 32 | lo) == int.from_bytes(window[:K], "little"):
            if not nxt:
                return ("code", code)
            if allow_redirect:
                return ("redirect", nxt)
        if home and not (meta & mine):
            return None
        home = False
        idx = (idx + 1) & ((1 << lg) - 1)


def prefix_candidates(h, lg, q2, window: bytes, coded: bool = False):
    """Python twin of prefix_code + the candidate list: the pattern ids K1b hands on for a
    haystack window (>= 8 bytes, zero padded).  coded: the codes as they are (pattern id | anchor shift << 24)."""
    got = prefix_walk(h, lg, q2, window, True)
    if got and got[0] == "redirect":
        assert got[1] > q2
        got = prefix_walk(h, lg, got[1], window, False)
    if not got:
        return []
    code = got[1]
    codes = [code]
    if code & 0x80000000:
        i = code & 0x7FFFFFFF
        cnt = int(h.prefix_lists[i])
        codes = [int(x) for x in h.prefix_lists[i + 1:i + 1 + cnt]]
    return codes if coded else [c & 0xFFFFFF for c in codes]


def test_prefilter_tables_have_every_pattern_prefix(monkeypatch):
    monkeypatch.setenv("ACX_NO_ANCHORS", "1")  # (the tables of patterns filed under their beginnings; anchors: the next test)
    uni = [p.encode() for p in dict.fromkeys(gen.gen_patterns(600, 5, 12, gen.AZ_UNI, 5))]
    zeros = [b"abcde", b"abcde\0", b"abcde\0\0\0", b"abcdefgh", b"abcdefgi", b"abcdefghij", b"abcdf\0x", b"abcdf\0y",
             b"abcde", b"zzzzzzzzzzzz"]
    for pats in ([b"a", b"bc"], [b"abc", b"zzzz"], gen.gen_patterns(500, 4, 9, gen.ALL_BYTES, 3),
                 gen.gen_patterns(500, 6, 9, gen.AZ, 4), gen.gen_patterns(300, 9, 12, gen.AZ, 5),
                 [p.encode() for p in gen.names_like(800, 6)], uni, zeros):
        h = capi.HostAutomaton(pats)
        # (round 4: patterns of 1 and 2 bytes are kept out of these tables -- test_short_pattern_side_tables)
        long_ = [p for p in pats if len(p) >= 3]
        minlen = min((len(p) for p in long_), default=5)
        q, q2 = int(h.t.filter_q), int(h.t.filter_q2)
        assert (q, q2) == (min(5, minlen), min(8, minlen)) and int(h.t.long_min_len) == minlen
        assert int(h.t.n_short) == len(pats) - len(long_)
        g = q - 1
        lg = int(h.t.prefix_table_log2)
        all_pats, pats = pats, long_
        if not pats:
            assert not np.any(h.filter_xy) and int(h.t.n_prefix_keys) == 0
            h.close()
            continue
        ids = [i for i, p in enumerate(all_pats) if len(p) >= 3]
        for pid, p in zip(ids, pats):
            # level 1: in the row of p[1..1+g): bit p[0] of X; in the row of p[0..g): bit p[q-1] of Y;
            # in both rows the gate bit (gram & 31) of X
            for gram, byte, col in ((p[1:1 + g], p[0], 0), (p[0:g], p[q - 1], 1)):
                H = capi.filter_hash(gram)
                W = int.from_bytes(gram, "little")
                assert H == ((W & 0xFFFFFF) * 0x9E3779 + W) & 0xFFFFFFFF
                assert int(h.filter_xy[H >> 18, col]) >> (byte & 31) & 1
                assert int(h.filter_xy[H >> 18, 0]) >> (W & 31) & 1
            # level 2: a window that starts with the pattern resolves to a candidate list that holds
            # it -- and every candidate agrees with the window on min(8, its own length) bytes of
            # the group's key, in pattern-id order
            for tail in (b"\0" * 16, b"\xff" * 16, b"abcdefgh12345678"):
                cands = prefix_candidates(h, lg, q2, (p + tail)[:16])
                assert pid in cands, (p, cands)
                assert cands == sorted(cands)
                klen = min(min(len(all_pats[c]) for c in cands), 8)
                assert all(all_pats[c][:klen] == p[:klen] for c in cands)
        # the bitmap in front of the table: the first Q2 bytes of every pattern have their bit
        # (bit = the top log2 + 3 bits of the hash the home slot is taken from), and it is as
        # sparse as one bit per group in 8 bits per slot can be
        for p_ in pats:
            bit = capi.prefix_hash(p_[:q2], q2) >> (32 - lg - 3)
            assert int(h.prefix_bitmap[bit >> 5]) >> (bit & 31) & 1
        assert sum(bin(int(x)).count("1") for x in h.prefix_bitmap) <= len({p_[:q2] for p_ in pats})
        # a window that agrees with no pattern on its group's key resolves to nothing
        rng = random.Random(9)
        for _ in range(300):
            p = pats[rng.randrange(len(pats))]
            w = bytearray((p + bytes(rng.randrange(256) for _ in range(16)))[:16])
            w[rng.randrange(min(len(p), 8))] ^= 1 + rng.randrange(255)
            for c in prefix_candidates(h, lg, q2, bytes(w)):
                klen = min(len(all_pats[c]), 8)
                group = [o for o in pats if o[:q2] == all_pats[c][:q2]]
                assert bytes(w[:min(klen, min(min(len(o) for o in group), 8))]) == all_pats[c][:min(klen, min(min(len(o) for o in group), 8))]
        # the MORE filter (bits 23:8 of the meta word) of a home slot holds exactly the bits of the keys
        # that hash there and live elsewhere; redirect entries and single keys sit at the hash of
        # their first Q2 bytes, the keys behind a redirect at the hash of their own bytes
        tab = np.asarray(h.prefix_table)
        used = np.nonzero(tab[:, 2] != 0xFFFFFFFF)[0]
        assert len(used) == int(h.t.n_prefix_keys) <= 0.25 * (1 << lg) + 1
        keys_of = {}
        for p_ in pats:
            g_ = [o for o in pats if o[:q2] == p_[:q2]]
            keys_of.setdefault(p_[:q2], set()).add(p_[:min(8, min(len(o) for o in g_))])
        want_more = {}
        for e in used:
            K, nxt = int(tab[e, 2]) & 15, (int(tab[e, 2]) >> 4) & 15
            assert q2 <= K <= 8
            gram = (int(tab[e, 1]) << 32 | int(tab[e, 0])).to_bytes(8, "little")[:K]
            multi = len(keys_of[gram[:q2]]) > 1
            assert bool(nxt) == (multi and gram not in keys_of[gram[:q2]]) or (nxt and K == q2)
            salt = q2 if (nxt or not multi) else K
            home = capi.prefix_slot(gram, salt, lg)
            if home != e:
                want_more[home] = want_more.get(home, 0) | 1 << (8 + ((capi.prefix_hash(gram, salt) >> 11) & 15))
            if nxt:
                assert home == e  # redirect entries are placed first: always in their home slot
        for e in used:
            assert int(tab[e, 2]) & 0x00FFFF00 == want_more.get(int(e), 0)
        assert 0 < h.t.filter_density <= 3 * len(pats) / (32 << 14)
        h.close()


def k1b_twin_occurrences(h, pats, hay: bytes):
    """Python twin of K1b + the verification over the host tables: level 1 (both positions of a pair), the
    prefix table, the candidate list, the anchor shift of every candidate, the comparison of the whole
    pattern at hit - shift.  -> sorted (pattern, start, end) of the LONG patterns (3 bytes or more)."""
    q, q2, lg = int(h.t.filter_q), int(h.t.filter_q2), int(h.t.prefix_table_log2)
    g = q - 1
    out = []
    pad = hay + bytes(32)
    for i in range(len(hay)):
        if i + int(h.t.long_min_len) > len(hay):
            break
        # level 1 as the kernel pairs it: position j (even) tests X of the gram at j+1 with byte j, position j+1
        # tests Y of the same gram with byte j+q; both need the gate bit of X
        if i % 2 == 0:
            gram = pad[i + 1:i + 1 + g]
            H = capi.filter_hash(gram)
            x = int(h.filter_xy[H >> 18, 0])
            ok = (x >> (pad[i] & 31)) & (x >> (int.from_bytes(gram, "little") & 31)) & 1
        else:
            gram = pad[i:i + g]
            H = capi.filter_hash(gram)
            x, y = int(h.filter_xy[H >> 18, 0]), int(h.filter_xy[H >> 18, 1])
            ok = (y >> (pad[i + q - 1] & 31)) & (x >> (int.from_bytes(gram, "little") & 31)) & 1
        if not ok:
            continue
        for code in prefix_candidates(h, lg, q2, pad[i:i + 16], coded=True):
            pid, d = code & 0xFFFFFF, (code >> 24) & 15
            assert d == int(h.pattern_shift[pid]) <= int(h.t.max_shift)
            if d:
                assert bytes(np.asarray(h.pattern_head[pid]).astype("<u4").tobytes()[:min(12, len(pats[pid]))]) == pats[pid][:12]
            s0 = i - d
            if s0 >= 0 and hay[s0:s0 + len(pats[pid])] == pats[pid]:
                out.append((pid, s0, s0 + len(pats[pid])))
    return sorted(out)


def test_anchors_move_crowded_beginnings_and_lose_nothing():
    """Round 4: patterns whose first Q bytes are common by the pattern set's own byte statistics are filed under a
    later, rarer offset (cfg5's kind of set: hundreds of patterns behind one 4-byte character; URL lists behind
    "http:").  Whatever
    the anchors, the twin of the scan + verification over the host tables finds every occurrence of every long
    pattern -- at the very start of the haystack (the anchor lies behind the start), at its very end, for
    duplicates, for patterns that are prefixes of one another."""
    face = "\U0001F926"
    urls = [b"http://" + w for w in (b"alpha.example/x", b"beta.example", b"gamma.org/abcdef", b"delta", b"epsilon.net", b"ab")]
    words = [(face + c + t).encode() for c in "abcdef" for t in ("tail", "xy", "longer tail here", "zzz", "tailor")] + ["xyzzy".encode(), (face + face + "q").encode()]
    for pats in (urls, words, urls + words + [urls[0], words[0], b"http:", b"http://a"]):
        for mk in (capi.MATCH_STANDARD, capi.MATCH_LEFTMOST_LONGEST):
            h = capi.HostAutomaton(pats, mk)
            assert int(h.t.max_shift) > 0
            shifted = [i for i in range(len(pats)) if h.pattern_shift[i]]
            assert shifted
            for i in shifted:  # the suffix behind the anchor keeps the set-wide minimum length
                assert len(pats[i]) - int(h.pattern_shift[i]) >= int(h.t.long_min_len)
            rng = random.Random(5)
            hay = bytearray()
            for _ in range(60):
                hay += rng.choice(pats) if rng.random() < 0.7 else bytes(rng.choice(b"htp:/ab\xf0\x9f\xa4\xa6xy") for _ in range(rng.randint(1, 9)))
                if rng.random() < 0.3:
                    hay = hay[:-rng.randint(1, 3)]  # truncated copies
            hay = bytes(hay)
            want = sorted(m for m in occurrences(pats, hay) if len(pats[m[0]]) >= 3)
            if mk != capi.MATCH_STANDARD:  # identical patterns: the tables hold the first only
                want = [m for m in want if pats.index(pats[m[0]]) == m[0]]
            assert k1b_twin_occurrences(h, pats, hay) == want
            h.close()
    # sets without crowded beginnings are filed as they were
    for pats in (gen.gen_patterns(3000, 5, 12, gen.AZ, 1), gen.gen_patterns(500, 3, 9, gen.ALL_BYTES, 3)):
        h = capi.HostAutomaton(pats)
        assert int(h.t.max_shift) == 0 and not np.any(h.pattern_shift) and len(h.pattern_head) == 0
        h.close()
    # cfg5's shape: patterns that begin with a multi-byte character are filed behind it
    uni = [p.encode() for p in dict.fromkeys(gen.gen_patterns(3000, 5, 12, gen.AZ_UNI, 5))]
    h = capi.HostAutomaton(uni)
    lead4 = [i for i, p in enumerate(uni) if p[0] == 0xF0 and len(p) >= 12]
    assert len(lead4) > 30 and sum(h.pattern_shift[i] >= 3 for i in lead4) >= 0.8 * len(lead4)
    plain = [i for i, p in enumerate(uni) if max(p) < 0x80]
    assert len(plain) > 300 and not any(h.pattern_shift[i] for i in plain)
    hay = gen.gen_unicode_textlike(3000, 56, [p.decode() for p in uni], plant_every=64).encode()
    assert k1b_twin_occurrences(h, uni, hay) == sorted(occurrences(uni, hay))
    h.close()


def short_survivor(h, hay: bytes, p: int, lead: int = 0) -> bool:
    """Python twin of K1b's side test (K1B_ROW_SHORT): the positions with an even / odd INDEX (position + lead)
    share the read of short_xy[middle byte]; bytes beyond the end are arbitrary (here: 0xAA)."""
    b = lambda i: hay[i] if i < len(hay) else 0xAA
    if (p + lead) % 2 == 0:
        return bool(int(h.short_xy[b(p + 1), 0]) >> (b(p) & 31) & 1)
    return bool(int(h.short_xy[b(p), 1]) >> (b(p + 1) & 31) & 1)


def short_codes_at(h, hay: bytes, p: int):
    """Python twin of the settle step: the pattern ids of the 1-byte and the 2-byte pattern at p."""
    out = []
    keys = [hay[p]] + ([256 + (hay[p] | hay[p + 1] << 8)] if p + 1 < len(hay) else [])
    for k in keys:
        code = int(h.short_codes[k])
        if code == 0xFFFFFFFF:
            continue
        if code & 0x80000000:
            i = code & 0x7FFFFFFF
            out += [int(x) for x in h.prefix_lists[i + 1:i + 1 + int(h.prefix_lists[i])]]
        else:
            out.append(code)
    return out


def test_short_pattern_side_tables():
    """Round 4: patterns of 1 and 2 bytes are found by K1b's side test.  Its pair table lets every true start
    through (at both parities of the index, at the very end of the haystack too) and the exact codes name the
    patterns -- all copies for Standard, the first only for the leftmost kinds."""
    rng = random.Random(17)
    for it in range(120):
        alpha = [b"ab", b"abcdefgh", bytes(range(97, 123)) + b" ", bytes(range(256))][it % 4]
        pats = [bytes(rng.choice(alpha) for _ in range(rng.choice([1, 1, 2, 2, 2, 3, 5, 7]))) for _ in range(rng.randint(1, 30))]
        if it % 5 == 0:
            pats += [pats[0], pats[-1]]  # duplicates
        hay = bytes(rng.choice(alpha) for _ in range(rng.randint(1, 200)))
        for mk in (capi.MATCH_STANDARD, capi.MATCH_LEFTMOST_FIRST):
            h = capi.HostAutomaton(pats, mk)
            shorts = [i for i, p in enumerate(pats) if len(p) <= 2]
            assert int(h.t.n_short) == len(shorts)
            if not shorts:
                assert len(h.short_codes) == 0
                h.close()
                continue
            assert int(h.t.short_min_len) == min(len(pats[i]) for i in shorts)
            for p in range(len(hay)):
                want = [i for i in shorts if hay[p:p + len(pats[i])] == pats[i]]
                if mk != capi.MATCH_STANDARD:  # identical patterns: only the first can be reported
                    want = [i for i in want if pats.index(pats[i]) == i]
                got = short_codes_at(h, hay, p)
                assert sorted(got) == want, (pats, hay, p)
                if want:
                    assert short_survivor(h, hay, p, 0) and short_survivor(h, hay, p, 1), (pats, hay, p)
            h.close()
    # a text-like case: the side test is selective where it can be (letters differ in their low five bits)
    h = capi.HostAutomaton([b"qz", b"~"] + gen.gen_patterns(50, 5, 9, gen.AZ, 3))
    hay = gen.gen_textlike(20000, 5).tobytes()
    surv = sum(short_survivor(h, hay, p) for p in range(len(hay)))
    true = sum(hay[p:p + 2] == b"qz" or hay[p:p + 1] == b"~" for p in range(len(hay)))
    assert true <= surv <= true + len(hay) // 200
    h.close()


def test_prefix_keys_extend_beyond_the_shortest_pattern():
    """A set that mixes a 5-byte pattern with patterns starting with a 4-byte UTF-8 character: the
    long ones are filed under 8 bytes, so the character followed by an arbitrary byte is NOT a hit."""
    pats = ["xyzzy".encode()] + [("\U0001F926" + c + "tail").encode() for c in "abcdefghijklmnopqrstuvwxyz"]
    h = capi.HostAutomaton(pats)
    lg, q2 = int(h.t.prefix_table_log2), int(h.t.filter_q2)
    assert q2 == 5
    face = "\U0001F926".encode()
    assert prefix_candidates(h, lg, q2, (face + b"a" + b"tail" + bytes(8))[:16]) == [1]
    assert prefix_candidates(h, lg, q2, (face + b"a" + b"tXil" + bytes(8))[:16]) == []
    assert prefix_candidates(h, lg, q2, (b"xyzzy" + bytes(11))) == [0]
    h.close()


def test_compile_errors():
    with pytest.raises(ValueError) as e:
        capi.HostAutomaton([b"x", b""])
    assert "empty pattern" in str(e.value)
    h = capi.HostAutomaton([])  # zero patterns is legal
    assert h.n_states == 1 and h.t.filter_q == 0
    h.close()


def test_drop_in_package_name():
    """`import ahocorasick_rs` -- the reference's package name and its native-submodule layout
    (/root/reference/pysrc/ahocorasick_rs/__init__.py:2-7, src/lib.rs:438-445) -- resolves to
    this build; importing it pulls in neither numpy nor torch (the reference needs neither)."""
    import subprocess
    import sys
    code = ("import sys, ahocorasick_rs, ahocorasick_rs.ahocorasick_rs as native, ahocorasick_rs_amd as amd;"
            "assert ahocorasick_rs.AhoCorasick is native.AhoCorasick is amd.AhoCorasick;"
            "assert ahocorasick_rs.MatchKind.Standard is amd.MatchKind.Standard;"
            "assert ahocorasick_rs.MATCHKIND_LEFTMOST_LONGEST is amd.MatchKind.LeftmostLongest;"
            "assert set(ahocorasick_rs.__all__) == set(amd.__all__);"
            "assert 'numpy' not in sys.modules and 'torch' not in sys.modules, 'heavy import';"
            "import ahocorasick_rs_amd.capi; assert 'numpy' in sys.modules and 'torch' not in sys.modules")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("name", ["ahocorasick_rs", "ahocorasick_rs_amd"])
def test_extension_api_surface_and_validation(name):
    ac = pytest.importorskip(name)
    assert ac.MATCHKIND_STANDARD == ac.MatchKind.Standard
    assert ac.MATCHKIND_LEFTMOST_FIRST == ac.MatchKind.LeftmostFirst
    assert ac.MATCHKIND_LEFTMOST_LONGEST == ac.MatchKind.LeftmostLongest
    assert ac.MatchKind.Standard != ac.MatchKind.LeftmostFirst
    assert repr(ac.Implementation.DFA) == "Implementation.DFA"
    assert set(ac.__all__) >= {"AhoCorasick", "BytesAhoCorasick", "MatchKind", "Implementation"}
    # validation happens before any device work (reference tests/test_ac.py:75-83,157-168)
    with pytest.raises(TypeError):
        ac.AhoCorasick(None)
    with pytest.raises(TypeError):
        ac.AhoCorasick(["x", 12])
    with pytest.raises(TypeError):
        ac.BytesAhoCorasick(None)
    with pytest.raises(TypeError):
        ac.BytesAhoCorasick([b"x", 12])
    with pytest.raises(TypeError):
        ac.BytesAhoCorasick([b"x", "y"])
    for bad in ([""], ["", "xx"], ["xx", ""]):
        for sp in (True, False):
            with pytest.raises(ValueError) as e:
                ac.AhoCorasick(bad, store_patterns=sp)
            assert "You passed in an empty string as a pattern" in str(e.value)
    for bad in ([b""], [b"", b"xx"], [b"xx", b""]):
        with pytest.raises(ValueError) as e:
            ac.BytesAhoCorasick(bad)
        assert "You passed in an empty pattern" in str(e.value)
    with pytest.raises(TypeError):
        ac.AhoCorasick(["a"], matchkind="standard")
    with pytest.raises(TypeError):
        ac.AhoCorasick(["a"], implementation=2)
    # pyclass(eq) without eq_int (src/lib.rs:93, 113): an enum member is not equal to its discriminant
    assert ac.MatchKind.Standard != 0 and ac.Implementation.DFA != 2 and not (ac.MatchKind.LeftmostFirst == 1)
    assert ac.MatchKind.Standard != ac.Implementation.NoncontiguousNFA
    # `store_patterns: Option<bool>` takes a real bool (src/lib.rs:135)
    for bad in (1, 0, "yes"):
        with pytest.raises(TypeError):
            ac.AhoCorasick(["a"], store_patterns=bad)
    # PyBuffer::<u8>::get (src/lib.rs:286): unsigned one-byte items only
    import array
    for bad in (array.array("b", [1, 2]), array.array("H", [1, 2])):
        with pytest.raises(BufferError):
            ac.BytesAhoCorasick([bad])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_a_device():
    ac = pytest.importorskip("ahocorasick_rs_amd")
    with pytest.raises(RuntimeError) as e:
        ac.BytesAhoCorasick([b"hello"])
    assert "no HIP device" in str(e.value)
    with pytest.raises(Exception):
        capi.Automaton([b"hello"])


def failureless_walk_occurrences(h, hay: bytes):
    """Test-side twin of k1a_scan + k1a_walk over the host compiler's walk tables (NOT a product
    path): level 1 on symbols, the depth-3 record by class triple, the depth-4 record (a tail is
    compared, not walked), everything else walked from its node or from the root."""
    OWN, TAIL, SHORT, MANY = 1 << 31, 1 << 30, 1 << 31, 0xFFFFFFFE
    nc, n, out = h.n_classes, len(hay), []
    cls = [int(c) for c in h.classes]

    def emit_own(node, pos, d, own1):
        if own1 == MANY:
            for k in range(int(h.own_off[node]), int(h.own_off[node + 1])):
                out.append((int(h.own_pid[k]), pos, pos + d))
        else:
            out.append((own1, pos, pos + d))

    def walk(pos, node, d, first_rec=None):
        while True:
            r = [int(x) for x in (h.walk_grec[node] if first_rec is None else first_rec)]
            first_rec = None
            if r[1] & TAIL:
                tn = (r[1] >> 24) & 15
                tail = (r[0] | (r[3] << 32)).to_bytes(8, "little")[:tn]
                if pos + d + tn <= n and hay[pos + d:pos + d + tn] == tail:
                    out.append((r[2], pos, pos + d + tn))
                return
            if r[1] & OWN:
                emit_own(node, pos, d, r[2])
            if pos + d >= n:
                return
            c = cls[hay[pos + d]]
            if not (r[0] >> c) & 1:
                return
            node = (r[1] & ID_MASK) + bin(r[0] & ((1 << c) - 1)).count("1")
            d += 1

    for pos in range(n):
        if pos + 4 <= n:  # level 1: is there a trie path (or a short pattern) for the symbols of 4 bytes?
            s = [b & 31 for b in hay[pos:pos + 4]]
            if not (int(h.walk_t3b[((s[0] << 5) | s[1]) * 33 + s[2]]) >> s[3]) & 1:
                continue
        c = [cls[b] for b in hay[pos:pos + 5]] + [0] * 5
        x, y = (int(v) for v in h.walk_t3r[(c[0] * nc + c[1]) * nc + c[2]]) if pos + 3 <= n else (0, SHORT)
        if (y & SHORT) or pos + 5 > n:
            walk(pos, 0, 0)
            continue
        if not (x >> c[3]) & 1:
            continue
        node = (y & ID_MASK) + bin(x & ((1 << c[3]) - 1)).count("1")
        walk(pos, node, 4)
    return out


def test_failureless_walk_tables_enumerate_every_occurrence():
    """walk_t3b (symbols: a superset test -- alphabets that alias in their low five bits), walk_t3r,
    walk_grec with tail records; short patterns, nested patterns, duplicates; > 32 classes: no tables."""
    rng = random.Random(11)
    alphas = [b"ab", b"abc", b"abcdefgh", bytes(range(97, 123)) + b" ", b"aAbBcC!", bytes(range(40, 70))]
    for it in range(240):
        alpha = alphas[it % len(alphas)]
        lo = rng.choice([1, 1, 2, 3, 4, 5])
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(lo, lo + rng.choice([0, 2, 6, 14]))))
                for _ in range(rng.randint(1, 40))]
        if it % 3 == 0:
            pats += [pats[rng.randrange(len(pats))] for _ in range(3)]               # duplicates
            pats += [pats[rng.randrange(len(pats))][:rng.randint(1, 5)] for _ in range(3)]  # prefixes of others
        hay = bytearray(rng.choice(alpha) for _ in range(rng.randint(0, 400)))
        for _ in range(6):  # planted occurrences, some cut by the end of the haystack
            p = pats[rng.randrange(len(pats))]
            at = rng.randint(0, max(len(hay) - 1, 0))
            hay[at:at + len(p)] = p
        hay = bytes(hay[:400])
        h = capi.HostAutomaton(pats)
        assert len(h.walk_t3b) == 33 * 1024 and len(h.walk_grec) == h.n_states
        got = sorted(failureless_walk_occurrences(h, hay))
        assert got == sorted(occurrences(pats, hay)), (it, pats[:5])
        h.close()
    h = capi.HostAutomaton([bytes([b]) * 3 for b in range(64)])  # 65 byte classes
    assert len(h.walk_t3b) == 0 and len(h.walk_t3r) == 0 and len(h.walk_grec) == 0
    h.close()
