import random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from ahocorasick_rs_amd import capi
from oracle_lib import KIND_DFA, Oracle
for kernel in (capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER):
    for mk in (0, 1, 2):
        rng = random.Random(1000 + mk)
        for it in range(60):
            alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][it % 4]
            pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 7))) for _ in range(rng.randint(1, 20))]
            hay = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 300)))
            for ov in ((False, True) if mk == 0 else (False,)):
                print("case", kernel, mk, it, ov, len(pats), len(hay), flush=True)
                a = capi.Automaton(pats, mk, kernel=kernel)
                print("  built", a.info.kernel, flush=True)
                got = [(int(p), int(s), int(e)) for (p, s, e) in a.find(hay, overlapping=ov)]
                want = Oracle(pats, mk, KIND_DFA).find(hay, overlapping=ov)
                a.close()
                assert got == want, (pats, hay)
print("ok")
