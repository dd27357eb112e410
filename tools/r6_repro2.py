import random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from ahocorasick_rs_amd import capi
from oracle_lib import KIND_DFA, Oracle
kernel, mk = capi.KERNEL_PREFILTER, 0
rng = random.Random(1000 + mk)
for it in range(2):
    alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][it % 4]
    pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 7))) for _ in range(rng.randint(1, 20))]
    hay = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 300)))
    if it < 1: continue
    print("case", it, pats, hay, flush=True)
    a = capi.Automaton(pats, mk, kernel=kernel)
    print("  built", a.info, flush=True)
    got = [(int(p), int(s), int(e)) for (p, s, e) in a.find(hay, overlapping=False)]
    want = Oracle(pats, mk, KIND_DFA).find(hay, overlapping=False)
    print(len(got), len(want), got == want, a.path_stats())
    a.close()
print("ok")
