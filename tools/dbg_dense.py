import sys, random
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
from oracle_lib import KIND_DFA, Oracle
from ahocorasick_rs_amd import capi
def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)
rng = random.Random(5)
pats = list({bytes(rng.choice(b"abc") for _ in range(rng.randint(1, 8))) for _ in range(3000)})
dense = bytes(rng.choice(b"abc") for _ in range(400_000))
hay = dense + b"z" * 1_600_000
for kernel in (capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER, None):
    for mk in (1, 0):
        a = capi.Automaton(pats, mk, kernel=kernel)
        o = Oracle(pats, mk, KIND_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            want = o.find_raw(hay, overlapping=ov)
            for rep in range(2):
                got = cols(a.find(hay, overlapping=ov))
                eq = np.array_equal(got, want)
                msg = ""
                if not eq:
                    n = min(len(got), len(want))
                    d = np.nonzero((got[:n] != want[:n]).any(1))[0]
                    msg = f"len got {len(got)} want {len(want)} first diff {d[:1]} got {got[d[0]] if len(d) else None} want {want[d[0]] if len(d) else None}"
                print("kernel", kernel, "mk", mk, "ov", ov, "rep", rep, "OK" if eq else "MISMATCH " + msg, flush=True)
        a.close()
