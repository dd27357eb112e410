"""per-step wall times of the headline workload from a cold start (how long does the transient last?)"""
import sys, time, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import gen
from ahocorasick_rs_amd import capi
pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
ac = capi.Automaton(pats, capi.MATCH_STANDARD, capi.IMPL_DFA)
n = 1 << 30
hay = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.generate(hay.data_ptr(), n, 1, 11)
torch.cuda.synchronize()
ac.profile_enable(True)
ts = []
for _ in range(80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ac.find_device(hay.data_ptr(), n); c = r.count; r.free()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms (synchronised each step):", " ".join("%.3f" % t for t in ts[:40]))
print("steps 40-80 mean %.4f" % (sum(ts[40:]) / 40))
time.sleep(2.0)
ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ac.find_device(hay.data_ptr(), n); c = r.count; r.free()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("after 2 s idle:", " ".join("%.3f" % t for t in ts))
