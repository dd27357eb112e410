import sys, time, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import gen
from ahocorasick_rs_amd import capi
pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
ac = capi.Automaton(pats, capi.MATCH_STANDARD, capi.IMPL_DFA)
n = 1 << 30
hay = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.generate(hay.data_ptr(), n, 1, 11)
def run(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        r = ac.find_device(hay.data_ptr(), n); c = r.count; r.free()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    for prof in (False, True, False, True):
        ac.profile_enable(prof); run(3)
        print("profile", prof, "ms/step %.4f" % run(30))
