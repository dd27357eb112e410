import sys, time, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import gen
from ahocorasick_rs_amd import capi
pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
N = 1 << 30
buf = capi.DeviceBuffer(N)
for kern in (capi.KERNEL_PREFILTER, capi.KERNEL_DFA_WALK):
    a = capi.Automaton(pats, 0, capi.IMPL_DFA, kernel=kern)
    i = a.info
    print("kernel", capi.KERNEL_NAMES[i.kernel], "states", i.n_states, "classes", i.n_classes, "hot_rows", i.lds_hot_rows, "q", i.filter_q, flush=True)
    for kind, seed, name in ((1, 11, "T"), (0, 12, "U")):
        a.generate(buf.ptr, N, kind, seed)
        a.profile_enable(True)
        for it in range(4):
            t0 = time.perf_counter()
            r = a.find_device(buf.ptr, N)
            t1 = time.perf_counter()
            n = r.count; r.free()
            p = a.profile_read(True)
            print(f"  {name} it{it}: matches={n} raw={p.raw_occurrences} scan_ms={p.scan_ms:.3f} post_ms={p.post_ms:.3f} wall_ms={(t1-t0)*1e3:.3f} scan_GBps={N/p.scan_ms/1e6:.1f} wall_GBps={N/(t1-t0)/1e9:.1f}", flush=True)
    a.close()
