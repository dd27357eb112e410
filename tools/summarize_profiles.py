#!/usr/bin/env python3
"""Turn the scratch output of tools/collect_profiles.sh (gpurun_out/<round>/) into the
committed evidence under profiles/<round>/:
  bench_*.json                              the bench lines
  rocprofv3_kernel_stats_bench_<X>.csv      rocprofv3 --kernel-trace --stats, as emitted
  rocprofv3_kernel_summary_bench_<X>.txt    per-kernel summary + timeline of the same trace
  rocprofv3_pmc.json                        PMC passes: mean counters per dispatch and kernel
  pmc_traffic.json                          HBM traffic of the scan kernels, one entry per measured
                                            configuration (bench.py reads it)
usage: summarize_profiles.py [round]"""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC, DST = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles", R)
os.makedirs(DST, exist_ok=True)


def short(n):
    return n.replace("acx::", "").replace("void ", "").split("(")[0].split("<")[0]


for f in sorted(glob.glob(os.path.join(SRC, "bench_*.json"))) + [os.path.join(SRC, x) for x in
                                                                  ("smoke.log", "bench_comparison.txt", "bench_comparison_stream_sync.txt", "ubench_stream.txt", "k0_probe.txt",
                                                                   "exp_k0_no_prefilter_bench_comparison.txt", "exp_no_resident_no_inplace_bench_comparison.txt", "pytest_gpu.log")]:
    if os.path.exists(f) and os.path.getsize(f):
        if f.endswith(".json"):  # (only the JSON line: RCCL prints its banner to stdout)
            lines = [l for l in open(f) if l.startswith("{")]
            if lines:
                open(os.path.join(DST, os.path.basename(f)), "w").write(lines[-1])
        else:
            shutil.copy(f, os.path.join(DST, os.path.basename(f)))
P = "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-size --no-cold --no-secondary"
for tag, cmd in (("T", "python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --no-secondary"),
                 ("T_8gib", "python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --no-secondary --bytes 8589934592"),
                 ("mixed", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-target-size --config mixed"),
                 ("cfg4", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --config cfg4"),
                 ("cfg5", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --config cfg5"),
                 ("dfa_walk", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-target-size --no-cold --kernel dfa_walk"),
                 ("dense_D", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-secondary --no-target-size --dist D"),
                 ("H1", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-secondary --no-target-size --dist H1"),
                 ("H100", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-secondary --no-target-size --dist H100"),
                 ("k0_short", "python tools/k0_probe.py short indexes 5000"),
                 ("large", "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cold --config large")):
    src = os.path.join(SRC, f"trace_{tag}")
    if not os.path.exists(os.path.join(src, "bench_kernel_stats.csv")):
        continue
    shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(DST, f"rocprofv3_kernel_stats_bench_{tag}.csv"))
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), src, "--timeline", "16"],
                         capture_output=True, text=True).stdout
    open(os.path.join(DST, f"rocprofv3_kernel_summary_bench_{tag}.txt"), "w").write(
        f"# rocprofv3 --kernel-trace --stats -- {cmd}\n" + txt)

# the hash collect_profiles.sh recorded on the GPU box next to its measurements (never today's working tree:
# a summary written after the kernels changed must not relabel old counters)
KSHA = open(os.path.join(SRC, "kernel_source_sha256.txt")).read().strip() if os.path.exists(os.path.join(SRC, "kernel_source_sha256.txt")) else None
if KSHA:
    open(os.path.join(DST, "kernel_source_sha256.txt"), "w").write(KSHA + "\n")
pmc = {}
for d in sorted(os.listdir(SRC)):
    f = os.path.join(SRC, d, "r_counter_collection.csv")
    if not d.startswith("pmc_") or not os.path.exists(f):
        continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    dur = collections.OrderedDict()
    for r in csv.DictReader(open(os.path.join(SRC, d, "r_kernel_trace.csv"))):
        dur.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    keep = ("k1b_prefilter", "k1a_scan", "k1a_walk", "k1a_walk16", "k1a_dfa_walk", "k_tile_main", "k_tile_write", "k_walk_hits",
            "k_dense_verify", "k_dense_main", "k_hot_verify")
    pmc[d] = {k: {"dispatches": len(dur.get(k, [])), "mean_duration_us": round(sum(dur[k]) / len(dur[k]), 1) if k in dur else None,
                  "mean_counters": {c: round(sum(v) / len(v)) for c, v in cs.items()}}
              for k, cs in acc.items() if k in keep}
json.dump({"command": P + " [--dist Z | --kernel dfa_walk | --config cfg4 | cfg5 | mixed | large]", "passes": pmc},
          open(os.path.join(DST, "rocprofv3_pmc.json"), "w"), indent=1)


def counter(pass_name, kernel, name):
    return pmc[pass_name][kernel]["mean_counters"][name]


NOTE = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh), mean per "
        "dispatch.  gfx950 correction: FETCH_SIZE reports half of a wide coalesced stream; calibrated on these very "
        "kernels with an all-zero haystack (nothing survives level 1: the scan reads the haystack and nothing else), "
        "which yields stream_calibration_factor.  traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as the MI355X "
        "guide prescribes; traffic_bytes_gathers_1to1 counts only the stream part (the zero-haystack reading) twice "
        "and the rest (window re-reads, table probes, trie records, table copies to LDS) once.")
entries = []


def entry(kernel, kernels, cfg, dist, bench_file, fetch_pass, write_pass, zero_pass):
    try:
        bench = json.load(open(os.path.join(DST, bench_file)))
        nbytes = bench["config"]["bytes_per_gpu"]
        fetch = sum(counter(fetch_pass, k, "FETCH_SIZE") for k in kernels if k in pmc[fetch_pass])
        write = sum(counter(write_pass, k, "WRITE_SIZE") for k in kernels if k in pmc[write_pass])
        e = {"kernel": kernel, "config": cfg, "workload_bytes": nbytes, "dist": dist, "FETCH_SIZE_KB": fetch,
             "WRITE_SIZE_KB": write, "traffic_bytes": int((2 * fetch + write) * 1024),
             "algorithmic_bytes": bench["roofline"]["algorithmic_bytes"], "bench_line": bench_file}
        if zero_pass and zero_pass in pmc:
            zero = sum(counter(zero_pass, k, "FETCH_SIZE") for k in kernels[:1] if k in pmc[zero_pass])
            e["FETCH_SIZE_KB_zero_haystack"] = zero
            e["stream_calibration_factor"] = round(nbytes / (zero * 1024), 3)
            e["traffic_bytes_gathers_1to1"] = int((2 * zero + max(fetch - zero, 0) + write) * 1024)
            e["traffic_over_algorithmic_gathers_1to1"] = round(e["traffic_bytes_gathers_1to1"] / e["algorithmic_bytes"], 3)
        e["traffic_over_algorithmic"] = round(e["traffic_bytes"] / e["algorithmic_bytes"], 3)
        e["kernel_source_sha256"] = KSHA  # the device code these counters were read from (tools/kernel_hash.py)
        entries.append(e)
    except (KeyError, FileNotFoundError, ZeroDivisionError) as ex:
        print("no traffic entry for", kernel, cfg, ":", repr(ex))


entry("k1b_prefilter", ["k1b_prefilter"], "cfg2", "T", "bench_T.json", "pmc_fetch", "pmc_write", "pmc_fetch_zero_haystack")
entry("k1a_scan+k1a_walk", ["k1a_scan", "k1a_walk"], "cfg2", "T", "bench_T_dfa_walk.json", "pmc_fetch_dfa_walk",
      "pmc_write_dfa_walk", "pmc_fetch_dfa_walk_zero_haystack")
entry("k1b_prefilter", ["k1b_prefilter"], "cfg4", "T", "bench_cfg4.json", "pmc_fetch_cfg4", "pmc_write_cfg4", None)
entry("k1b_prefilter", ["k1b_prefilter"], "cfg5", "T", "bench_cfg5.json", "pmc_fetch_cfg5", "pmc_write_cfg5", None)
json.dump({"note": NOTE, "entries": entries} if False else entries, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(DST, "pmc_traffic_README.txt"), "w").write(NOTE + "\n")
print(json.dumps(entries, indent=1))
for tag in ("T", "dfa_walk"):
    f = os.path.join(DST, f"rocprofv3_kernel_summary_bench_{tag}.txt")
    if os.path.exists(f):
        print(open(f).read())
