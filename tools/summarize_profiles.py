"""Turn the rocprofv3 CSVs that tools/prof_bench.sh and tools/prof_pmc.sh leave under gpurun_out/ into the committed summaries
under profiles/ (run here after the gpurun call).  Usage: python tools/summarize_profiles.py rNN "<one-line description>" """
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, desc = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "profiles")

def newest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None

def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")

# ---- kernel stats from the kernel trace of prof_bench.sh
f = newest("gpurun_out/prof1/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(short(r["Kernel_Name"]), []).append(d)
tot = sum(sum(v) for v in agg.values())
items = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open(os.path.join(P, f"{tag}_bench_b32_t50_bf16_kernel_stats.txt"), "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline  ({desc})\n")
    o.write("# name | calls | avg_us | total_ms | percent\n")
    for k, v in items:
        o.write(f"{k[:100]} | {len(v)} | {sum(v)/len(v):.1f} | {sum(v)/1e3:.2f} | {100*sum(v)/tot:.1f}\n")
with open(os.path.join(P, f"{tag}_bench_b32_t50_bf16_kernel_stats.csv"), "w") as o:
    o.write("name,calls,avg_us,min_us,max_us,total_ms,percent\n")
    for k, v in items:
        o.write(f"\"{k}\",{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v)/1e3:.3f},{100*sum(v)/tot:.2f}\n")
line = [l for l in open(os.path.join(ROOT, "gpurun_out/prof1.log")) if l.startswith("{")]
if line:
    open(os.path.join(P, f"{tag}_bench_line_under_rocprof.json"), "w").write(line[-1])

# ---- HBM traffic from the two PMC passes
def pmc(dirpat, counter):
    f = newest(dirpat)
    if not f:
        return {}
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        out.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return out
fe = pmc("gpurun_out/pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE")
wr = pmc("gpurun_out/pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
if fe and wr:
    with open(os.path.join(P, f"{tag}_pmc_hbm_traffic.txt"), "w") as o:
        o.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 10 --warmup 3 (B=32,T=50,bf16), {desc}\n")
        o.write("# counters are KiB per dispatch.  MI355X_MICROARCH.md (HBM): FETCH_SIZE under-reports wide coalesced 16-B/lane streams by 2x on\n")
        o.write("# gfx950; the recurrent kernels mix 4-B, 8-B and 16-B accesses (uncalibrated widths), so the raw value is kept and the x2\n")
        o.write("# figure is shown beside it; WRITE_SIZE is exact for streaming stores.\n")
        o.write("# kernel | dispatches | FETCH_SIZE KiB (raw mean) | x2 MB | WRITE_SIZE KiB (mean) | MB\n")
        keys = sorted(set(fe) | set(wr), key=lambda k: -(sum(fe.get(k, [0])) + sum(wr.get(k, [0]))))
        traffic = {}
        for k in keys:
            a = fe.get(k, [0.0]); b = wr.get(k, [0.0])
            fm, wm = sum(a) / len(a), sum(b) / len(b)
            o.write(f"{k[:70]} | {len(a)} | {fm:.1f} | {2*fm*1024/1e6:.3f} | {wm:.1f} | {wm*1024/1e6:.3f}\n")
            traffic[k] = int((fm + wm) * 1024)
    def pick(sub):
        c = [v for k, v in traffic.items() if sub in k]
        return max(c) if c else None
    tj = {"batch": 32, "seq_len": 50, "precision": "bf16",
          "source": f"profiles/{tag}_pmc_hbm_traffic.txt (FETCH_SIZE raw + WRITE_SIZE, bytes per launch)",
          "bytes_per_launch": {"lstm_fwd_kernel": pick("lstm_fwd"), "lstm_bwd_kernel": pick("lstm_bwd")}}
    json.dump(tj, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(P)))
