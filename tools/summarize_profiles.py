"""Turn the rocprofv3 CSVs that tools/prof_bench.sh and tools/prof_pmc.sh leave under gpurun_out/ into the committed summaries
under profiles/ (run here after the gpurun calls).

    python tools/summarize_profiles.py r02 32 "<one-line description of the state>"

Writes profiles/<tag>_bench_b<B>_t50_bf16_kernel_stats.{txt,csv}, profiles/<tag>_bench_b<B>_line_under_rocprof.json,
profiles/<tag>_pmc_b<B>.txt (HBM traffic + MFMA-busy per kernel) and, for B = 32 (the headline config), profiles/pmc_traffic.json
(what bench.py quotes as roofline.traffic / mfma_busy)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, B, desc = sys.argv[1], int(sys.argv[2]), sys.argv[3]
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")


# ---- kernel stats from the kernel trace of prof_bench.sh
f = newest(f"gpurun_out/prof_b{B}/*/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(short(r["Kernel_Name"]), []).append(d)
    tot = sum(sum(v) for v in agg.values())
    items = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    with open(os.path.join(P, f"{tag}_bench_b{B}_t50_bf16_kernel_stats.txt"), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --batch {B} --no-cpu-baseline  ({desc})\n")
        o.write("# name | calls | avg_us | total_ms | percent\n")
        for k, v in items:
            o.write(f"{k[:100]} | {len(v)} | {sum(v)/len(v):.1f} | {sum(v)/1e3:.2f} | {100*sum(v)/tot:.1f}\n")
    with open(os.path.join(P, f"{tag}_bench_b{B}_t50_bf16_kernel_stats.csv"), "w") as o:
        o.write("name,calls,avg_us,min_us,max_us,total_ms,percent\n")
        for k, v in items:
            o.write(f"\"{k}\",{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v)/1e3:.3f},{100*sum(v)/tot:.2f}\n")
    log = os.path.join(ROOT, f"gpurun_out/prof_b{B}.log")
    line = [l for l in open(log) if l.startswith("{")] if os.path.exists(log) else []
    if line:
        open(os.path.join(P, f"{tag}_bench_b{B}_line_under_rocprof.json"), "w").write(line[-1])


# ---- PMC passes
def pmc(dirpat):
    """{kernel: {counter: [values per dispatch]}, ...}, {kernel: [duration us]}"""
    f = newest(dirpat)
    out, dur = collections.OrderedDict(), collections.OrderedDict()
    if not f:
        return out, dur
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        out.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out, dur


fe, _ = pmc(f"gpurun_out/pmc_fetch_b{B}/*/*counter_collection.csv")
wr, _ = pmc(f"gpurun_out/pmc_write_b{B}/*/*counter_collection.csv")
mf, mdur = pmc(f"gpurun_out/pmc_mfma_b{B}/*/*counter_collection.csv")
mean = lambda d, k, c: (sum(d[k][c]) / len(d[k][c])) if (k in d and c in d[k]) else 0.0
if fe and wr:
    traffic, busy = {}, {}
    with open(os.path.join(P, f"{tag}_pmc_b{B}.txt"), "w") as o:
        o.write(f"# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 10 --warmup 3 --batch {B} --no-cpu-baseline   ({desc})\n")
        o.write("# three passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16/F32 GRBM_GUI_ACTIVE\n")
        o.write("# HBM (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports HALF the bytes of wide\n")
        o.write("# coalesced reads -> `fetch_MB` below is 2 x FETCH_SIZE; WRITE_SIZE is exact for streaming stores; Infinity-Cache hits are counted.\n")
        o.write("# MFMA: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs); gui = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs);\n")
        o.write("# mfma_util_chip = mfma_busy / (gui x 1024 SIMDs) (the MfmaUtil expression of rocprofv3); mfma_busy / wave_cycles = mfma_busy over\n")
        o.write("# 4 x SQ_WAVE_CYCLES (that counter is in quad-cycles): the share of a resident wave's life in which its SIMD's matrix pipe is busy.\n")
        o.write("# kernel | dispatches | avg_us | fetch_MB (2 x FETCH_SIZE) | write_MB | hbm_MB | mfma_busy_cyc | gui_cyc | mfma_util_chip % | mfma_busy / wave_cycles % | MFMA MOPS bf16 | f32\n")
        keys = sorted(set(fe) | set(wr) | set(mf), key=lambda k: -(mean(fe, k, "FETCH_SIZE") * 2 + mean(wr, k, "WRITE_SIZE")) * len(fe.get(k, {}).get("FETCH_SIZE", [0])))
        for k in keys:
            n = len(fe.get(k, {}).get("FETCH_SIZE", [])) or len(mf.get(k, {}).get("GRBM_GUI_ACTIVE", []))
            fm = 2.0 * mean(fe, k, "FETCH_SIZE") * 1024 / 1e6
            wm = mean(wr, k, "WRITE_SIZE") * 1024 / 1e6
            mb = mean(mf, k, "SQ_VALU_MFMA_BUSY_CYCLES")
            gui = mean(mf, k, "GRBM_GUI_ACTIVE") / 8.0
            wc = 4.0 * mean(mf, k, "SQ_WAVE_CYCLES")
            du = (sum(mdur[k]) / len(mdur[k])) if k in mdur else 0.0
            util = 100.0 * mb / (gui * 1024) if gui > 0 else 0.0
            uw = 100.0 * mb / wc if wc > 0 else 0.0
            o.write(f"{k[:72]} | {n} | {du:.1f} | {fm:.3f} | {wm:.3f} | {fm + wm:.3f} | {mb:.0f} | {gui:.0f} | {util:.2f} | {uw:.2f} | "
                    f"{mean(mf, k, 'SQ_INSTS_VALU_MFMA_MOPS_BF16'):.0f} | {mean(mf, k, 'SQ_INSTS_VALU_MFMA_MOPS_F32'):.0f}\n")
            traffic[k] = int((fm + wm) * 1e6)
            busy[k] = (round(util, 3), round(uw, 3))

    def pick(d, sub):
        c = [(k, v) for k, v in d.items() if sub in k]
        return max(c, key=lambda kv: kv[1] if not isinstance(kv[1], tuple) else kv[1][0])[1] if c else None
    if B == 32:
        tj = {"batch": 32, "seq_len": 50, "precision": "bf16",
              "source": f"profiles/{tag}_pmc_b32.txt (2 x FETCH_SIZE + WRITE_SIZE bytes per launch; MFMA-busy from SQ_VALU_MFMA_BUSY_CYCLES)",
              "bytes_per_launch": {"lstm_fwd_kernel": pick(traffic, "lstm_fwd"), "lstm_bwd_kernel": pick(traffic, "lstm_bwd")},
              "mfma_util_chip_percent": {"lstm_fwd_kernel": (pick(busy, "lstm_fwd") or (None, None))[0],
                                         "lstm_bwd_kernel": (pick(busy, "lstm_bwd") or (None, None))[0]},
              "mfma_busy_share_of_wave_cycles_percent": {"lstm_fwd_kernel": (pick(busy, "lstm_fwd") or (None, None))[1],
                                                         "lstm_bwd_kernel": (pick(busy, "lstm_bwd") or (None, None))[1]}}
        json.dump(tj, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(P)))
