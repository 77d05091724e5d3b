"""Per-kernel table of SQ counters from several rocprofv3 --pmc passes (tools/pmc_sq_bench.sh): mean per dispatch and the ratios that say
where a wave's cycles go.  usage: pmc_sq_table.py <dir with p1..pN> [name filter]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "conv_dma_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if filt in k:
                acc[k.split("(")[0] if "<" not in k else k[:k.index(">") + 1]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# rocprofv3 --pmc, bench.py --steps 2 --warmup 1 --no-graph (B=4), mean per dispatch; ratios: share of SQ_WAVE_CYCLES (all waves) unless noted")
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print(k.replace("void ddx::(anonymous namespace)::", ""), f"  dispatches {n}")
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:16.1f}")
    r = lambda c: m.get(c, 0.0) / wc
    print(f"   -> wave parked (s_waitcnt / barrier) {r('SQ_WAIT_ANY'):.3f}, issue stalled {r('SQ_WAIT_INST_ANY'):.3f}, issuing {r('SQ_ACTIVE_INST_ANY'):.3f}"
          f" (VALU {r('SQ_ACTIVE_INST_VALU'):.3f}, LDS {r('SQ_ACTIVE_INST_LDS'):.3f}, VMEM {r('SQ_ACTIVE_INST_VMEM'):.3f}, scalar {r('SQ_ACTIVE_INST_SCA'):.3f})")
    if m.get("SQ_INSTS_MFMA"):
        print(f"   -> per MFMA: VALU {m.get('SQ_INSTS_VALU', 0) / m['SQ_INSTS_MFMA']:.2f} (of which transcendental {m.get('SQ_INSTS_VALU_TRANS_F32', 0) / m['SQ_INSTS_MFMA']:.2f}),"
              f" SALU {m.get('SQ_INSTS_SALU', 0) / m['SQ_INSTS_MFMA']:.2f}, LDS {m.get('SQ_INSTS_LDS', 0) / m['SQ_INSTS_MFMA']:.2f};"
              f" MFMA pipe busy / SQ busy cycles {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(m.get('SQ_BUSY_CYCLES', 1), 1):.3f}")
