"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv (our kernels only)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if n.startswith("void at::") or "rocclr" in n or "Cijk" in n:
        continue
    acc[n.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-28s %.4g  (mean of %d)" % (c, sum(v) / len(v), len(v)))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        # SQ_BUSY_CYCLES sums over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
        print("    MFMA pipe busy               %.3f of the kernel's cycles" % ((sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024) / (sum(cs["SQ_BUSY_CYCLES"]) / 32)))
