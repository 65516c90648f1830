"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES,
GRBM_GUI_ACTIVE) over eager UNet steps.   usage: pmc_mfma.py <pass dir> <out.json>
util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES)  (matrix-pipe busy cycles summed over SIMDs / SIMD-cycles
of the CUs that had waves), and vs the whole chip: / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv, glob, json, re, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"^void ", "", r["Kernel_Name"]); k = re.sub(r"\(.*$", "", k)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {}
for k, v in acc.items():
    L = max(n[(k, c)] for c in v)
    mf, cu, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CU_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    out[k] = {"launches": L, "mfma_busy_cycles_per_launch": round(mf / L), "busy_cu_cycles_per_launch": round(cu / L),
              "gui_active_per_launch": round(gui / L),
              "mfma_util_of_busy_cus": round(mf / (4 * cu), 4) if cu else None,
              "mfma_util_of_chip": round(mf / (1024 * gui / 8), 4) if gui else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_per_launch"] * kv[1]["launches"])[:16]:
    print(f'{v["launches"]:5d} x  chip {v["mfma_util_of_chip"]}  busyCU {v["mfma_util_of_busy_cus"]}  {k[:100]}')
