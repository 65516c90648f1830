"""Busy fraction and gap histogram of the last denoising loop in a rocprofv3 kernel trace."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 50 cfg_ddpm_step kernels delimit the timed generate() call's denoising loop
idx = [i for i, r in enumerate(rows) if "cfg_ddpm_step" in r[2]]
last, first = idx[-1], idx[-50]
# one full step: between two consecutive sampler kernels in the middle of the loop
for a, b in ((idx[-26], idx[-25]), (idx[-2], idx[-1])):
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
    print(f"step: {len(seg)} kernels, span {span/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms ({100*busy/span:.1f} %), "
          f"gaps total {sum(gaps)/1e6:.3f} ms, mean {sum(gaps)/len(gaps)/1e3:.2f} us, max {max(gaps)/1e3:.1f} us")
    h = collections.Counter(min(int(g / 1000), 10) for g in gaps)
    print("  gap histogram (us: count):", dict(sorted(h.items())))
    by = collections.defaultdict(float)
    for i in range(len(seg) - 1):
        by[seg[i][2].split("(")[0][:60]] += gaps[i]
    for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]:
        print(f"  gap after {k}: {v/1e3:.1f} us")
