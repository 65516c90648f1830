"""Diff two outputs of experiments/unet_profile.py (old vs new library, same box): per-shape time deltas."""
import re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\s*([\d.]+) ms\s+[\d.]+% cum\s+[\d.]+%\s+x\s*(\d+)\s+[\d.]+ (?:TF|GB/s)\s+(\S+)\s+(.*)", l)
        if m: d[(m.group(3), m.group(4).strip())] = (float(m.group(1)), int(m.group(2)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(((b.get(k, (0, 0))[0] - v[0], k, v[0], b.get(k, (0, 0))[0]) for k, v in a.items()), key=lambda r: r[0])
print(f"total old {sum(v[0] for v in a.values()):.3f} ms  new {sum(v[0] for v in b.values()):.3f} ms")
for d, k, x, y in rows[:14] + rows[-10:]:
    print(f"{d*1e3:+8.1f} us  {x:.3f} -> {y:.3f}  {k[0]} {k[1]}")
