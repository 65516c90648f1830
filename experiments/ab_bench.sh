#!/bin/bash
# Same-box A/B of two builds of libsdamd.so on the headline bench: experiments/ab/libsdamd_old.so vs the in-tree one.
set -e
for i in 1 2; do
  SDAMD_LIB=$PWD/experiments/ab/libsdamd_old.so timeout -k 10 300 python bench.py --steps 2 --warmup 1 --skip-cpu-baseline --skip-roofline --skip-fp32-latency > gpurun_out/ab_old_$i.json
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --skip-cpu-baseline --skip-roofline --skip-fp32-latency > gpurun_out/ab_new_$i.json
done
python - <<'PY'
import json
for t in ("old", "new"):
    for i in (1, 2):
        d = json.loads(open(f"gpurun_out/ab_{t}_{i}.json").read().strip().splitlines()[-1])
        print(t, i, d["value"], d["ms_per_step"])
PY
