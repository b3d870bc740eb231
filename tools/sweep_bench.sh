#!/bin/sh
# Sweep of bench.py flags on one box; each line: the flags, rows/s, per-kernel microseconds.
#   tools/sweep_bench.sh "--m-chunk 32768" "--m-chunk 0" ...
i=0
for flags in "$@"; do
    i=$((i + 1))
    out=gpurun_out/sweep_$i.json
    timeout -k 10 300 python bench.py --cpu-rows 0 --steps 1 --warmup 1 --rows 16384 $flags > $out 2> gpurun_out/sweep.err || { echo "FAILED: $flags"; tail -2 gpurun_out/sweep.err; continue; }
    python - "$out" "$flags" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "| rows/s", round(r["value"], 1), {k: round(v["avg_ms"] * 1e3, 1) for k, v in r.get("kernels", {}).items()}, flush=True)
PY
done
