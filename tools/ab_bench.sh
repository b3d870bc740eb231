#!/bin/sh
# A/B one dp_tune key in full bench runs on one box: tools/ab_bench.sh KEY "V0 V1 V0 V1" [bench.py flags...]
# Prints rows/s and per-kernel microseconds per run; the JSON lines go to gpurun_out/ab_<KEY>_<V>_<i>.json.
key=$1; vals=$2; shift 2
i=0
for v in $vals; do
    i=$((i + 1))
    out=gpurun_out/ab_${key}_${v}_${i}.json
    timeout -k 10 300 python bench.py --cpu-rows 0 --tune ${key}=${v} "$@" > $out 2> gpurun_out/ab_${key}.err || exit 1
    python - "$out" "$key=$v" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "rows/s", round(r["value"], 1), {k: round(v["avg_ms"] * 1e3, 1) for k, v in r.get("kernels", {}).items()}, flush=True)
PY
done
