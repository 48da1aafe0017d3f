#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# A/B of library variants on the whole benchmark step inside ONE GPU call: each argument is a quoted list of VAR=VALUE
# settings ("" = defaults); prints images/s and the per-family ms per step of every run.
#   tools/benchvar.sh "" "FQ_PWS_AUTO=1" "FQ_PWS_AUTO=1 FQ_PWS_CW=4"
for v in "$@"; do
  out=$(env $v python3 bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1)
  python3 - "$v" "$out" <<'P'
import json, sys
v, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
except Exception:
    print("%-40s FAILED: %s" % (v, line[:200])); sys.exit(0)
k = d["roofline"]["kernels"]
print("%-40s %9.1f img/s  %.4f ms/step | " % (v or "(defaults)", d["value"], d["ms_per_step"]) +
      "  ".join("%s %.4f" % (n, k[n]["ms_per_step"]) for n in sorted(k)))
P
done
