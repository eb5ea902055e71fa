#!/bin/bash
# Runs bench.py on every workload of BASELINE.json (GPU box, via gpurun) and prints one line each:
#   bash tools/bench_all.sh <out-subdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd "$R"
run() {  # name args...
  local name=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras "$@" > "$OUT/bench_$name.json" || { echo "$name failed"; exit 1; }
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-12s %7.3f Gsamples/s  fwd %.3f ms  bwd %.3f ms" % (sys.argv[2], d["value"], d["detail"]["fwd_ms"], d["detail"]["bwd_ms"]))
PY
}
run C2 --workload C2 &&
run C3 --workload C3 &&
run C5 --workload C5 &&
run C5x16 --workload C5x16 &&
run C3_bf16 --workload C3 --dtype bf16 &&
run C2_uniform --workload C2 --dist uniform
