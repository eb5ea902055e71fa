#!/bin/bash
# rocprofv3 per-kernel averages for one library build: bash tools/kstats.sh <tag> <lib.so> [time_variant args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; LIB=$(cd "$(dirname "$2")" && pwd)/$(basename "$2"); shift; shift
OUT=$R/gpurun_out/ks_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- \
  python3 "$R/tools/time_variant.py" "$LIB" "$@" > "$OUT/out.txt" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
echo "== $TAG"; cat "$OUT/out.txt"
python3 - "$OUT/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "msda" in n:
        print("  %-60s calls %5s avg %9.1f us" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
