#!/bin/bash
# rocprofv3 per-kernel time of bench.py.  Usage (on the GPU box, via gpurun):
#   bash tools/kernel_stats.sh <out-subdir-under-gpurun_out> [bench.py args...]
# Leaves <subdir>/run_kernel_stats.csv (copy the one to be judged into profiles/rNN/).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- \
  python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
cut -d, -f1-8 "$OUT"/run_kernel_stats.csv | head -14
