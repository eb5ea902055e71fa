#!/bin/bash
# LDS / VALU counters of the tile kernel for library variants: bash tools/pmc_lds.sh lib1.so lib2.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  T=$(basename $lib .so); OUT=$R/gpurun_out/pl_$T; mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT -o run -- python3 $R/tools/time_variant.py $R/$lib > $OUT/out.txt 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
  python3 - $OUT $T <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "binned_kernel" in k:
        print(sys.argv[2], " ".join("%s=%.1fM" % (n.replace("SQ_", ""), sum(v) / len(v) / 1e6) for n, v in sorted(c.items())))
PY
done
