#!/bin/bash
# Two PMC passes (tcp, tcc) + sq of bench.py for quick A/B of a kernel variant:
#   bash tools/pmc_quick.sh <out-subdir-under-gpurun_out> [bench.py args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$R/bench.py" $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -3 "$OUT/$name.err"; return 1; }
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE &&
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU &&
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR &&
python3 "$R/tools/pmc_summarize.py" "$OUT" > "$OUT/summary.json"
