#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py, one counter group per run (rocprofv3 does not
# split counters over passes; limits per run: 8 SQ, 4 TCC [FETCH_SIZE=3, WRITE_SIZE=2], 4 TCP,
# 2 TA, 2 GRBM).  Usage (on the GPU box, via gpurun):
#   bash tools/pmc_passes.sh <out-subdir-under-gpurun_out> [bench.py args...]
# Then: python tools/pmc_summarize.py gpurun_out/<subdir> > profiles/rNN/<name>.json
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$R/bench.py" $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -3 "$OUT/$name.err"; return 1; }
  echo "pass $name ok"
}
run fetch FETCH_SIZE &&
run write WRITE_SIZE TCC_EA0_ATOMIC_sum &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum &&
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE &&
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR
