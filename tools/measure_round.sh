#!/bin/bash
# One GPU call's worth of round measurements (GPU box, via gpurun): bash tools/measure_round.sh <tag>
# bench.py on every workload, rocprofv3 kernel stats, PMC passes for the headline shapes, the training harness.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
cd "$R"
bash tools/bench_all.sh ${TAG}_bench || exit 1
bash tools/kernel_stats.sh ${TAG}_ks_C3_f32 > /dev/null || exit 1
bash tools/kernel_stats.sh ${TAG}_ks_C3_bf16 --dtype bf16 > /dev/null || exit 1
bash tools/kernel_stats.sh ${TAG}_ks_C2_uniform --workload C2 --dist uniform > /dev/null || exit 1
bash tools/kernel_stats.sh ${TAG}_ks_C5 --workload C5 > /dev/null || exit 1
bash tools/pmc_passes.sh ${TAG}_pmc_C3_f32 || exit 1
python tools/pmc_summarize.py gpurun_out/${TAG}_pmc_C3_f32 > gpurun_out/${TAG}_pmc_C3_f32.json
bash tools/pmc_passes.sh ${TAG}_pmc_C3_bf16 --dtype bf16 || exit 1
python tools/pmc_summarize.py gpurun_out/${TAG}_pmc_C3_bf16 > gpurun_out/${TAG}_pmc_C3_bf16.json
for mode in "" "--all-trainable" "--fused"; do
  name=train_$(echo "${mode:-adapters}" | tr -d '-')
  timeout -k 10 200 python train_bench.py --image 1024 --global-batch 16 --steps 10 --warmup 3 $mode > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { tail -3 gpurun_out/${TAG}_$name.err; exit 1; }
  cat gpurun_out/${TAG}_$name.json
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_ks_train" -o run -- \
  python3 "$R/train_bench.py" --image 1024 --global-batch 16 --steps 5 --warmup 2 > /dev/null 2>&1 || exit 1
echo done
