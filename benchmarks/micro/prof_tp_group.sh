#!/bin/bash
# Kernel trace of rank 0 of benchmarks/tp_group_decode_bench.py (the other ranks run unprofiled):
#   prof_tp_group.sh PRESET TP BATCH CTX OUTDIR
PRESET=${1:-llama-3-70b}; TP=${2:-4}; B=${3:-64}; CTX=${4:-512}; OUT=${5:-gpurun_out/prof_tpg}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
export WORLD_SIZE=$TP MASTER_ADDR=127.0.0.1 MASTER_PORT=$((20000 + RANDOM % 20000)) HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for r in $(seq 1 $((TP-1))); do
  RANK=$r LOCAL_RANK=$r timeout -k 5 400 python3 benchmarks/tp_group_decode_bench.py --worker $PRESET $TP $B $CTX "$OUT/bench.json" > "$OUT/rank$r.out" 2>&1 &
  pids+=($!)
done
export RANK=0 LOCAL_RANK=0
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 benchmarks/tp_group_decode_bench.py --worker $PRESET $TP $B $CTX "$OUT/bench.json" > "$OUT/rank0.out" 2>&1
rc=$?
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
echo "rc=$rc"; tail -2 "$OUT/bench.json.rank0.log"
python3 benchmarks/prof_summary.py "$OUT/trace" 24 "$OUT/kernel_stats.json" | tee "$OUT/kernel_stats.txt"
exit $rc
