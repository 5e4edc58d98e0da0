#!/bin/bash
# Starts W ranks of tests/p2p_worker.py on one GPU: run_p2p_worker.sh W LOGDIR
W=${1:-2}; LOGDIR=${2:-gpurun_out/p2p}; mkdir -p "$LOGDIR"
PORT=$((20000 + RANDOM % 20000)); pids=()
for r in $(seq 0 $((W-1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT LC_P2P_LOG_DIR="$LOGDIR" \
    timeout -k 5 ${P2P_TIMEOUT:-120} python -u tests/p2p_worker.py "$LOGDIR/ok$W" > "$LOGDIR/stdout$W.$r.log" 2>&1 &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait "$p" || rc=$?; done
echo "world $W exit $rc"; tail -n 4 "$LOGDIR"/p2p_worker.*.log; tail -n 15 "$LOGDIR/stdout$W.0.log"
exit $rc
