#!/bin/bash
# Kernel statistics of one bench.py workload under rocprofv3:
#   prof_bench.sh WORKLOAD OUTDIR [extra bench.py args]
# Leaves OUTDIR/kernel_stats.{txt,json}, OUTDIR/kernel_medians.txt and OUTDIR/bench.json; the raw
# per-dispatch trace (tens of MB) is summarised and removed.
W=${1:-A}; OUT=${2:-gpurun_out/prof_$W}; shift 2
mkdir -p "$OUT"; ROOT=$PWD; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 bench.py --workload "$W" --steps 3 --warmup 1 "$@" > "$OUT/bench.out" 2> "$OUT/bench.err" || exit $?
grep '^{' "$OUT/bench.out" | tail -1 > "$OUT/bench.json"
python3 benchmarks/prof_summary.py "$OUT/trace" 40 "$OUT/kernel_stats.json" > "$OUT/kernel_stats.txt"
python3 benchmarks/prof_medians.py "$OUT/trace" 40 > "$OUT/kernel_medians.txt"
rm -rf "$OUT/trace"
head -30 "$OUT/kernel_stats.txt"
