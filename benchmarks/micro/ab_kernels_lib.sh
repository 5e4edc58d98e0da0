#!/bin/bash
# Same-box A/B/A of two builds of the kernel library under one bench.py workload:
#   ab_kernels_lib.sh ALT_LIB.so TAG [bench.py args]
# runs the workload with the in-tree library, with ALT_LIB copied over it, and with the in-tree
# library again (boxes differ by a few per cent, so only runs on one box compare).
ALT=$1; TAG=$2; shift 2
LIB=llm_consensus_amd/_native/liblc_kernels.so
run() { timeout -k 10 280 python3 -u bench.py "$@" > /dev/null 2>&1; }
cp "$LIB" /tmp/main_lib.so || exit 1
run "$@" --out gpurun_out/ab_${TAG}_main1.json || exit 1
cp "$ALT" "$LIB" && run "$@" --out gpurun_out/ab_${TAG}_alt.json; rc=$?
cp /tmp/main_lib.so "$LIB"; [ $rc -eq 0 ] || exit $rc
run "$@" --out gpurun_out/ab_${TAG}_main2.json || exit 1
python3 - "$TAG" <<'PY'
import json, sys
for k in ("main1", "alt", "main2"):
    d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}_{k}.json"))
    print(k, d["value"], d["ms_per_step"])
PY
