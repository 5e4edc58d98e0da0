#!/bin/bash
# Same-box comparison of environment knobs under one bench.py workload: the baseline run is repeated
# between the variants (boxes differ by a few per cent and drift by a few tenths).
#   ab_env.sh TAG "bench.py args" "VAR=val [VAR=val]" ["VAR=val" ...]
TAG=$1; ARGS=$2; shift 2
run() { env "$@" timeout -k 10 280 python3 -u bench.py $ARGS --out gpurun_out/abenv_$TAG.json > /dev/null 2>&1 || exit 1
        python3 -c "import json,sys; d=json.load(open('gpurun_out/abenv_$TAG.json')); print('%-40s %8.3f %9.2f' % (' '.join(sys.argv[1:]) or 'baseline', d['value'], d['ms_per_step']))" "$@"; }
run LC_NOOP=1
for v in "$@"; do run $v; run LC_NOOP=1; done
