#!/bin/bash
# Three hardware-counter passes over benchmarks/pmc_kernels.py (one counter group each: the
# profiler does not split a request over passes), folded into gpurun_out/pmc_hot_kernels.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rm -f gpurun_out/pmc_hot_kernels.json
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc1 -- python3 benchmarks/pmc_kernels.py > gpurun_out/pmc1.log 2>&1
python3 benchmarks/pmc_kernels.py --summarise /tmp/pmc1 gpurun_out/pmc_hot_kernels.json > /dev/null
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d /tmp/pmc2 -- python3 benchmarks/pmc_kernels.py > gpurun_out/pmc2.log 2>&1
python3 benchmarks/pmc_kernels.py --summarise /tmp/pmc2 gpurun_out/pmc_hot_kernels.json > /dev/null
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc3 -- python3 benchmarks/pmc_kernels.py > gpurun_out/pmc3.log 2>&1
python3 benchmarks/pmc_kernels.py --summarise /tmp/pmc3 gpurun_out/pmc_hot_kernels.json
