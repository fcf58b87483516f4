#!/bin/bash
# Longer-running evidence on the GPU box (one gpurun call): steady-state rate over 200 steps, randomized stress with several seeds,
# LDS counters of the kernels that park state in LDS.   gpurun --timeout 1100 -- 'tools/gpu_soak.sh r03_soak'
set -e -o pipefail
TAG=${1:-soak}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_200_steps.json 2> $O/bench_200.err
python3 -c "import json; d=json.load(open('$O/bench_200_steps.json')); print('200 steps:', round(d['value']), d['unit'], round(d['ms_per_step'],2), 'ms/step')"
for i in 1 2 3; do timeout -k 10 200 python3 $R/tools/gpu_stress.py > $O/stress_$i.txt 2>&1 || { tail -5 $O/stress_$i.txt; exit 1; }; tail -1 $O/stress_$i.txt; done
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $O/lds_groth16 -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload groth16 --batch-log2 18 > /dev/null 2> $O/lds_groth16.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $O/lds_ref_opt -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload ref_opt --batch-log2 16 > /dev/null 2> $O/lds_ref_opt.err
echo done
