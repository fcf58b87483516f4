#!/bin/bash
# One gpurun call's worth of round evidence: GPU tests, then one bench line per workload (outputs under gpurun_out/<tag>/).
#   gpurun --timeout 1100 -- 'tools/gpu_round_check.sh r03a'
set -e -o pipefail
TAG=${1:-r03}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1 || { tail -30 $O/gputest.txt; exit 1; }
tail -2 $O/gputest.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --from-host --single-process > $O/bench_pairing.json 2> $O/bench_pairing.err || { tail -5 $O/bench_pairing.err; exit 1; }
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --workload miller --batch-log2 16 > $O/bench_miller16.json 2> $O/bench_miller16.err || { tail -5 $O/bench_miller16.err; exit 1; }
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --workload ref_opt --batch-log2 16 > $O/bench_ref_opt16.json 2> $O/bench_ref_opt16.err || { tail -5 $O/bench_ref_opt16.err; exit 1; }
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --workload groth16 --batch-log2 20 --single-process > $O/bench_groth16.json 2> $O/bench_groth16.err || { tail -5 $O/bench_groth16.err; exit 1; }
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --workload groth16_prepared --batch-log2 20 > $O/bench_groth16_prepared.json 2> $O/bench_groth16_prepared.err || { tail -5 $O/bench_groth16_prepared.err; exit 1; }
for f in pairing miller16 ref_opt16 groth16 groth16_prepared; do python3 -c "
import json,sys
d=json.load(open('$O/bench_$f.json'))
print('$f', round(d['value']), d['unit'], round(d['ms_per_step'],2),'ms', 'spot', d['config']['parity_spot_check'], d['config']['parity_spot_check_cover'], 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('single_thread'))
"; done
