#!/bin/bash
# Collect the rocprofv3 evidence for one workload on the GPU box (run via gpurun from the repo root):
#   tools/collect_profile.sh <tag> [bench.py flags, e.g. --workload groth16 --batch-log2 20]
#       -> gpurun_out/prof_<tag>/{stats,fetch,write,sq}/..., then
#   python tools/summarize_profile.py <tag> [workload@log2]   (here) copies the summaries into profiles/<tag>/ and records the
#   HBM traffic in profiles/latest_traffic.json, which bench.py reports as roofline.traffic.
# Counter passes are separate runs with --pmc only (never combined with trace domains other than kernel-trace); the program
# stands directly behind `--` (no env / shell hop).
set -e
TAG=${1:-r01}
shift || true
FLAGS="$@"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$FLAGS" > $OUT/bench_flags.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $FLAGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $FLAGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $FLAGS > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $FLAGS > /dev/null 2> $OUT/sq.err
python3 $R/bench.py --steps 3 --warmup 1 $FLAGS > $OUT/bench_unprofiled.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench_unprofiled.json
