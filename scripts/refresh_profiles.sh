#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ (run on the GPU box through gpurun); the
# summaries are then copied into profiles/ by hand (see profiles/README.md).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ceilings > $REPO/gpurun_out/prof_${TAG}_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch_$TAG -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ceilings > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/pmc_write_$TAG -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ceilings > /dev/null 2>&1
cd $REPO
python scripts/pmc_summary.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_summary.txt
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --workload uniform_n1e4_m5e3 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_config3.json 2>> gpurun_out/${TAG}_bench.err
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
