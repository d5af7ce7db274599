#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ (run on the GPU box through gpurun); the
# summaries are then copied into profiles/ (tracked) by scripts/collect_profiles.py.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-ceilings --no-extras"
# per-kernel time (kernel trace + stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $REPO/bench.py $ARGS > $OUT/prof_${TAG}_bench.log 2>&1
# HBM traffic: counters in passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass), no tracing
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ceilings --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ceilings --no-extras > /dev/null 2>&1
# the three sparse products share a kernel template: one pass per counter and product
for OP in J_x JT_y H_sym_x; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_spmv_fetch_${OP}_$TAG -- python3 $REPO/bench.py --spmv-only $OP > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_spmv_write_${OP}_$TAG -- python3 $REPO/bench.py --spmv-only $OP > /dev/null 2>&1
done
cd $REPO
python scripts/pmc_summary.py $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_summary.txt 2>&1
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
python scripts/pmc_spmv_summary.py $OUT $TAG $OUT/${TAG}_pmc_spmv.json >> $OUT/${TAG}_pmc_summary.txt 2>&1
cp $OUT/${TAG}_pmc_spmv.json $REPO/profiles/${TAG}_pmc_spmv.json
# the bench line below reads roofline.traffic from profiles/ (and checks the kernel sources' hash recorded in it)
cp $OUT/${TAG}_pmc_traffic.json $REPO/profiles/${TAG}_pmc_traffic.json
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --workload uniform_n1e4_m5e3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_config3.json 2>> $OUT/${TAG}_bench.err
# config 3: per-kernel stats (the largest k_front_schur launch is the MFMA-bound one: MaxNs) and the matrix-pipe counters
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3_$TAG -- python3 $REPO/bench.py --workload uniform_n1e4_m5e3 --steps 10 --warmup 2 --no-cpu-baseline --no-ceilings --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_c3_$TAG -- python3 $REPO/bench.py --workload uniform_n1e4_m5e3 --steps 3 --warmup 1 --no-cpu-baseline --no-ceilings --no-extras > /dev/null 2>&1
cd $REPO
find $OUT/prof_c3_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats_config3.csv
python scripts/pmc_mfma_summary.py $OUT/pmc_mfma_c3_$TAG > $OUT/${TAG}_pmc_mfma_config3.txt 2>&1
