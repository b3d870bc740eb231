#!/bin/sh
# One round of evidence on the GPU box (run through gpurun): the default bench line, rocprofv3 kernel stats of the same
# command, and the three PMC passes (own runs, as the pool requires) on a short run of the same kernels and launch sizes.
#   tools/profile_round.sh TAG        -> gpurun_out/TAG_bench.json, TAG_stats/, TAG_pmc_{fetch,write,sq}/
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 1 --warmup 0 --cpu-rows 0 > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.err
short="--steps 1 --warmup 0 --rows 8192 --trials 8 --num-steps 4 --resample 2 --cpu-rows 0 --no-profile"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py $short > /dev/null 2> gpurun_out/${tag}_pmc.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py $short > /dev/null 2>> gpurun_out/${tag}_pmc.err
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_sq -- python3 bench.py $short > /dev/null 2>> gpurun_out/${tag}_pmc.err
echo done
