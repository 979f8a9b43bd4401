#!/bin/bash
# GPU box, end of a round: the full -m gpu suite, then the profile set the DESIGN / profiles README cite (tag r03z)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03z_tests.log 2>&1 || { tail -20 gpurun_out/r03z_tests.log; exit 1; }
tail -2 gpurun_out/r03z_tests.log
bash tools/profile_round.sh r03z && bash tools/prof_stats.sh r03z_single single && bash tools/prof_stats_bf16.sh r03z_bf16 && bash tools/prof_stats_bf16.sh r03z_bf16_single single && bash tools/pmc_mfma.sh r03z > gpurun_out/r03z_mfma_util.txt 2>&1 && bash tools/pmc_conv.sh r03z_pmc_bwd16 bwd16 16 16 256 64 10 > gpurun_out/r03_pmc_bwd16.txt 2>&1 && echo profiles-done
python bench.py > gpurun_out/r03z_bench_train.json 2> gpurun_out/r03z_bench_train.err && tail -c 600 gpurun_out/r03z_bench_train.json
