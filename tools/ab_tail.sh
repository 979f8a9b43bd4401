#!/bin/bash
# GPU box: tests of the tail kernels, then their stand-alone and in-step timings
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q -k "bwd16 or conv_out or tail or edges or thin" > gpurun_out/t_tail_ops.log 2>&1 || { tail -30 gpurun_out/t_tail_ops.log; exit 1; }
tail -2 gpurun_out/t_tail_ops.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/t_tail_model.log 2>&1 || { tail -30 gpurun_out/t_tail_model.log; exit 1; }
tail -2 gpurun_out/t_tail_model.log
bash tools/ab_tail_reduce.sh
for i in 1 2; do
    echo -n "fp32: "; python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo --no-also 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['ms_per_step_median'])"
    echo -n "bf16: "; python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo --no-also --dtype bf16 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['ms_per_step_median'])"
done
