#!/bin/bash
# GPU box: tests of the head kernels, then the step with the first layer's weight gradient in its linear / fused form
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -x -q -k "head or bwd16 or border or dgrad" > gpurun_out/t_head_ops.log 2>&1 || { tail -30 gpurun_out/t_head_ops.log; exit 1; }
tail -2 gpurun_out/t_head_ops.log
timeout -k 10 800 python -m pytest tests/test_model_gpu.py tests/test_bf16_gpu.py tests/test_switches_gpu.py -x -q > gpurun_out/t_head_model.log 2>&1 || { tail -30 gpurun_out/t_head_model.log; exit 1; }
tail -2 gpurun_out/t_head_model.log
for i in 1 2 3; do
  for v in 1 0; do
    echo -n "HEAD_LINEAR=$v: "; SIFSR_HEAD_LINEAR=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo --no-also 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['ms_per_step_median'])"
  done
done
for v in 1 0; do echo -n "bf16 HEAD_LINEAR=$v: "; SIFSR_HEAD_LINEAR=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo --no-also --dtype bf16 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done
