#!/bin/bash
# round-3 GPU session 19: where do parameters go non-finite under graph replay? (config x dtype, graph + in-kernel gradient clearing)
for cd in "hapke fp16" "hapke fp16" "hapke bf16" "microfacet fp16" "rpv_nan fp16" "hapke fp32"; do
  set -- $cd
  timeout -k 10 200 python profiles/debug_nan_hapke.py $1 $2 400 --graph > gpurun_out/debug_nan_$1_$2.txt 2>&1
  echo "== $cd: $(grep -v amdgpu gpurun_out/debug_nan_$1_$2.txt | grep -v 'finite g True p True' | grep '^step\|^done' | head -2 | cut -c1-200 | tr '\n' '|')"
done
