#!/bin/bash
# round-3 GPU session 20: amax words cleared by a kernel instead of a memset node - repeated graph-replay runs, fp16 + analytic normals
for cd in "hapke fp16" "rpv_nan fp16" "hapke fp16" "rpv_nan fp16" "microfacet fp16" "hapke fp16" "rpv_nan fp16" "hapke fp16"; do
  set -- $cd
  timeout -k 10 200 python profiles/debug_nan_hapke.py $1 $2 500 --graph > gpurun_out/debug_nan_$1_$2.txt 2>&1
  echo "== $cd: $(grep -v amdgpu gpurun_out/debug_nan_$1_$2.txt | grep -v 'finite g True p True' | grep '^step\|^done' | head -2 | cut -c1-200 | tr '\n' '|')"
done
