#!/bin/bash
# A/B harness: run bench.py (and the sigma-only kernel) against variant builds of the library.
#   python -m brdf_nerf_amd.build -DNAME[=V] ...   builds brdf_nerf_amd/build/<tag>/libbrdfnerf_hip.so
#   bash profiles/ab_bench.sh <tag> [<tag> ...]      ("default" = the product library)
mkdir -p gpurun_out
for tag in "$@"; do
  if [ "$tag" = default ]; then unset BRDFNERF_HIP_LIB; else export BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/$tag/libbrdfnerf_hip.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || echo "FAILED $tag"
done
# sigma-only inference kernel for the same variants
for tag in "$@"; do
  if [ "$tag" = default ]; then unset BRDFNERF_HIP_LIB; else export BRDFNERF_HIP_LIB=$PWD/brdf_nerf_amd/build/$tag/libbrdfnerf_hip.so; fi
  echo "$tag $(python profiles/time_sigma.py 2>/dev/null | tail -1)" >> gpurun_out/ab_sigma.txt
done
