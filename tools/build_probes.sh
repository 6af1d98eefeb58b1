#!/bin/bash
# builds the stand-alone probes of tools/ into tools/bin/ (git-ignored; the binaries travel to the GPU box with the gpurun snapshot)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tools/bin
for f in $ROOT/tools/*.hip; do
  n=$(basename $f .hip)
  extra=""
  [ $n = clock_probe ] && extra="-L/opt/rocm/lib -lhsa-runtime64"       # hsa_amd_pointer_info: where the loader put the kernel code
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I/opt/rocm/include -I$ROOT/include -I$ROOT/elegantrl_amd/csrc -o $ROOT/tools/bin/$n $f $extra && echo "built tools/bin/$n"
done
