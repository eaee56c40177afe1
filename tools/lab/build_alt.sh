#!/bin/bash
# Build esm-efficient_amd/esme/libesme_hip_alt.so with ONE source file replaced (A/B experiments on the GPU box:
# ESME_HIP_LIB=.../libesme_hip_alt.so python tools/attn_lab.py).  usage: build_alt.sh attn.hip /path/to/variant.hip
set -e
cd "$(dirname "$0")/../../esm-efficient_amd/csrc"
name=$1; variant=$2
mkdir -p build_alt
flags="$ALT_DEFS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I."
objs=""
for f in api.hip rowops.hip gemm.hip attn.hip quant.hip forward.hip; do
  if [ "$f" == "$name" ]; then
    cp "$variant" build_alt/variant_$f
    /opt/rocm/bin/hipcc $flags -c build_alt/variant_$f -o build_alt/${f%.hip}.o
    objs="$objs build_alt/${f%.hip}.o"
  else
    objs="$objs build/${f%.hip}.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../esme/libesme_hip_alt.so $objs
echo built ../esme/libesme_hip_alt.so
