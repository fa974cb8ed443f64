#!/bin/bash
# Kernel experiment builds: recompiles ONE kernel source with extra -D flags and links csrc/libqcolloc_hip.<name>.so next to
# the product library (loaded with QCOLLOC_HIP_VARIANT=<name>; never the default).
#   profiles/build_variant.sh <name> <source.hip> "<flags>"
set -e
cd "$(dirname "$0")/../quantumcollocation.jl_amd/csrc"
name=$1; src=$2; flags=$3
obj=${src%.*}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function $flags -c $src -o /tmp/${obj}.${name}.o
objs=$(ls *.o | grep -v "^${obj}\.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libqcolloc_hip.${name}.so $objs /tmp/${obj}.${name}.o -ldl -lpthread
echo built libqcolloc_hip.${name}.so
