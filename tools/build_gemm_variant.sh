#!/bin/bash
# tools/build_gemm_variant.sh <name> <extra hipcc flags...> -- exploration builds of the library that differ only in gm_gemm.hip
# (A/B builds of a modified source tree); output graphmaker_amd/lib/libgm_<name>.so for tools/gemm_ab.py.  Never the product library.
set -e
cd "$(dirname "$0")/../graphmaker_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-function "$@" -c gm_gemm.hip -o /tmp/gm_gemm_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libgm_$name.so gm_tower.o gm_adjacency.o gm_nodeops.o /tmp/gm_gemm_$name.o gm_onehot.o gm_pair.o gm_pair_bf16.o gm_pair_generic.o gm_attr.o gm_noise.o gm_train.o gm_graphstats.o
