#!/bin/bash
# tools/build_pair_variant.sh <name> [-DGM_...]: builds graphmaker_amd/lib/variants/<name>.so with extra defines
set -e
cd "$(dirname "$0")/../graphmaker_amd/csrc"
name=$1; shift
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-function "$@" -shared -o ../lib/variants/$name.so gm_pair.hip gm_pair_bf16.hip gm_pair_generic.hip gm_onehot.hip gm_tower.hip gm_adjacency.hip gm_nodeops.hip gm_gemm.hip gm_attr.hip gm_noise.hip gm_train.hip gm_graphstats.hip
