#!/bin/bash
# an experiment's build of the library beside the product's:  tools/build_variant.sh <name> [-DMACRO=value ...]
# -> _ab/libmegakv_hip_<name>.so (git-ignored; travels to the GPU box), for tools/gpu_ab_bench.sh
name=$1; shift
cd "$(dirname "$0")/../megakv_amd/csrc" || exit 1
make -s -j8 so EXTRA="$*" OBJDIR=../_lib/obj_var_$name SOOUT=../../_ab/libmegakv_hip_$name.so
