#!/bin/bash
# A/B two builds of libmpformer_hip.so on the same GPU box, interleaved: usage tools/ab_lib.sh <libA.so> <libB.so> <python script + args>
A=$1; B=$2; shift 2
for i in 1 2 3; do
  echo "--- A ($A) run $i"; MPF_LIB_PATH=$A python "$@" 2>&1 | grep -v amdgpu.ids
  echo "--- B ($B) run $i"; MPF_LIB_PATH=$B python "$@" 2>&1 | grep -v amdgpu.ids
done
