#!/bin/bash
# Builds variants of libcmhsa_hip.so for A/B timing:  tools/ab_build.sh NAME "-DFLAG ..." [NAME2 "-DFLAGS2" ...]
# -> tools/ab/NAME/libcmhsa_hip.so ; run with  LD_LIBRARY_PATH=tools/ab/NAME tools/cmhsa_bench ...
set -e
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  mkdir -p tools/ab/$1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $2 -Iinclude -o tools/ab/$1/libcmhsa_hip.so \
      llm-inference_amd/csrc/cmhsa_fwd.hip llm-inference_amd/csrc/cmhsa_decode.hip llm-inference_amd/csrc/gelu_fwd.hip llm-inference_amd/csrc/add_layernorm_fwd.hip llm-inference_amd/csrc/gpt2_step.hip &
  shift 2
done
wait
ls -la tools/ab/*/libcmhsa_hip.so
