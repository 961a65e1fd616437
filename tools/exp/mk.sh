#!/bin/bash
# usage: tools/exp/mk.sh VARIANT [extra hipcc flags]   builds tools/exp/libpcr_hip_<VARIANT>.so from the working tree
V=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -I include -I pcrhpg24_amd/csrc "$@" \
    pcrhpg24_amd/csrc/pcr_api.hip -o tools/exp/libpcr_hip_$V.so
