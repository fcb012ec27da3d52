#!/bin/bash
# development: build libvechat_hip.so with extra -D flags into vechat_amd/lib/variants/ for A/B runs on the GPU box
#   tools/build_variant.sh NAME [-DVC_TILE=1 ...];  run with VECHAT_HIP_LIB=vechat_amd/lib/variants/libvechat_hip_NAME.so
#   VC_PLAIN_CFG=1 tools/build_variant.sh plain     the library WITHOUT -structurizecfg-skip-uniform-regions (tools/gpu_flag_parity.sh)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p vechat_amd/lib/variants
FAST="-mllvm -structurizecfg-skip-uniform-regions"
[ "$VC_PLAIN_CFG" = "1" ] && FAST=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $FAST -I include "$@" \
  vechat_amd/csrc/vc_api.hip vechat_amd/csrc/vc_align.hip vechat_amd/csrc/vc_host.cpp vechat_amd/csrc/vc_windows.cpp vechat_amd/csrc/vc_io.cpp \
  -lz -o vechat_amd/lib/variants/libvechat_hip_$name.so
echo built $name
