#!/bin/bash
# Rebuild only the host-side C++ of both libraries (vc_host / vc_windows / vc_io) against the device objects that are already built.
set -e
cd "$(dirname "$0")/.."
L=vechat_amd/lib; C=vechat_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include $L/obj/vc_api.o $L/obj/vc_align.o $C/vc_host.cpp $C/vc_windows.cpp $C/vc_io.cpp -o $L/libvechat_hip.so -lpthread -lz
g++ -std=c++17 -O2 -fPIC -shared -Wall -Wextra -I include $C/vc_host.cpp $C/vc_windows.cpp $C/vc_io.cpp -o $L/libvechat_host.so -lpthread -lz
