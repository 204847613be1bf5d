#!/bin/bash
# developer iteration: recompile only csrc/aogym.hip (the main translation unit) and relink against the cached fused-kernel objects
set -e
C=$(cd "$(dirname "$0")/../adaptive_optics_gym_amd/csrc" && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -fno-slp-vectorize $AOG_EXTRA -c $C/aogym.hip -o $C/build/aogym.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/../libaogym.so $C/build/aogym.o $C/build/fused_apad16.o $C/build/fused_apad32.o $C/build/fused_apad64.o $C/build/fused_apad128.o -lhipfft
echo linked
