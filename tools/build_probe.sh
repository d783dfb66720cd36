#!/bin/bash
# Builds tools/cu_probe (gfx950 pipe-interference probe; see the header of cu_probe.hip).  The binary is git-ignored
# but travels with the gpurun snapshot:   bash tools/build_probe.sh && gpurun -- 'timeout 120 tools/cu_probe'
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 cu_probe.hip -o cu_probe
echo "built tools/cu_probe"
