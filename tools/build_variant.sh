#!/bin/bash
# build_variant.sh NAME "extra flags": a copy of libmde_hip.so with other compile-time knobs of mde_ring.hip
# (design experiments on the GPU box: LD_LIBRARY_PATH=tools/variants/NAME ./tools/kbench ...)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants/$1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -Wno-unused-result -ffp-contract=fast $2 -c pymde_amd/csrc/mde_ring.hip -o tools/variants/$1/mde_ring.o
objs=$(ls pymde_amd/csrc/build/*.o | grep -v mde_ring.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$1/libmde_hip.so $objs tools/variants/$1/mde_ring.o
