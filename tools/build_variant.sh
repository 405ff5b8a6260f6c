#!/bin/bash
# build_variant.sh NAME "extra flags": a copy of libmde_hip.so with other compile-time knobs of the LDS-ring
# kernel (layout builder, dispatcher and the Log1p / PushAndPull units are recompiled with the flags)
# (design experiments on the GPU box: LD_LIBRARY_PATH=tools/variants/NAME ./tools/kbench ...)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants/$1
units="mde_ring mde_ring_k_log1p mde_ring_k_pushpull mde_ring_k_penalty mde_ring_k_loss"
for u in $units; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -Wno-unused-result -ffp-contract=fast $2 -c pymde_amd/csrc/$u.hip -o tools/variants/$1/$u.o &
done
wait
objs=$(ls pymde_amd/csrc/build/*.o | grep -v -e "mde_ring.o" -e "mde_ring_k_log1p.o" -e "mde_ring_k_pushpull.o" -e "mde_ring_k_penalty.o" -e "mde_ring_k_loss.o")
vobjs=$(for u in $units; do echo tools/variants/$1/$u.o; done)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$1/libmde_hip.so $objs $vobjs
