#!/bin/bash
# builds the timing-only variants of the forward kernel into build_ab/ (git-ignored; travels to the GPU box)
cd "$(dirname "$0")/../.."
mkdir -p build_ab
for v in 1 2 3 4 5 6 7 8 9; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DIONO_FWD_ABL=$v -Iinclude \
        -o build_ab/libionotomo_fwd_abl$v.so ionotomo_amd/csrc/ionotomo_hip.hip &
done
wait
ls -la build_ab
