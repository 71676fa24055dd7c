#!/bin/bash
# Per-phase times of the grouped decode kernel.  Build step (works without a GPU): decode.hip with -DI2L_GROUP_STAMPS
# linked into a SEPARATE library under csrc/build/ (the product library stays untouched); run step (GPU box):
#   bash profiles/run_stamps.sh build && gpurun -- 'python profiles/decode_stamps.py'
set -e
cd "$(dirname "$0")/../hmer-img2latex_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DI2L_GROUP_STAMPS -c decode.hip -o build/decode_stamps.o
OBJS=$(ls build/*.o | grep -v "build/decode.o" | grep -v decode_stamps.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libimg2latex_hip_stamps.so $OBJS build/decode_stamps.o
echo built build/libimg2latex_hip_stamps.so
