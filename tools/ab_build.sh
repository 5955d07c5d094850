#!/bin/bash
# A/B build of the HIP library with extra -D flags: tools/ab_build.sh NAME -DFLAG... -> lemo_amd/csrc/build_ab/NAME.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../lemo_amd/csrc"; mkdir -p build_ab/$NAME
for f in conv_kernels conv_split_kernels conv_pair_kernels conv_wino_kernels conv_head_kernels gemm_kernels pose_kernels lbs_kernels loss_kernels scene_kernels ae_kernels ae_engine marker_kernels prox_kernels lemo_prox lemo_hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-vectorize -DLEMO_NO_PACKED_FP32 -I../../include -Wno-unused-function "$@" -c $f.hip -o build_ab/$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/$NAME.so build_ab/$NAME/*.o
echo built build_ab/$NAME.so
