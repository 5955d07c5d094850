#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout 600 python tools/ae_concurrent.py > $OUT/ae_concurrent.txt 2>&1; grep -v amdgpu $OUT/ae_concurrent.txt | cut -c1-200
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/ae_concurrent.py > $OUT/ae_concurrent_q8.txt 2>&1; echo "--- GPU_MAX_HW_QUEUES=8"; grep -v amdgpu $OUT/ae_concurrent_q8.txt | cut -c1-200
