#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
LEMO_AE_SECOND_STREAM=0 timeout 600 python tools/ae_concurrent.py > $OUT/ae_concurrent_single.txt 2>&1; grep -v amdgpu $OUT/ae_concurrent_single.txt | cut -c1-200
timeout 600 python tools/ae_concurrent.py > $OUT/ae_concurrent_two.txt 2>&1; grep -v amdgpu $OUT/ae_concurrent_two.txt | cut -c1-200
