#!/bin/bash
# Round-5 A/B visit: infilling-AE finetune, weight-gradient slab length (LEMO_AE_SLAB_SCALE x 576 padded pixels) with 8 and 16 clips per engine
TAG=${1:-r05abae}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  for S in 1 2 4 8 16; do LEMO_AE_SLAB_SCALE=$S timeout 120 python tools/ae_prof_k8.py 8 2>&1 | grep "clips per engine" | sort | head -n 1; done
done
for S in 1 4 8 16; do LEMO_AE_SLAB_SCALE=$S timeout 120 python tools/ae_prof_k8.py 16 2>&1 | grep "clips per engine" | sort | head -n 1; done
timeout 120 python tools/ae_prof_k8.py 8 2>&1 | grep "clips per engine" | sort | head -n 1
timeout 120 python tools/ae_prof_k8.py 2 2>&1 | grep "clips per engine" | sort | head -n 1
LEMO_AE_SLAB_SCALE=1 timeout 120 python tools/ae_prof_k8.py 2 2>&1 | grep "clips per engine" | sort | head -n 1
