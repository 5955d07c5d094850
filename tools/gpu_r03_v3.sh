#!/bin/bash
TAG=${1:-r03c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/race_hunt.py 120 B > $OUT/race_hunt_B.txt 2>&1; tail -30 $OUT/race_hunt_B.txt | cut -c1-900
timeout 300 python tools/race_hunt.py 60 torch > $OUT/race_hunt_torch.txt 2>&1; tail -12 $OUT/race_hunt_torch.txt | cut -c1-900
