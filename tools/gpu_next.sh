#!/bin/bash
# First visit of the NEXT round: the measurements the round-2 notes ask for, every profiler call with a timeout
# and csv output.  gpurun --timeout 900 -- 'bash tools/gpu_next.sh r03a'
TAG=${1:-r03a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
# 1. stage 1: which launches make up a B = 1 iteration (eager launches: one row per kernel; graphs hide nothing here)
cd /tmp && PERFRAME_EAGER=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/pf -o p -- python $R/tools/perframe_prof.py 6 > /dev/null 2>&1
cd $R; find $OUT/pf -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/perframe_kernel_stats.csv; done; rm -rf $OUT/pf
timeout 200 python tools/perframe_concurrent.py 8 8 2>&1 | grep -E "clip\(s\)|built" > $OUT/perframe_concurrent.txt
# 2. AE training step kernel table
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ae -o p -- python $R/tools/ae_prof.py > /dev/null 2>&1
cd $R; find $OUT/ae -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/ae_kernel_stats.csv; done; rm -rf $OUT/ae
# 3. the box: peaks, one / two / three clips
./tools/ubench/peak_ubench > $OUT/peaks.txt 2>&1
timeout 200 python tools/concurrent_clips.py 100 3 2>&1 | grep -E "clip\(s\)" > $OUT/concurrent_clips.txt
# 4. conv census + LBS census
timeout 120 python tools/split_check.py 2>&1 | tail -4 > $OUT/conv_census.txt
timeout 120 python tools/lbs_census.py 2>&1 | tail -2 > $OUT/lbs_census.txt
head -20 $OUT/perframe_kernel_stats.csv | cut -c1-200; cat $OUT/perframe_concurrent.txt $OUT/concurrent_clips.txt $OUT/conv_census.txt $OUT/lbs_census.txt
