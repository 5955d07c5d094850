#!/bin/bash
export TMPDIR=/tmp
echo "== product"; timeout 200 python tools/grad_vs_golden.py 2>&1 | grep conv_variant
echo "== prev (before the fused tail)"; LEMO_HIP_LIB=$PWD/lemo_amd/csrc/build_ab/prev.so timeout 200 python tools/grad_vs_golden.py 2>&1 | grep conv_variant
