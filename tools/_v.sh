mkdir -p gpurun_out/r03v
timeout 900 python tools/ae_conv_tune.py > gpurun_out/r03v/ae_conv_tune.txt 2>&1; echo rc=$?
cat gpurun_out/r03v/ae_conv_tune.txt
