#!/bin/bash
# Samples the shader clock while LocalBA solves run (are the small serial kernels running at idle clocks?)
python tools/lba_prof.py 400 > gpurun_out/clock_lba.log 2>&1 &
PID=$!
sleep 3
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4; echo ---; sleep 0.5; done
wait $PID
cat gpurun_out/clock_lba.log | tail -1
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
rocm-smi --showperflevel 2>/dev/null | grep -i perf | head -2
