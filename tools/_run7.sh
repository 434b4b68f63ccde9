export MI_EXPECT_GPU=1
time python tools/parity_report.py --frames 16 --height 1000 --width 1500 > gpurun_out/parity_small.json 2> gpurun_out/parity_small.err; tail -3 gpurun_out/parity_small.err; cat gpurun_out/parity_small.json | head -c 3000; echo
time python tools/parity_report.py > gpurun_out/parity_full.json 2> gpurun_out/parity_full.err; tail -12 gpurun_out/parity_full.err; cat gpurun_out/parity_full.json | head -c 6000
