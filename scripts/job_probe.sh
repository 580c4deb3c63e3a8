#!/bin/bash
rm -f gpurun_out/r3_probe7.log
for cfg in "3 28 128 1" "2 56 64 1" "3 20 96 1" "128 28 128 1" "128 28 480 1" "128 56 64 1" "128 56 224 1" "128 14 256 24" "128 7 512 16"; do
  timeout -k 10 200 build/block_probe $cfg >> gpurun_out/r3_probe7.log 2>&1 || echo "probe rc=$? ($cfg)" >> gpurun_out/r3_probe7.log
done
cat gpurun_out/r3_probe7.log
