#!/bin/bash
rm -f gpurun_out/r3_probe8.log
for cfg in "3 28 128 1" "2 56 64 1" "3 20 96 1" "5 28 480 1" "3 56 224 1" "128 28 128 1" "128 28 480 1" "128 56 64 1" "128 56 224 1"; do
  timeout -k 10 200 build/block_probe $cfg 2>&1 | grep -v "cycles per layer" >> gpurun_out/r3_probe8.log || echo "probe rc=$? ($cfg)" >> gpurun_out/r3_probe8.log
done
cat gpurun_out/r3_probe8.log
