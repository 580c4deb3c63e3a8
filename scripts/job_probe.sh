#!/bin/bash
rm -f gpurun_out/r3_probe6.log
for exe in build/block_probe_v3 build/block_probe_np; do
for cfg in "128 14 256 24" "128 7 512 16"; do
  echo "== $exe" >> gpurun_out/r3_probe6.log
  timeout -k 10 120 $exe $cfg >> gpurun_out/r3_probe6.log 2>&1 || echo "probe rc=$? ($cfg)" >> gpurun_out/r3_probe6.log
done; done
cat gpurun_out/r3_probe6.log
