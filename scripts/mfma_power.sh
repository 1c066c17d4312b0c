#!/bin/bash
./tools/mfma_power 40000 30 &
sleep 2
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk"; sleep 1; done
wait
