#!/bin/bash
# usage (GPU box): tools/exp_rolling_lengths.sh -- ops.scores of 8 / 16 entries at several prompt lengths, rolling launch off / on
for S in 8192 12000 16384 20000 24576 32768; do
  for m in 0 1; do
    EXP_S=$S FASTKV_FUSED_ROLLING=$m timeout 200 python tools/exp_interleave.py | grep "B=8\|B=16" | sed "s/^/S=$S /"
  done
done
