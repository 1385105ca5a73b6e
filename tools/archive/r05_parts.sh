#!/bin/bash
# round 5: eight 32k layers as 16 half-entries (4 KV heads each, four on the chip) or 32 quarter-entries (eight on the chip) of the
# rolling launch instead of 8 entries two at a time: finer interleaving of the K-streaming and the vector phases?
cd $GRAFT_REPO_ROOT
run() { echo -n "$* :: "; env "$@" EXP_B=8 python tools/exp_occ3.py 2>&1 | grep -E "us per call|Error|error" | head -2; }
for i in 1 2; do
run A=1
run FASTKV_FUSED_ROLLING_PARTS=2
run FASTKV_FUSED_ROLLING_PARTS=2 FASTKV_FUSED_STAGGER_US=26
run FASTKV_FUSED_ROLLING_PARTS=2 FASTKV_FUSED_STAGGER_US=40
run FASTKV_FUSED_ROLLING_PARTS=4
run FASTKV_FUSED_ROLLING_PARTS=4 FASTKV_FUSED_STAGGER_US=40
run FASTKV_FUSED_ROLLING_PARTS=4 FASTKV_FUSED_STAGGER_US=80
run FASTKV_FUSED_STAGGER_US=22
done 2>&1 | tee gpurun_out/r05_parts.log
