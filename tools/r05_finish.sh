#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in "" recipe; do
for m in 0 1 0 1; do
  FASTKV_FINISH_STREAM=$m timeout 300 python tools/exp_finish_stream.py $r 2>&1 | grep -E "ms per step|Error|error|Traceback" | head -5
done; done 2>&1 | tee gpurun_out/r05_finish_stream.log
