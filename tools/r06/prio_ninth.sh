#!/bin/bash
# round 6, ninth contact: issue priority of the rolling launch's phases under the fma chain (experiments build, FASTKV_FUSED_TUNE bits:
# 1 = phase A raised (the default), 4 = the phases behind phase A raised, 0 = none)
cd $GRAFT_REPO_ROOT
X=$GRAFT_REPO_ROOT/build_x_exp
out=gpurun_out/r06i_prio_ab.log
: > $out
for c in fmaf mfma16; do for i in 1 2; do for t in 1 0 4; do
  echo -n "$c TUNE=$t :: " >> $out
  FASTKV_CONTRACTION=$c FASTKV_BUILD_DIR=$X FASTKV_FUSED_TUNE=$t python tools/exp_interleave.py 2>&1 | grep "B=8\|B=16" | tr '\n' ' ' >> $out; echo >> $out
done; done; done
cat $out
