#!/bin/bash
cd $GRAFT_REPO_ROOT
export FASTKV_CXXFLAGS=-DFK_STAMP
S=$GRAFT_REPO_ROOT/build_x_stamp
( FASTKV_BUILD_DIR=$S python tools/stamp_arrivals.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06k_arrivals_fmaf.log
( FASTKV_BUILD_DIR=$S FASTKV_CONTRACTION=mfma16 python tools/stamp_arrivals.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06k_arrivals_mfma16.log
tail -n 4 gpurun_out/r06k_arrivals_fmaf.log; tail -n 4 gpurun_out/r06k_arrivals_mfma16.log
