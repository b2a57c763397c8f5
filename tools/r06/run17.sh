#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06sb
tools/small_batch_profile.sh r06sb > gpurun_out/r06sb/log.txt 2>&1
head -24 gpurun_out/r06sb/kernel_summary_b12.txt | cut -c1-170
