#!/bin/bash
# kernel trace of the H/14 16-frame 48-pair step with e4m3 forward + input-gradient + weight-gradient GEMMs (tools/profile_step.sh)
cd $GRAFT_REPO_ROOT
tools/profile_step.sh ${1:-r04}_h14fp8w --arch H_14 --frames 16 --batch 48 --fp8-wgrad > gpurun_out/profile_h14fp8w.log 2>&1
cp gpurun_out/prof_${1:-r04}_h14fp8w/summary.txt gpurun_out/kernel_summary_h14_b48_fp8_wgrad.txt
head -30 gpurun_out/kernel_summary_h14_b48_fp8_wgrad.txt
