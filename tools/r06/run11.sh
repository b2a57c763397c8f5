#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
experiments/probes/cu_store > gpurun_out/r06/cu_store_probe.txt 2>&1; cat gpurun_out/r06/cu_store_probe.txt
