#!/bin/bash
# round 6, GPU call 7: the round's measurement set on the final tree + the whole GPU suite
cd $GRAFT_REPO_ROOT
tools/round_profile.sh r06 > gpurun_out/r06_round_profile.log 2>&1
tail -20 gpurun_out/r06_round_profile.log
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gputest_run.log 2>&1; echo "full suite rc=$?"; tail -3 gpurun_out/r06/gputest_run.log
