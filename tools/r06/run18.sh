#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -5
