#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s -k "folded_into or implicit_convolution_paths" 2>&1 | grep -E "bf16x6|passed|failed|assert|Error" | tail -12
