#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_train.py -q 2>&1 | tail -3
