#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "inner_operator or bf16x6" 2>&1 | tail -4
