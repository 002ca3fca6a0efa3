#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -q 2>&1 | tail -5
