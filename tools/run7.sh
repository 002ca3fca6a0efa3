#!/bin/bash
export TMPDIR=/tmp
timeout 400 python tools/decode_ab.py --efforts 0.25 --split 1 2>/dev/null
timeout 300 python tools/decode_ab.py --efforts 0.25 --fused-glue 1 2>/dev/null
