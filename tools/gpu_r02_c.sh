#!/bin/bash
set -u
mkdir -p gpurun_out/r02c
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
