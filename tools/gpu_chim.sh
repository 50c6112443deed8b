#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_chimera_gpu.py -x -q 2>&1 | tail -15
timeout 600 python tools/microbench.py chimera 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin)['chimera']; print('full', round(d['ms'],2), d['split_frac'], d['multi_frac'], d['overflow'])"
