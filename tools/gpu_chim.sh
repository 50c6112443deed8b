#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_chimera_gpu.py tests/test_write_gpu.py -x -q 2>&1 | tail -3
SMI_CHIM_GENERIC=1 timeout 900 python -m pytest tests/test_chimera_gpu.py -x -q 2>&1 | tail -2
SMI_CHIM_NO_PREFILTER=1 timeout 900 python -m pytest tests/test_chimera_gpu.py -x -q 2>&1 | tail -2
for a in 0 16 2; do SMI_CHIM_ABLATE=$a timeout 600 python tools/microbench.py chimera 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin)['chimera']; print('ablate $a', round(d['ms'],2), d['split_frac'], d['multi_frac'])"; done
SMI_CHIM_NO_PREFILTER=1 timeout 600 python tools/microbench.py chimera 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin)['chimera']; print('no prefilter', round(d['ms'],2), d['split_frac'], d['multi_frac'])"
