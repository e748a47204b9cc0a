#!/bin/bash
# GPU box: attention parity tests with the default backward form, then the standalone timing of the three forms
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attn or attention" > gpurun_out/t_attn.log 2>&1; tail -3 gpurun_out/t_attn.log
for v in 0 1 2; do echo "NKB_ATTN_BWD_PAIR=$v"; NKB_ATTN_BWD_PAIR=$v python scripts/attn_microbench.py; done
