#!/bin/bash
# the GPU suite under each diagnostic switch of DESIGN.md section 7 (every alternative path stays parity-green): bash profiles/scripts/toggles.sh
for env in "TTSK_CHAIN_FUSED=0" "TTSK_CHAIN_FUSED=2" "TTSK_TT_MERGE=0" "TTSK_SPARSE_DEDUPE=0" "TTSK_JACOBI_PRECOND=0" "TTSK_STREAM_SMALL=0" "TTSK_SUM_PSI_SPLIT=0" "TTSK_SUM_PSI_SPLIT=1000" "TTSK_SINGLE_STREAM=1" "TTSK_FAST_SOLVES=0"; do
  echo "== $env"
  env $env timeout -k 10 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
done
