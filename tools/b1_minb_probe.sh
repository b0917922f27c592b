#!/bin/bash
# Training step (cfg2 shape) at a handful of episodes per step, eager and replayed from a hipGraph (tools/b1_step_trace.py).  Measurement tooling.
#   tools/b1_minb_probe.sh [MINB ...]    (DKT_GRAM_EP_MINB values to compare; default: the library's)
cd $GRAFT_REPO_ROOT
for b in 1 2 4 7 8 16 48 64; do
  for mb in ${@:-default}; do
    echo "== B=$b DKT_GRAM_EP_MINB=$mb"
    if [ "$mb" = default ]; then python tools/b1_step_trace.py $b 2>&1 | tail -2; else DKT_GRAM_EP_MINB=$mb python tools/b1_step_trace.py $b 2>&1 | tail -2; fi
  done
done
