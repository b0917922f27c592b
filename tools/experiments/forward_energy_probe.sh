#!/bin/bash
# MEASUREMENT ONLY (wrong results by construction): the cfg2 Gram forward with a third of its MFMAs (libdkt_hip_one.so: -DDKT_PROBE_ONE_PRODUCT in dkt_gram_ep.hip) and without
# the low-plane arithmetic on top (libdkt_hip_nosplit.so) against the product, alternating processes on one box -- how much of the forward's time at the power cap is MFMA / VALU.
# The two libraries are built by:  _lib.build(out=..., replace={'dkt_gram_ep.hip': <copy of the source with the #define on top>})
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for lib in libdkt_hip.so libdkt_hip_one.so libdkt_hip_nosplit.so; do
    DKT_AMD_LIB=$GRAFT_REPO_ROOT/deep-kernel-transfer_amd/$lib python tools/experiments/lib_ab_gram.py 2>&1 | grep forward
  done
done
