cd $GRAFT_REPO_ROOT
for p in 0 25000 50000; do echo "DKT_PAD_FE_FWD=$p"; DKT_TWINS=force DKT_PAD_FE_FWD=$p python tools/glue_probe.py trunk 2>&1 | grep "from trunk"; done
