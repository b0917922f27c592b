cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in hip p20 p02 p00; do
  echo "== $v"; DKT_AMD_LIB=$GRAFT_REPO_ROOT/deep-kernel-transfer_amd/libdkt_$v.so python tools/check_band.py time 2>&1 | grep "B=1024 C=20\|B=1024 C=32"
done; done
