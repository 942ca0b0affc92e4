for i in 1 2 3; do
for lib in build/ab/libdsac_hip_zacc8.so ../libdsac_hip.so; do
DSAC_HIP_LIB=$GRAFT_REPO_ROOT/dsac_amd/csrc/$lib DSAC_AB_MODES=exact DSAC_AB_ROUNDS=3 DSAC_AB_ROUNDS2=0 timeout 300 python scripts/r06_k2_exact_ab.py 2>&1 | grep "auto policy" | awk -v l=$lib '{s+=$(NF-5); n++} END {print l, s/n}'
done; done
