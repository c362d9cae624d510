cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_mlp_fused.py -m gpu -x -q 2>&1 | tail -1
for i in 1 2 3; do
for lib in libduodiff_noprojpair.so libduodiff.so; do
echo "$lib $(DUODIFF_LIB=duodiff_amd/$lib timeout -k 10 200 python tools/mlp_unit.py --D 512 --M 32768 --proj --iters 40 2>&1 | tail -1 | grep -o 'max|err|=[0-9.e+-]*\|[0-9.]* us/launch' | tr '\n' ' ')"
done; done
