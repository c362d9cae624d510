#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=gpurun_out/r5k; rm -rf $out; mkdir -p $out
for v in stamps gsig stamps gsig; do
  echo "== $v" | tee -a $out/gelu_sigmoid.txt
  DUODIFF_LIB=$R/duodiff_amd/libduodiff_$v.so timeout -k 10 200 python3 tools/power_probe.py --iters 2000 --tiles 256 128 2>> $out/err.txt | grep "random" | head -2 | tee -a $out/gelu_sigmoid.txt
done
echo "== accuracy (tools/mlp_unit.py: max / rms error of the fused MLP against a float64 reference with the exact-erf GELU)" | tee -a $out/gelu_sigmoid.txt
for v in stamps gsig; do
  echo "-- $v" | tee -a $out/gelu_sigmoid.txt
  DUODIFF_LIB=$R/duodiff_amd/libduodiff_$v.so timeout -k 10 200 python3 tools/mlp_unit.py --M 4096 --D 512 --proj 2>> $out/err.txt | tee -a $out/gelu_sigmoid.txt
  DUODIFF_LIB=$R/duodiff_amd/libduodiff_$v.so timeout -k 10 200 python3 tools/mlp_unit.py --M 1024 --D 256 --ln 2>> $out/err.txt | tee -a $out/gelu_sigmoid.txt
done
DUODIFF_LIB=$R/duodiff_amd/libduodiff_gsig.so timeout -k 10 300 python3 -m pytest tests/test_mlp_fused.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out/gelu_sigmoid.txt
