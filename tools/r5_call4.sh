#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5d; rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 tools/poison_probe.py > $out/poison_probe.txt 2> $out/poison_probe.err || { echo "poison probe failed"; tail -5 $out/poison_probe.err; }
cat $out/poison_probe.txt
for v in stamps abl_nogelu abl_noread abl_nodma abl_nomfma; do
  [ -f duodiff_amd/libduodiff_$v.so ] || continue
  echo "== $v" | tee -a $out/ablations.txt
  DUODIFF_LIB=$PWD/duodiff_amd/libduodiff_$v.so timeout -k 10 200 python3 tools/power_probe.py --iters 2000 --tiles 256 128 2>> $out/ablations.err | grep -v "^tiles= *... zero" | tee -a $out/ablations.txt
done
