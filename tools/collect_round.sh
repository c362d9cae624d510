#!/bin/bash
# final collection of the round: the whole GPU suite (its -s log = profiles/rNN/parity_numbers.txt), then -- only if it is green -- the widened-loop bench and
# the profiles of all three workloads (tools/collect_profiles.sh; or of the workloads named as arguments: gpurun calls are capped at 20 minutes), on one box and one build
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=gpurun_out/final_${ROUND:-r05}; rm -rf $out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/parity_numbers.txt 2>&1; rc=$?; tail -3 $out/parity_numbers.txt
[ $rc -eq 0 ] || { echo "GPU suite failed: nothing collected"; grep "^FAILED" $out/parity_numbers.txt; exit 1; }
timeout -k 10 300 python3 tools/widened_bench.py --out $out/widened_bench.json > $out/widened.log 2>&1 && tail -1 $out/widened.log | cut -c1-700 &&
ROUND=${ROUND:-r05} bash tools/collect_profiles.sh "$@" > $out/collect.log 2>&1; tail -4 $out/collect.log | cut -c1-300
