#!/bin/bash
# The CPU oracle (test infrastructure) under AddressSanitizer + UBSan: rebuilds oracle/libpedn_oracle.so with the sanitizers,
# runs the oracle-only test files against it, restores the normal build.  CPU only (GPU ASAN is not available on this pool).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/oracle
cp libpedn_oracle.so /tmp/libpedn_oracle_keep.so 2>/dev/null || true
trap 'cd $R/oracle && rm -f libpedn_oracle.so && make -s libpedn_oracle.so' EXIT
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -shared -o libpedn_oracle.so pedn_oracle.c -lm
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 -m pytest tests/test_oracle_golden.py tests/test_node_lp.py \
    tests/test_rng_contract.py -x -q -m "not gpu"
