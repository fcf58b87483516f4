#!/bin/bash
# Host-side sanitizer pass (GPU sanitizers are not available on this pool): builds the C oracle and the host build of the
# device headers (tests/hostsim) with sanitizers and runs the CPU parity tests against those builds.
#   tools/sanitize_host.sh            UBSan only: recoverable, -O1, no -g (~5 min of build here); fails on any "runtime error:" line
#   tools/sanitize_host.sh full       AddressSanitizer + UBSan (-O1 -g, the hostsim build takes ~15 min here)
# The instrumented libraries are built under /tmp/bls_san and handed to the tests through BLS381_ORACLE_SO / BLS381_HOSTSIM_SO
# (tests/blsutil.py, oracle/coracle.py); the in-tree libraries are not touched.  pytest runs with -s: a passing test would
# otherwise swallow the sanitizer's report.  Log: /tmp/bls_san/<mode>.log (copy to profiles/rNN_misc/ to commit it).
set -e -o pipefail
cd "$(dirname "$0")/.."
MODE=${1:-ubsan}
INC=plonky2-bls12-381-pairing_amd/csrc
OUT=/tmp/bls_san
mkdir -p $OUT
if [ "$MODE" = full ]; then
  SAN="-O1 -g -fsanitize=address,undefined"
  PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
else
  SAN="-O1 -fsanitize=undefined"
  PRE="$(gcc -print-file-name=libubsan.so)"
fi
LOG=$OUT/$MODE.log
{
  echo "== sanitize_host.sh $MODE: $(git rev-parse --short HEAD 2>/dev/null) $(date -u +%FT%TZ)"
  echo "== gcc $SAN (oracle/bls381_oracle.c)"
  gcc $SAN -fPIC -shared oracle/bls381_oracle.c -o $OUT/libbls381_oracle_$MODE.so -lpthread
  echo "== g++ -std=c++17 $SAN (tests/hostsim/hostsim.cpp)"
  time g++ -std=c++17 $SAN -fPIC -shared -I $INC tests/hostsim/hostsim.cpp -o $OUT/libhostsim_$MODE.so
} 2>&1 | tee $LOG
set +e
BLS381_ORACLE_SO=$OUT/libbls381_oracle_$MODE.so BLS381_HOSTSIM_SO=$OUT/libhostsim_$MODE.so \
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=0 \
  python -m pytest tests/test_oracle_c.py tests/test_hostsim.py tests/test_witness_hostsim.py -q -s 2>&1 | tee -a $LOG
RC=${PIPESTATUS[0]}
set -e
ERRS=$(grep -c "runtime error:\|ERROR: AddressSanitizer" $LOG || true)
echo "== pytest exit $RC, sanitizer reports: $ERRS" | tee -a $LOG
[ "$RC" = 0 ] && [ "$ERRS" = 0 ]
