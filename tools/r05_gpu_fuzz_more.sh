#!/bin/bash
# a second fuzz pass over the final library with seeds the round's other passes did not use (the library before it was caught by these gates: DESIGN 10)
OUT=gpurun_out/${1:-r05i}; mkdir -p $OUT
{ echo '```'
  for args in "800 91" "700 92 fallen" "400 94 lookahead"; do printf "tools/fuzz_parity.py %-28s" "$args:"; timeout 1500 python tools/fuzz_parity.py $args 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400; done
  printf "QS_STEP_VARIANT=2 tools/fuzz_parity.py 500 93 fallen:  "; QS_STEP_VARIANT=2 timeout 1500 python tools/fuzz_parity.py 500 93 fallen 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
  printf "QS_STEP_VARIANT=2 tools/fuzz_parity.py 300 95 lookahead:  "; QS_STEP_VARIANT=2 timeout 1500 python tools/fuzz_parity.py 300 95 lookahead 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
  echo '```'; } > $OUT/fuzz_more.txt 2>&1
cat $OUT/fuzz_more.txt
