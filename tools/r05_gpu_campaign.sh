#!/bin/bash
# closing campaign of round 5's final library (the links' contact response on by default): parity fuzz in its four modes, the long soak
OUT=gpurun_out/${1:-r05q}; mkdir -p $OUT
{ echo '```'
  for args in "700 81" "500 82 fallen" "300 84 lookahead"; do printf "tools/fuzz_parity.py %-28s" "$args:"; timeout 1500 python tools/fuzz_parity.py $args 2>&1 | grep -v amdgpu.ids | tail -4 | tr '\n' ' '; echo; done
  printf "QS_STEP_VARIANT=2 tools/fuzz_parity.py 300 83 fallen:  "; QS_STEP_VARIANT=2 timeout 1500 python tools/fuzz_parity.py 300 83 fallen 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' '; echo
  echo; echo "tools/long_soak.py 300000 150000   (the benchmark's workload with the default handle: body_contacts=True)"
  timeout 1500 python tools/long_soak.py 300000 150000 2>&1 | grep -v amdgpu.ids | tail -4
  echo '```'; } > $OUT/fuzz_soak.txt 2>&1
cat $OUT/fuzz_soak.txt
