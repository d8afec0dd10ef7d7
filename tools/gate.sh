#!/bin/bash
# The binary gate: ONE command, ONE verdict for a built libqs_hip.so.  The step kernels sit at the
# edge of what ROCm 7.2's register allocator handles -- round 5 shipped a source that was right and a binary that was not -- so a library is
# accepted as a binary: the GPU suite with either step kernel, the parity fuzz in its four modes, the soak.  The verdict is one JSON line
# (library_sha256, source_sha256 as the library itself reports it, counts, deviated = 0) appended to gpurun_out/validated_libraries.jsonl;
# commit it as a line of profiles/validated_libraries.jsonl.  Exit code 0 = accepted.
#     bash tools/gate.sh <dir> [quick]        (on the GPU box: gpurun -- 'bash tools/gate.sh gate_r06x'; about 15 minutes, "quick" 8)
OUT=gpurun_out/${1:-gate}; mkdir -p $OUT
Q=${2:-full}
if [ "$Q" = quick ]; then F1=150; F2=150; F3=100; F4=100; SOAK="60000 30000"; else F1=500; F2=400; F3=250; F4=250; SOAK="300000 150000"; fi
rm -f gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_default.log 2>&1; tail -2 $OUT/pytest_default.log
# (the parity records of this run, to be committed as profiles/rNN_impact_parity.jsonl, ..._full_size_oracle_sampled.jsonl, ..._terminal_observation_parity.json)
cp gpurun_out/impact_parity.jsonl gpurun_out/full_size_oracle_sampled.jsonl gpurun_out/terminal_observation_parity.json $OUT/ 2>/dev/null
QS_STEP_VARIANT=2 timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q > $OUT/pytest_dense.log 2>&1; tail -2 $OUT/pytest_dense.log
timeout 1500 python tools/fuzz_parity.py $F1 91 > $OUT/fuzz_plain.log 2>&1; tail -1 $OUT/fuzz_plain.log
timeout 1500 python tools/fuzz_parity.py $F2 92 fallen > $OUT/fuzz_fallen.log 2>&1; tail -1 $OUT/fuzz_fallen.log
timeout 1500 python tools/fuzz_parity.py $F3 94 lookahead > $OUT/fuzz_lookahead.log 2>&1; tail -1 $OUT/fuzz_lookahead.log
QS_STEP_VARIANT=2 timeout 1500 python tools/fuzz_parity.py $F4 93 fallen > $OUT/fuzz_fallen_dense.log 2>&1; tail -1 $OUT/fuzz_fallen_dense.log
timeout 1500 python tools/long_soak.py $SOAK > $OUT/soak.log 2>&1; tail -2 $OUT/soak.log
python tools/gate_record.py $OUT $Q
