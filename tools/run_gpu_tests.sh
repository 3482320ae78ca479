#!/bin/bash
# GPU test wrapper: stdout and stderr are kept SEPARATELY under gpurun_out/<tag>/ and pytest captures at the sys level
# only, so what the HSA runtime writes to fd 2 when a queue dies (the fault address / "Memory access fault by GPU node"
# line) lands in stderr.log even when the process aborts (round 5's test2.log held stdout + faulthandler only).
#   tools/run_gpu_tests.sh <tag> [pytest args ...]      default args: tests -m gpu -x -q
tag=${1:-gputests}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
args=("$@"); [ ${#args[@]} -eq 0 ] && args=(tests -m gpu -x -q)
export AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-0} HSA_ENABLE_DEBUG=${HSA_ENABLE_DEBUG:-0}
timeout -k 10 ${VT_TEST_TIMEOUT:-1000} python -X faulthandler -m pytest --capture=sys "${args[@]}" > "$out/stdout.log" 2> "$out/stderr.log"
rc=$?
tail -5 "$out/stdout.log"
[ -s "$out/stderr.log" ] && { echo "---- stderr (tail) ----"; tail -20 "$out/stderr.log"; }
exit $rc
