#!/bin/bash
# launch-order view of one B = 36 step (and B = 256 with "256"): gpurun_out/seq/step_b<B>.txt
B=${1:-36}
O=gpurun_out/seq; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/trace_b$B
rocprofv3 --kernel-trace --output-format csv -d $O/trace_b$B -- python3 bench.py --batch $B --steps 4 --warmup 3 --no-cpu-baseline --no-fwd-rate --no-kernel-timing > $O/b$B.out 2> $O/b$B.err
T=$(find $O/trace_b$B -name "*kernel_trace.csv" | head -1)
python3 tools/step_sequence.py "$T" $O/step_b$B.txt
rm -rf $O/trace_b$B
tail -45 $O/step_b$B.txt
