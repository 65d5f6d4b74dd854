#!/bin/bash
# SQ wave-cycle counters on the bs=1 dispatches of the 16x16x4 conv kernels (dev tool; run through gpurun from the repo root:
# `bash tools/sq_counters.sh`): where a latency-bound layer's wave cycles go (active / issue stalls / LDS / parked at waitcnt or
# barrier) and its LDS bank conflicts.  Two separate --pmc passes (counter groups), CSVs under gpurun_out/sq1 and sq2.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/sq1 -- python $R/tools/conv_report.py 1 4 yolact > $R/gpurun_out/sq1.log 2>&1 &&
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq2 -- python $R/tools/conv_report.py 1 4 yolact > $R/gpurun_out/sq2.log 2>&1
