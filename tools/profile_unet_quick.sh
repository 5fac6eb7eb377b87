#!/bin/bash
# rocprofv3 kernel stats of tools/unet_time.py (UNet backbone, 8 patches of 256x256: inference and training forward +
# backward); run on the GPU box from the repo root.  Summary: gpurun_out/prof_unet_quick/t_kernel_stats.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_unet_quick
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/tools/unet_time.py 8 > $OUT/run.log 2>&1 || exit 1
echo "trace done"
