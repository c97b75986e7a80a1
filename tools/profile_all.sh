#!/bin/bash
# rocprofv3 kernel statistics + the PMC passes of the tree (C3, C2, C5: tools/profile.sh) and the grid mode's kernel statistics (tools/grid_profile.sh);
# afterwards: python tools/summarize_profile.py gpurun_out/prof_TAG TAG   (TAG = first argument, default r5)
export TMPDIR=/tmp
bash tools/profile.sh ${1:-r5} 2>&1 | tail -25
bash tools/grid_profile.sh ${1:-r5}grid 2>&1 | tail -8
