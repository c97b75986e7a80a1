#!/bin/bash
# round 5: rocprofv3 kernel statistics + the PMC passes of the final tree (C3, C2, C5) and the grid mode's kernel statistics
export TMPDIR=/tmp
bash tools/profile.sh r5 2>&1 | tail -25
bash tools/grid_profile.sh r5grid 2>&1 | tail -8
