#!/bin/bash
# The ablation / stamp builds of tools/lab are measurement artefacts of rounds 1-2: instrumented variants (-DLAB_*) of the
# kernels AS THEY WERE at the end of round 2.  Since round 3 the product sources carry no lab code at all; this script puts
# the round-2 tree (commit 2491dc7: csrc + include) under tools/lab/src, and tools/lab/Makefile builds the variants from
# there -- self-contained libraries that never mix with the product build.  Run in the build container (needs .git).
set -e
cd "$(dirname "$0")"
rm -rf src && mkdir -p src
git -C ../.. archive ${LAB_COMMIT:-2491dc7} bayesian-neural-nets_amd/csrc include | tar -x -C src
echo "lab sources of ${LAB_COMMIT:-2491dc7} in tools/lab/src"
