#!/bin/bash
# Diagnostic build of libptnn with in-kernel cycle stamps (-DPTNN_STAMPS; NOSTAMPS=1: without them, e.g. for -DPTNN_ABLATE=k in $EXTRA), one shape only (default REG 4 -> 1; pass
# "task I O" for another, e.g. `build_stamps.sh 1 34 2`).  Never the product.
set -e
cd "$(dirname "$0")/../.."
T=${1:-0}; I=${2:-4}; O=${3:-1}
C=parallel-tempering-neural-net_amd/csrc
DEF=${NOSTAMPS:+-DPTNN_NO_STAMPS_BUILD}; [ -z "$NOSTAMPS" ] && DEF=-DPTNN_STAMPS
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $DEF $EXTRA -DPTNN_SHAPES(X)=X($T,$I,$O)"
/opt/rocm/bin/hipcc $F -c -o /tmp/ptnn_stamps_main.o $C/ptnn.hip
/opt/rocm/bin/hipcc $F -DPTNN_T=$T -DPTNN_I=$I -DPTNN_O=$O -DPTNN_SHAPE_SYMBOL=ptnn_shape_${T}_${I}_${O} -c -o /tmp/ptnn_stamps_shape.o $C/ptnn_shape.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/tools/libptnn_stamps.so /tmp/ptnn_stamps_main.o /tmp/ptnn_stamps_shape.o
echo built profiles/tools/libptnn_stamps.so
