#!/bin/bash
# Diagnostic build of libptnn with in-kernel cycle stamps (-DPTNN_STAMPS), one shape only: REG 4 -> 1.  Never the product.
set -e
cd "$(dirname "$0")/../.."
C=parallel-tempering-neural-net_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPTNN_STAMPS -DPTNN_SHAPES(X)=X(0,4,1)"
/opt/rocm/bin/hipcc $F -c -o /tmp/ptnn_stamps_main.o $C/ptnn.hip
/opt/rocm/bin/hipcc $F -DPTNN_T=0 -DPTNN_I=4 -DPTNN_O=1 -DPTNN_SHAPE_SYMBOL=ptnn_shape_0_4_1 -c -o /tmp/ptnn_stamps_shape.o $C/ptnn_shape.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/tools/libptnn_stamps.so /tmp/ptnn_stamps_main.o /tmp/ptnn_stamps_shape.o
echo built profiles/tools/libptnn_stamps.so
