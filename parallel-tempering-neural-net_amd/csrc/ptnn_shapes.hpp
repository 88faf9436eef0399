// ptnn_shapes.hpp -- the compiled (task, n_in, n_out) shapes and the per-shape kernel table.
// Every shape is its own translation unit (ptnn_shape.hip with -DPTNN_T/-DPTNN_I/-DPTNN_O), so that the build can compile them
// in parallel; ptnn.hip only references the tables.
//
// REG: the shipped time series have 4 lag inputs (REG:916); 5 and 32 cover BASELINE.json's literal [5,H,1] and the synthetic
// [32,H,1].  CLS: the reference's problem table (CLS:909-995): iris 4/3, ionosphere 34/2, cancer 9/2, wine 11/10, bank 20/2,
// pendigit 16/10, chess 6/18.  n_hidden is a run-time value.
#pragma once
#include "ptnn_device.hpp"

#ifndef PTNN_SHAPES
#define PTNN_SHAPES(X) X(0, 4, 1) X(0, 5, 1) X(0, 32, 1) X(1, 4, 3) X(1, 34, 2) X(1, 9, 2) X(1, 11, 10) X(1, 20, 2) X(1, 16, 10) X(1, 6, 18)
#endif

namespace ptnn {
typedef void (*seg_fn)(const SegParams, const PersistParams, int);
typedef void (*model_fn)(const SegParams, int, const float*, const float*, float*, int, int);

struct Shape {
    int task, I, O;
    seg_fn seg;             // cooperative schedule, H <= 64
    seg_fn spec;            // speculative schedule over work-groups, H <= 64
    model_fn model;
    seg_fn seg_wide;        // 64 < H <= 512
    model_fn model_wide;
    seg_fn pack;            // H <= 8: packed speculative schedule
    seg_fn tree;            // prefetching tree schedule (random-walk classification), H <= 64
    seg_fn seg_wide_res;    // 64 < H <= 512, H % 32 == 0, state + proposal resident in LDS
    int loops;              // which kernels carry the interval loop of a persistent launch: bit 0 seg, bit 1 pack (wide: always; spec, tree: never)
    int split_ch, split_kr; // split-operand forward pass (SplitK<I>): 16-byte chunks per image row (0: not available for this n_in), fp32 k-steps
    seg_fn packm;           // 9 <= H <= 16: packed schedule over several CUs per replica
};
}  // namespace ptnn
