// ptnn_shape.hip -- one (task, n_in, n_out) instantiation of every templated kernel: compiled once per shape of
// ptnn_shapes.hpp with -DPTNN_T=<task> -DPTNN_I=<n_in> -DPTNN_O=<n_out> -DPTNN_SHAPE_SYMBOL=ptnn_shape_<T>_<I>_<O>.
#define PTNN_SHAPE_TU 1
#include "ptnn_shapes.hpp"

#if !defined(PTNN_T) || !defined(PTNN_I) || !defined(PTNN_O) || !defined(PTNN_SHAPE_SYMBOL)
#error "compile with -DPTNN_T -DPTNN_I -DPTNN_O -DPTNN_SHAPE_SYMBOL"
#endif

using namespace ptnn;

extern "C" const Shape PTNN_SHAPE_SYMBOL = {PTNN_T, PTNN_I, PTNN_O,
                                            &segment_kernel<PTNN_T, PTNN_I, PTNN_O>, &segment_spec_kernel<PTNN_T, PTNN_I, PTNN_O>,
                                            &model_kernel<PTNN_T, PTNN_I, PTNN_O>, &segment_wide_kernel<PTNN_T, PTNN_I, PTNN_O>,
                                            &model_wide_kernel<PTNN_T, PTNN_I, PTNN_O>, &segment_pack_kernel<PTNN_T, PTNN_I, PTNN_O>,
                                            &segment_tree_kernel<PTNN_T, PTNN_I, PTNN_O>, &segment_wide_res_kernel<PTNN_T, PTNN_I, PTNN_O>,
                                            (coop_has_loop<PTNN_T, PTNN_I, PTNN_O>() ? 1 : 0) | (pack_has_loop<PTNN_T, PTNN_I, PTNN_O>() ? 2 : 0),
                                            SplitK<PTNN_I>::OK ? SplitK<PTNN_I>::CH : 0, SplitK<PTNN_I>::KR,
                                            &segment_packm_kernel<PTNN_T, PTNN_I, PTNN_O>};
