// ptnn_device.hpp -- gfx950 (CDNA4, wave64) device code of the parallel-tempering FNN sampler.
//
// One work-group owns one replica (one temperature of the ladder).  Wave 0 of the group runs the
// inherently sequential parts (the row-by-row SGD sweep of Network.langevin_gradient, REG:99-118);
// all NW waves share the row-parallel parts (evaluate_proposal over train+test rows, REG:120-134,
// the Philox noise pass and the trace-row store).  Weights, the data set and every per-step
// vector live in LDS; HBM sees only the trace rows the result-file layout requires.
//
// Layout: this file holds the parameter blocks (SegParams, SegDyn, the trace-row layout) and includes the parts in order
// (ptnn_dev_*.hpp, textually, inside namespace ptnn): math and tape, the SGD epoch and the forward pass, then the interval BODIES of
// the five schedules (segment_body, segment_spec_body, segment_pack_body, segment_wide_body, segment_tree_body: all MH steps of one
// swap interval; the tree and the packed multi-CU body also run the swap rounds between their intervals), swap_block (one
// work-group's share of a swap round), and at the end persistent_loop + the __global__ kernels: body, grid barrier, swap_block,
// next interval -- one launch per run.  ptnn_diag.hpp is included by diagnostic builds only.
//
// Written for gfx950 only: wave size 64, DPP row operations, v_permlane{16,32}_swap.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

// No implicit fused multiply-adds anywhere in the device code: whether `a * b + c` becomes one v_fma or a multiply and an add
// would otherwise be decided per inlined copy by what surrounds it, and the same function could round differently in two kernels
// (found by the stress tool: the bias update `b1 -= g * ldh` of the SGD epoch differed by one ulp between the packed and the
// multi-CU speculative kernel on classification nets).  Every fused operation in this file is written as fmaf / v_pk_fma.
#pragma clang fp contract(off)

namespace ptnn {

constexpr int TASK_REG = 0;
constexpr int TASK_CLS = 1;
constexpr int WAVE = 64;
// Diagnostic hooks.  The product build contains none of the diagnostic code: STAMP / FW_DBG / PTNN_DIAG expand to nothing.  A
// diagnostic build (-DPTNN_STAMPS, profiles/tools/build_stamps.sh; never the product) includes ptnn_diag.hpp, which holds the
// bodies: in-kernel cycle stamps per phase of a round, summed into SegParams::stamps and read back by ptnn_debug_stamps.
#ifdef PTNN_STAMPS
#include "ptnn_diag.hpp"
#else
#define STAMP(slot) do { } while (0)
#define FW_DBG(q_) do { } while (0)
#define PTNN_DIAG(name)
#endif

constexpr int MAX_WAVES = 8;            // waves per work-group: 2 per SIMD, 256 VGPRs each
constexpr int MAX_THREADS = MAX_WAVES * WAVE;

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float LOG_2PI = 1.8378770664093453f;

// Philox4x32-10 streams (specification shared with oracle/ptnn_oracle.py)
constexpr uint32_t STREAM_STEP = 0, STREAM_WNOISE = 1, STREAM_SWAP = 2, STREAM_INIT = 3;

// per-replica float state (st_f) and int state (st_i) slots
enum { SF_LIK = 0, SF_PRIOR, SF_TAU_LAST, SF_REC_RMSE_TR, SF_REC_RMSE_TE, SF_REC_ACC_TR, SF_REC_ACC_TE, SF_COUNT = 8 };
enum { SI_NACC = 0, SI_REC_ROW, SI_LG_COUNT, SI_LG_ACC, SI_COUNT = 4 };   // accepted steps, trace row holding the last recorded pos_w (compact traces), Langevin steps proposed / accepted
// scalar trace row of MH step i (row i + 1): what the per-chain result files need beside pos_w (REG:454-481) -- likeh_list
// column 0 (REG:391 / CLS:404), rmse_train, rmse_test, acc_train, acc_test, accept_list (the count BEFORE the step, REG:380, as
// int bits) -- plus the step's log alpha as the kernel computed it (diagnostic: the parity tests measure its fp32 error with it)
// TR_ACC_TR of a REGRESSION (whose acc_train is identically 0, REG:403): eta of the recorded state (finish_eval<TASK, true>)
// TR_SRC (int bits): compact traces only -- the trace row that holds this step's pos_w values (the step's own row when it was
// accepted, else the row of the last accepted step: pos_w[i+1] = pos_w[i] on a reject, REG:417); 0 in the full layout
enum { TR_LIKEH = 0, TR_RMSE_TR, TR_RMSE_TE, TR_ACC_TR, TR_ACC_TE, TR_ACCEPT, TR_LOGALPHA, TR_SRC, TR_COUNT = 8 };

// one lane writes the whole row: two 16-byte stores into one 32-byte sector
__device__ __forceinline__ void store_trace_row(float* __restrict__ row, float likeh, float rm_tr, float rm_te, float ac_tr, float ac_te,
                                                int accept_before, float logalpha, int src_row = 0) {
    float4* q = reinterpret_cast<float4*>(row);
    q[0] = make_float4(likeh, rm_tr, rm_te, ac_tr);
    q[1] = make_float4(ac_te, __int_as_float(accept_before), logalpha, __int_as_float(src_row));
}

struct SegParams {
    int H, P, PS;            // hidden units, parameters, state row length (P + 1 rounded up to 4)
    int Ntr, Nte, IPY, FWS;  // rows, data row stride (floats), packed forward row stride (floats)
    int pk_nred;             // packed schedule: log2 of the lane-group width (3: n_hidden <= 8, 4: n_hidden <= 16)
    int noise_shared;        // Q14: every replica reads the step tape of replica 0
    int fw_mfma;             // cooperative schedule: forward pass on the matrix cores (host decides: 24 <= H <= 64, I >= 6): 1 exact fp32
                             // instruction, 2 split bf16 operands (SplitK; cooperative kernel only)
    int xy_global;           // split forward pass without room for the row-major data image in LDS: rows are read from p.data
    int S, switch_step, use_lg;
    int trace_cap;           // rows per replica in the trace rings
    int first_global;
    float l_prob, lr, step_w, step_eta;
    float inv_2sig2, prior_c, nu1, nu2;
    uint32_t seed_lo, seed_hi;
    const float* data;       // [Ntr+Nte][IPY]: x_0..x_{I-1}, y, pad
    float* w_state;          // current (w, eta) rows [Rl][PS]
    float* rec_w;            // last recorded pos_w row [Rl][PS]
    float* gd_w;             // cached langevin_gradient(w) [Rl][PS]
    float* st_f;             // [Rl][SF_COUNT]
    int* st_i;               // [Rl][SI_COUNT]
    int* gd_valid;           // [Rl] 1 when gd_w holds langevin_gradient of the current w
    const float* temps;      // [Rl]
    float* L_handoff;        // [Rglobal] posted scalar at a hand-off (REG:430 / CLS:439)
    float* L_final;          // [Rglobal] end-of-chain scalar (REG:442 / CLS:451)
    float* L_raw;            // [Rglobal] untempered log-likelihood of the current state (swap_rule 1 only)
    float* prior_post;       // [Rglobal] prior of the current state (swap_rule 1 only)
    float* tr_pos_w;         // [Rl][S][PW]: pos_w rows padded to whole 64-byte sectors (pad written as zeros: no partial-sector writes)
    int PW;                  // P rounded up to 16 floats
    float* tr_scal;          // [Rl][S][TR_COUNT]: the scalars of a step in ONE 32-byte row (one sector instead of seven)
    // speculative schedule across G work-groups (CUs) per replica
    int G;                   // work-groups per replica (1 = no cross-CU exchange)
    unsigned epoch_base;     // granule tags of this launch are epoch_base + round
    int tree_ahead;          // tree schedule: LDS holds two sets of tapes, the next round's are drawn while the records travel
    int wide_window;         // wide nets, several work-groups: steps per round (<= WIDE_WINDOW)
    unsigned long long* xverdict; // [Rl][2][MAX_SLOTS] one {tag, accepted?} granule per slot and round: all a foreign group polls
    unsigned long long* xslots;   // [Rl][2][MAX_SLOTS][16] result granules of an ACCEPTED slot (read by the other groups at commit)
    unsigned long long* xw;       // [Rl][2][MAX_SLOTS][2 PS] accepted-proposal granules
    int* error_flag;         // != 0 after a launch: a bounded spin expired
    unsigned long long* stamps;   // diagnostic build only (PTNN_STAMPS): cycle sums per phase, else unused
    float* wide_scratch;     // [Rl][3][PS] proposal, its SGD epoch, noise (wide nets only: these do not fit in LDS)
    const float* xt;         // wide nets: transposed data image Xt[k][Npad] (B operand of the MFMA forward pass), or null
    const uint4* xs;         // wide nets, split-operand forward pass (fw_mfma == 2): [3 levels][Npad rows][SplitK::CH chunks of 8 bf16], or null
    int Npad;                // rows of Xt, Nall rounded up to 32
    int forward_bf16;        // 1: forward GEMM operands rounded to bf16 (fp32 accumulate); 0: exact fp32 MFMA
    int* seg_progress;       // pinned host word or null (RCCL communicator attached): block 0 stores seg_ordinal when this launch ends
    int seg_ordinal;         // with a swap round due -- "the collective of round seg_ordinal - 1 is next on the stream" (ptnn.hip: wait_stream)
    int xcd_granules;        // 1: work-groups of a replica that find themselves on one XCD exchange through its L2 (granule_*_xcd); 0: always agent scope
    unsigned long long* xswap;    // granules of the swap rounds a multi-group launch runs by itself (tree, packed multi-CU): 2 x swap_xchg_granules, or null
    int compact;             // wide nets, all rows resident (trace_cap == S): a REJECTED step writes no pos_w row, only the index of the
                             // row it repeats (TR_SRC); ptnn_get_traces fills the rows in.  A 70 KB copy per rejected step otherwise.
};

// What changes from one swap interval to the next inside one launch (persistent_loop): which of the two state buffers is
// current, and where the granule tags continue.  Kept apart from SegParams so that the kernel argument itself stays constant (a
// modified copy of it, live across the whole interval, cost ~60 scalar registers and pushed two kernels into scratch).
struct PersistParams;
typedef __attribute__((address_space(4))) const PersistParams* persist_cptr;
struct SegDyn {
    persist_cptr pp;         // the launch's PersistParams (kernel-argument segment), found by the KERNEL and handed down: a body that the
                             // compiler does not inline (many-class heads) must not look for the kernel arguments itself
    float* w_state;          // current (w, eta) rows [Rl][PS]
    float* gd_w;             // cached langevin_gradient(w) [Rl][PS]
    int* gd_valid;           // [Rl]
    unsigned epoch_base;     // granule tags of this interval are epoch_base + round
};

#include "ptnn_dev_math.hpp"                 // scalar math, the Philox tape, wave reductions, LDS helpers, the hand-scheduled SGD rows of the 4-H-1 nets
#include "ptnn_dev_sweep_forward.hpp"        // R4/R5 the SGD epoch (sgd_sweep), R1-R3/R6/R7 forward image, eval_rows, likelihood / prior / proposal ratio, chain start-up
#include "ptnn_dev_coop.hpp"                 // cooperative schedule: matrix-core forward passes (exact fp32 / split bf16 operands) and segment_body
#include "ptnn_dev_spec.hpp"                 // speculative schedules: slots, {tag, value} granules (agent scope / through an XCD's L2), the swap cascade, PersistParams, segment_spec_body
#include "ptnn_dev_pack.hpp"                 // packed speculative schedule on one CU and over several (segment_pack_body), with its own swap rounds inside a launch
#include "ptnn_dev_wide.hpp"                 // wide nets (64 < H <= 512): sgd_sweep_wide, matrix-core forward, segment_wide_body, model_wide_kernel; then swap_block and the non-template kernels
#include "ptnn_dev_tree.hpp"                 // prefetching tree schedule (segment_tree_body), with its own swap rounds inside a launch
#include "ptnn_dev_kernels.hpp"              // model_kernel, persistent_loop, the __global__ segment kernels, the per-shape table

}  // namespace ptnn
