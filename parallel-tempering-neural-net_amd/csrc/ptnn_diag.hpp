// ptnn_diag.hpp -- DIAGNOSTIC BUILD ONLY (-DPTNN_STAMPS; profiles/tools/build_stamps.sh).  Never part of the product: ptnn_device.hpp
// includes this file only when PTNN_STAMPS is defined, otherwise STAMP / FW_DBG / PTNN_DIAG expand to nothing.
//
// In-kernel cycle stamps: wave 0 of the first work-group of replica 0 adds up shader-clock cycles per phase of a round and writes
// the sums to SegParams::stamps at the end of the launch (ptnn_debug_stamps reads and resets them; profiles/tools/stamps*.py print
// them).  The hook bodies below are pasted into the kernels by name (PTNN_DIAG(name)) and use the local names of the place they
// are pasted into (tid, lane, wave, p, r, grp, ...).
#pragma once

#define PTNN_DIAG(name) PTNN_DIAG_##name

#define STAMP(slot)                                                                          \
    do {                                                                                     \
        if (stamp_on) {                                                                      \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
            __builtin_amdgcn_s_waitcnt(0xC07F);                                              \
            stamp_acc[slot] += t_ - stamp_last; stamp_last = t_;                             \
        }                                                                                    \
    } while (0)

// cycles of the phases of eval_rows_mfma_coop / eval_rows_mfma_split (block 0, wave 0)
static __device__ unsigned long long fw_dbg[8];
#define FW_DBG(q_) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                        fw_dbg[q_] += t_ - fw_t; fw_t = t_; } } while (0)

#define PTNN_DIAG_fw_begin \
    unsigned long long fw_t = __builtin_amdgcn_s_memtime();

#define PTNN_DIAG_coop_begin \
    const bool stamp_on = (blockIdx.x == 0 && tid < WAVE); \
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F); \
    const unsigned long long stamp_t0 = stamp_last;

#define PTNN_DIAG_coop_flush \
    if (stamp_on && (tid & 63) == 0 && p.stamps) { \
        for (int q_ = 0; q_ < 9; ++q_) atomicAdd(p.stamps + q_, stamp_acc[q_]); \
        atomicAdd(p.stamps + 9, (unsigned long long)n_steps); \
        atomicAdd(p.stamps + 10, __builtin_amdgcn_s_memtime() - stamp_t0); \
        if (tid == 0) for (int q_ = 0; q_ < 8; ++q_) { atomicAdd(p.stamps + 150 + q_, fw_dbg[q_]); fw_dbg[q_] = 0; } \
    }

#define PTNN_DIAG_spec_entry \
    const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);

#define PTNN_DIAG_spec_begin \
    const bool stamp_on = (blockIdx.x == 0 && wave == 0); \
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F); \
    unsigned long long stamp_rounds = 0; \
    const unsigned long long stamp_t0 = stamp_last; \
    const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);

#define PTNN_DIAG_count_round \
    stamp_rounds += 1;

#define PTNN_DIAG_spec_flush \
    if (stamp_on && lane == 0 && p.stamps) { \
        for (int q_ = 0; q_ < 9; ++q_) atomicAdd(p.stamps + q_, stamp_acc[q_]); \
        atomicAdd(p.stamps + 9, stamp_rounds); \
        atomicAdd(p.stamps + 10, __builtin_amdgcn_s_memtime() - stamp_t0); \
        atomicAdd(p.stamps + 11, __builtin_amdgcn_s_memrealtime() - stamp_rt0); \
    } \
    if (tid == 0 && p.stamps) { \
        const unsigned long long now = __builtin_amdgcn_s_memrealtime(); \
        atomicMin(p.stamps + 12, stamp_entry); \
        atomicMax(p.stamps + 13, stamp_rt0 - stamp_entry); \
        atomicMax(p.stamps + 14, now - stamp_rt0); \
        atomicMax(p.stamps + 15, now); \
    } \
    if (grp == 0 && tid == 0 && p.stamps && r < 64) { \
        atomicAdd(p.stamps + 16 + 2 * r, __builtin_amdgcn_s_memtime() - stamp_t0); \
        atomicAdd(p.stamps + 17 + 2 * r, stamp_rounds); \
    }

#define PTNN_DIAG_pack_begin \
    const bool stamp_on = (blockIdx.x == 0 && wave == 0); \
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F); \
    unsigned long long stamp_rounds = 0, stamp_eval = 0; \
    const unsigned long long stamp_t0 = stamp_last;

#define PTNN_DIAG_pack_eval_begin \
    const unsigned long long ev_t0 = __builtin_amdgcn_s_memtime();

#define PTNN_DIAG_pack_eval_end \
    if (ev_i == 0) stamp_eval += __builtin_amdgcn_s_memtime() - ev_t0;

#define PTNN_DIAG_pack_flush \
    if (stamp_on && lane == 0 && p.stamps) { \
        for (int q_ = 0; q_ < 9; ++q_) atomicAdd(p.stamps + q_, stamp_acc[q_]); \
        atomicAdd(p.stamps + 9, stamp_rounds); \
        atomicAdd(p.stamps + 10, __builtin_amdgcn_s_memtime() - stamp_t0); \
    } \
    if (blockIdx.x == 0 && ev_i == 0 && lane == 0 && p.stamps) atomicAdd(p.stamps + 11, stamp_eval); \
    if (tid == 0 && p.stamps && r < 64) { \
        atomicAdd(p.stamps + 16 + 2 * r, __builtin_amdgcn_s_memtime() - stamp_t0); \
        atomicAdd(p.stamps + 17 + 2 * r, stamp_rounds); \
    }

#define PTNN_DIAG_tree_begin \
    const bool stamp_on = (lb == 0 && tid < WAVE); \
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F); \
    const unsigned long long stamp_t0 = stamp_last; \
    unsigned long long stamp_rounds = 0;

#define PTNN_DIAG_tree_flush \
    if (stamp_on && tid == 0) { \
        for (int q_ = 0; q_ < 9; ++q_) atomicAdd(p.stamps + q_, stamp_acc[q_]); \
        atomicAdd(p.stamps + 9, stamp_rounds); \
        atomicAdd(p.stamps + 10, __builtin_amdgcn_s_memtime() - stamp_t0); \
    }
