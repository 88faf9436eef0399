/*
 * ptnn.h -- C ABI of libptnn.so: the MI355X (gfx950) parallel-tempering Bayesian-FNN sampler.
 *
 * The reference (sydney-machine-learning/parallel-tempering-neural-net) has no FFI: its seam is the Python
 * class ParallelTempering (REG = multicore-pt-regression/pt_timeseries_regression.py:487-875,
 * CLS = multicore-pt-classification/pt_classification.py:497-897).  This header is the boundary a maintainer
 * binds with ctypes to replace what that class does by forking one ptReplica process per chain; every entry
 * point names the reference code it stands in for.  INTEGRATION.md shows the binding.
 *
 * Conventions: plain C, no C++ or torch types; return 0 = OK, negative = error (text via ptnn_last_error());
 * the caller owns every host buffer (C-contiguous float32 / int32, alive for the call only); the library owns
 * device memory and its stream; a handle is not thread-safe (one host thread per handle, one handle per GPU);
 * calls are synchronous unless their comment says "asynchronous".
 */
#ifndef PTNN_H
#define PTNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTNN_ABI_VERSION 4

#define PTNN_TASK_REG 0 /* Gaussian likelihood + eta = log tau^2 (REG) */
#define PTNN_TASK_CLS 1 /* multinomial likelihood on softmax-of-sigmoid outputs (CLS) */

#define PTNN_SCHED_AUTO 0
#define PTNN_SCHED_COOPERATIVE 1
#define PTNN_SCHED_SPECULATIVE 2
#define PTNN_SCHED_PACKED 3      /* speculative, all slots of a round on one CU (n_hidden <= 8) */
#define PTNN_SCHED_TREE 4        /* prefetching: 2^D - 1 work-groups evaluate every outcome of the next D decisions (random-walk classification) */

typedef struct ptnn_handle ptnn_handle;

/* Everything ParallelTempering.__init__ / initialize_chains / ptReplica.__init__ fix for a run
 * (REG:489-527, 639-650, 140-176; CLS:499-535, 648-659, 159-193). */
typedef struct ptnn_config {
    int32_t struct_bytes;         /* = sizeof(ptnn_config): ABI guard */
    int32_t device_id;            /* HIP device ordinal */
    int32_t task;                 /* PTNN_TASK_REG | PTNN_TASK_CLS */
    int32_t n_in, n_hidden, n_out;/* topology [I, H, O] (REG:30) */
    int32_t n_replicas_local;     /* replicas (temperatures) this handle owns */
    int32_t n_replicas_global;    /* replicas in the whole ladder (== local on one GPU) */
    int32_t first_global_replica; /* global index of local replica 0 (contiguous block of the ladder) */
    int32_t n_samples;            /* S = NumSamples per replica = int(NumSample/num_chains) (REG:506) */
    int32_t swap_interval;        /* REG:496; hand-off trigger differs per task (REG:427 vs CLS:438) */
    int32_t pt_switch_step;       /* step i at which adapttemp drops to 1 (REG:320), or -1 if 0.6*S is not integral */
    int32_t use_langevin;         /* use_langevin_gradients (REG:329) */
    int32_t waves_per_replica;    /* 0 = auto; 1,2,4,8: wavefronts per work-group */
    int32_t schedule;             /* 0 = auto, 1 = cooperative (all waves share one MH step), 2 = speculative
                                   * (wave v pre-computes step i+v; identical chain), 3 = packed
                                   * speculative (n_hidden <= 8: 16 slots on one CU, SGD epochs in lane groups), 4 = prefetching
                                   * tree (random-walk classification runs: groups_per_replica = 2^D - 1 work-groups evaluate the
                                   * proposals of all outcomes of the next D accept/reject decisions, D steps per round; identical
                                   * chain; chosen automatically when replicas x 3 work-groups fit the GPU) */
    int32_t groups_per_replica;   /* speculative schedule: work-groups (CUs) cooperating on one replica; 0 = auto
                                   * (as many of 1, 2, 4 as keeps replicas x groups <= number of CUs).  Tree schedule: 3, 7, 15
                                   * or 31 (0 = auto: the deepest tree that is resident).  Wide nets (n_hidden > 64):
                                   * 1, 2 or 4 work-groups speculating over windows of steps (0 = auto: 4 or 2 where resident).
                                   * Packed schedule: 1, 2 or 4 CUs, each running a packed round over its slots of one window
                                   * (0 = auto: 4 or 2 where resident for 9 <= n_hidden <= 16, whose lane groups leave 8 slots per
                                   * CU; 1 for n_hidden <= 8, which has its 16 slots on one CU) */
    int32_t trace_capacity;       /* rows per replica kept on the device (ring); 0 = all n_samples rows.  With a smaller
                                   * value the caller drains with ptnn_get_traces at least every trace_capacity steps */
    int32_t forward_bf16;         /* forward pass on the matrix cores (cooperative schedule with 24 <= n_hidden <= 64; wide nets with
                                   * n_hidden % 32 == 0).  0 (default) = fp32 accuracy: where the split images fit, every fp32 operand
                                   * is split into three bf16 terms and the six leading partial products run on
                                   * v_mfma_f32_32x32x16_bf16 (errors of the size of fp32 rounding; not bit-identical to the VALU
                                   * schedules), else the exact v_mfma_f32_32x32x2_f32.  1 = operands ROUNDED to bf16 (wide nets only;
                                   * the precision study of BASELINE config 5, changes 0.3 % of the decisions).  2 = always the exact
                                   * fp32 instruction (k-ordered fma chains, bit-identical to the VALU schedules) */
    int32_t swap_rule;            /* 0 = the reference's cascade (REG:659-690, default); 1 = even/odd Metropolis exchange
                                   * min(1, exp((1/T_k - 1/T_k+1)(L_k+1 - L_k))) on untempered log-likelihoods, the moved state
                                   * brings its likelihood and prior along, no phantom round (not in the reference) */
    int32_t shared_noise;         /* 0 = every (replica, step) has its own Philox counter (default); 1 = all replicas read the
                                   * step tape of replica 0 (proposal noise, Langevin coin, MH uniform, eta noise): what the
                                   * reference's forked chains do, which all inherit one numpy / random state (REG:709-712,
                                   * SURVEY Q14).  Initial weights and swap uniforms are not shared (the parent draws them). */
    int32_t label_swap;           /* 0 = a swap round moves (w, eta) between the temperature slots, as the reference does (default);
                                   * 1 = label swapping (not in the reference): the chains stay in place and the
                                   * TEMPERATURES move -- a round only rewrites the slot <-> temperature maps, so nothing but the R
                                   * posted scalars crosses a GPU boundary (zero payload).  A chain keeps its own likelihood (re-tempered
                                   * for its new temperature) and prior: no stale values (Q12 does not apply).  Works with both swap
                                   * rules; needs ptnn_set_ladder.  Trace rows are recorded per chain slot: ptnn_get_labels / the swap log
                                   * say which temperature a slot held when (the Python host stitches the per-temperature files). */
    int32_t shared_device;        /* 1 = other handles or processes run on this GPU at the same time (several blocks of one ladder
                                   * rehearsing the N > 1 path on one device): automatic choices then avoid every schedule whose
                                   * work-groups wait for each other -- several work-groups per replica, the persistent launch --
                                   * because their residency cannot be guaranteed (a non-resident partner is a bounded spin and
                                   * error -5).  Explicit schedule / groups_per_replica requests are still honoured.  0 = the GPU is
                                   * this handle's alone (default). */
    int32_t reserved_;            /* keeps the floats 8-byte aligned with the seed; set 0 */
    float l_prob;                 /* langevin_prob (REG:174); CLS fixes 0.5 (CLS:192) */
    float learn_rate;             /* SGD step of langevin_gradient (REG:33) */
    float step_w;                 /* 0.025 (REG:258) */
    float step_eta;               /* 0.2   (REG:260) */
    float sigma_squared;          /* 25    (REG:273) */
    float nu_1, nu_2;             /* 0, 0  (REG:274-275) */
    uint64_t seed;                /* Philox4x32-10 key; streams are documented in DESIGN.md */
} ptnn_config;

int ptnn_abi_version(void);
/* thread-local, valid until the next failing call on this thread */
const char *ptnn_last_error(void);
/* 1 if a kernel is compiled for (task, n_in, n_out); n_hidden may be anything in [1, 512] (<= 64: one wave per SGD
 * sweep, speculative or cooperative schedule; 65..512: one thread per hidden unit, cooperative schedule) */
int ptnn_supports(int task, int n_in, int n_hidden, int n_out);

/* replaces ParallelTempering.__init__ + the construction of the ptReplica objects (REG:489, 650) */
int ptnn_create(const ptnn_config *cfg, ptnn_handle **out);
int ptnn_destroy(ptnn_handle *h);

/* traindata / testdata (REG:491-492): row-major [n, ncols] float32, columns [x_0..x_{I-1}, y, ...]; copied to HBM */
int ptnn_set_data(ptnn_handle *h, const float *train, int ntr, const float *test, int nte, int ncols);

/* w0 [R_local, P] (REG:649) and temperatures [R_local] (REG:615-636).  Also runs the chain start-up on the device:
 * eta0 = log var(fx_train(w0) - y) for REG (REG:270), initial prior and tempered likelihood (REG:280-285). */
int ptnn_set_state(ptnn_handle *h, const float *w0, const float *temperatures);

/* all R_global temperatures (needed by swap_rule 1 only) */
int ptnn_set_ladder(ptnn_handle *h, const float *temperatures_global);

/* Advances every local replica by up to n_steps MH steps (ptReplica.run loop body, REG:313-437) and performs the
 * swap rounds that fall inside (ParallelTempering.swap_procedure + round loop, REG:659-690, 719-752), including the
 * phantom last round when the chain end is reached.  n_steps < 0 = run to the end.  A handle that owns only a
 * block of the ladder (n_replicas_local < n_replicas_global) needs a communicator (ptnn_comm_init / ptnn_comm_init_host)
 * and every rank calls ptnn_run with the same n_steps: the swap rounds then exchange through it.  Asynchronous: returns
 * once the work is queued (the "boundary" exchange waits once per swap round for the permutation); ptnn_sync waits.
 * Without a communicator, and when every work-group of the grid is resident (ptnn_describe: "launches"), the whole call is ONE
 * kernel launch: the intervals and the swap rounds between them run inside it (grid barriers); otherwise one launch per swap
 * interval plus one for the round.  The chains are identical either way. */
int ptnn_run(ptnn_handle *h, int n_steps);
int ptnn_sync(ptnn_handle *h);
/* number of MH steps queued so far (0 .. S-1) */
int ptnn_steps_done(ptnn_handle *h);

/* ---- sharded ladder: one handle per GPU, each owning a contiguous, equally sized block of the temperature ladder ----
 * Replaces the reference's star topology (every replica ships [w, eta, L, T] to the parent through a multiprocessing.Queue
 * each round and blocks on an Event, REG:427-437 <-> 694-759).  Replicas are independent for a swap interval, so the only
 * exchange step is the swap round; every rank computes the identical cascade (uniforms are Philox(seed; round, pair)), hence
 * identical chains for every GPU count.  Two exchanges (ptnn_comm_set_mode):
 *   PTNN_XCHG_GATHER    one in-place all-gather per round of the exchange rows {(w, eta), cached langevin_gradient, L} of all
 *                       replicas (R_global x (8 P + 16) bytes), cascade + row copy on the device, NO host wait: pack kernel,
 *                       collective and swap kernel are queued on the handle's stream behind the segment kernel.
 *   PTNN_XCHG_BOUNDARY  all-gather of the R_global posted scalars L (4 R bytes), cascade on the device, permutation to the
 *                       host (one wait), then ONE grouped send/recv of the rows that cross a GPU boundary: per GPU at most one
 *                       (w, eta) row arrives from below (the carried state; its source may be several GPUs down: xGMI is a
 *                       full mesh, it goes there directly), at most one from the GPU above, and at most one leaves each way
 *                        -- 4 (P + 1) bytes each.  swap_rule 0 only.
 *   PTNN_XCHG_AUTO      GATHER while the gathered buffer is <= 4 MiB (every BASELINE net but the 32-512-1 one), else BOUNDARY.
 */
#define PTNN_XCHG_AUTO 0
#define PTNN_XCHG_GATHER 1
#define PTNN_XCHG_BOUNDARY 2

/* RCCL transport (xGMI on one node).  Rank 0 obtains a unique id (ncclGetUniqueId; 128 bytes) and hands it to every rank by
 * whatever rendezvous launched them; every rank then calls ptnn_comm_init(h, id, 128, rank, nranks) with
 * rank == first_global_replica / n_replicas_local (ncclCommInitRank on the handle's device: collective, blocks until all
 * ranks have joined).  librccl.so is loaded on the first call (dlopen; $PTNN_RCCL_LIBRARY overrides the path), so single-GPU
 * users never load it.  Collectives run on the handle's own stream.
 * Nothing here blocks for ever (the reference's parent polls is_alive() each round, REG:721-727): loading the library,
 * ncclGetUniqueId and ncclCommInitRank run on a helper thread that is abandoned after $PTNN_COMM_TIMEOUT_S seconds (default 120),
 * and every wait behind a collective (ptnn_sync, the getters, the boundary exchange's per-round wait) gives up when the stream
 * is busy but the device has completed no swap round for that long; all of them return -7 with the stage that stalled.
 * The library never changes the environment.  The ladder is sharded inside one node, so the bootstrap needs neither a routable
 * interface nor a verbs probe: callers that own their process export NCCL_SOCKET_IFNAME=lo and NCCL_IB_DISABLE=1 before their
 * first thread starts (the Python host does: distributed.single_node_rccl_env). */
int ptnn_comm_unique_id(void *id_out, int nbytes);
int ptnn_comm_init(ptnn_handle *h, const void *unique_id, int nbytes, int rank, int nranks);
/* One bounded RCCL round trip among `devices` (distinct) from the calling process: unique id, ncclCommInitRank on a thread per
 * device, a 4-byte all-gather, destroy; *seconds = how long it took.  Run it in a fresh CHILD process before the long-lived
 * process touches RCCL: a bring-up that fails or stalls half-way can keep the process it happened in from exiting.  0 = RCCL
 * works among these devices; -7 = it does not (ptnn_last_error names the stage). */
int ptnn_comm_probe(const int32_t *devices, int n, double *seconds);
/* what is attached to the handle: transport (0 none, 1 RCCL, 2 host-staged), this rank, the number of ranks -- for RCCL as the
 * communicator itself reports it (ncclCommCount), not as the caller passed it in -- and the handle's device */
int ptnn_comm_info(ptnn_handle *h, int32_t *transport, int32_t *rank, int32_t *nranks, int32_t *device);
/* the last stage a communicator bring-up / exchange entered in this process, as text ("ncclCommInitRank(rank 0 of 1, device 0)
 * (entered 0.4 s ago)"); $PTNN_COMM_TRACE=1 prints every stage to stderr as it is entered.  Returns the length written. */
int ptnn_comm_last_stage(char *buf, int nbytes);

/* Host-staged transport: the library stages through pinned host memory and calls back.  For fabrics other than RCCL and
 * for tests (RCCL refuses two ranks on one device; a one-GPU box rehearses the N > 1 path with this).
 *   all_gather(ctx, buf, bytes_per_rank): buf holds nranks blocks, this rank's block is filled; fill the others. 0 = OK.
 *   send_recv(ctx, n, peer[n], is_send[n], buf[n], bytes): n messages of `bytes` bytes each, in an order both ends of
 *     every pair agree on (ascending global destination slot); complete all of them before returning.  0 = OK. */
typedef int (*ptnn_all_gather_fn)(void *ctx, void *buf, int64_t bytes_per_rank);
typedef int (*ptnn_send_recv_fn)(void *ctx, int n, const int32_t *peer, const int32_t *is_send, void *const *buf, int64_t bytes);
int ptnn_comm_init_host(ptnn_handle *h, int rank, int nranks, ptnn_all_gather_fn all_gather, ptnn_send_recv_fn send_recv, void *ctx);

int ptnn_comm_set_mode(ptnn_handle *h, int mode);
/* what crossed GPU boundaries so far: payload bytes this rank sent and received, swap rounds exchanged, the mode in use */
int ptnn_comm_stats(ptnn_handle *h, int64_t *bytes_sent, int64_t *bytes_received, int64_t *rounds, int32_t *mode);
/* releases the communicator (ncclCommDestroy); ptnn_destroy does it too */
int ptnn_comm_finalize(ptnn_handle *h);

/* Pure host function (no GPU): which rows rank `rank` receives and sends for the permutation src[R_global] of one round
 * (slot k receives the state of slot src[k]) when every rank owns n_local consecutive slots.  Messages are listed in ascending
 * global destination slot.  msg[4 * m + {0,1,2,3}] = {is_send, peer rank, local row (destination row of a receive, source row
 * of a send), global destination slot}.  Returns the number of messages (<= max_msgs) or negative. */
int ptnn_route(const int32_t *src, int n_global, int n_local, int rank, int32_t *msg, int max_msgs);

/* ---- the pieces of one swap round, for callers that drive the exchange themselves ---- */
/* queue MH steps up to and including the next hand-off step (or the chain end); returns in *handoff 1 when a swap
 * round is due after it, 2 when the due round is the phantom end-of-chain round, 0 otherwise.  Asynchronous. */
int ptnn_run_segment(ptnn_handle *h, int *handoff);
/* device address of the posted scalars L[R_global] (REG:430 / CLS:439); the local block is filled by the segment,
 * the caller all-gathers the rest in place */
int ptnn_swap_L_ptr(ptnn_handle *h, int phantom, void **dev_ptr);
/* overwrite L[R_global] from the host (host-staged transports, tests) */
int ptnn_swap_set_L(ptnn_handle *h, int phantom, const float *L_host);
/* run the cascade on L[R_global] (identical on every rank: uniforms are Philox(seed; round, pair)); writes
 * src[R_global] to the host: slot k receives the (w, eta) of slot src[k] */
int ptnn_swap_cascade(ptnn_handle *h, int phantom, int32_t *src_host);
/* device addresses of the (w, eta) row of a local replica in the current (send) and next (receive) state buffers;
 * row length ptnn_state_row_floats() floats */
int ptnn_swap_row_ptr(ptnn_handle *h, int local_replica, void **cur_row, void **next_row);
int ptnn_state_row_floats(ptnn_handle *h);
/* copy the rows whose source is local, flip the buffers, count the round.  Rows with remote sources must have been
 * received into next_row before this call. */
int ptnn_swap_apply(ptnn_handle *h, const int32_t *src_host, int phantom);

/* Gathered exchange (the default of the sharded-ladder driver; replaces the Queue traffic of REG:427-437 <-> 730-752 with ONE
 * collective per swap round): ptnn_swap_pack writes, for every local replica, the exchange row
 *   { (w, eta) row | cached langevin_gradient row | its valid flag | posted L | swap_rule 1: untempered L, prior | pad }
 * into this rank's block of the buffer ptnn_xchg_ptr returns ([n_replicas_global][row_floats], same layout on every rank);
 * the caller all-gathers the buffer in place; ptnn_swap_apply_gathered then runs the cascade on the gathered L values and
 * copies every local slot's source row out of the buffer, wherever that replica ran, flips the buffers and counts the round. */
int ptnn_xchg_ptr(ptnn_handle *h, void **base, int *row_floats);
int ptnn_swap_pack(ptnn_handle *h, int phantom);
int ptnn_swap_apply_gathered(ptnn_handle *h, int phantom);

/* the HIP stream (hipStream_t) all of this handle's work is queued on: lets the caller order its collectives after the
 * segment / before the swap kernels on the device instead of synchronising the host */
int ptnn_stream(ptnn_handle *h, void **hip_stream);

/* ---- results ---- */
/* traces of rows [step0, step0+nsteps) for all local replicas (row i+1 is written by MH step i; the rows must still be in
 * the ring: step0 >= steps_done + 1 - trace_capacity); any pointer may be NULL.
 * pos_w [R,nsteps,P] (REG:240,408,417); likeh [R,nsteps] = column 0 of likeh_list (REG:391 / CLS:404);
 * rmse_* / acc_* [R,nsteps] (REG:403-423); accept_count [R,nsteps] = accept_list (REG:380). */
int ptnn_get_traces(ptnn_handle *h, int step0, int nsteps, float *pos_w, float *likeh, float *rmse_train,
                    float *rmse_test, float *acc_train, float *acc_test, int32_t *accept_count);
/* The scalar trace rows as the device keeps them, rows [R, nsteps, 8] float32: {likeh, rmse_train, rmse_test, acc_train,
 * acc_test, accept_count (int32 bits), log alpha of the step as the kernel computed it (REG:372: diff_likelihood + diff_prior +
 * diff_prop; diagnostic, the parity tests measure the fp32 error of the MH decision with it), 0}.  Regression (task 0): the
 * acc_train slot -- identically 0 in the reference (REG:403) and in ptnn_get_traces -- holds eta = log tau^2 of the recorded
 * state here (the chain's eta after an accepted step; tests set the oracle's state from it).  Does not mark rows as
 * fetched.  Same range rules as ptnn_get_traces. */
int ptnn_get_trace_rows(ptnn_handle *h, int step0, int nsteps, float *rows);
/* Trace images: the trace download overlapped with sampling (the reference's chains write their files after their last step and
 * the parent reads them back, REG:454-481, 775-871; here rows can leave while later steps are sampled).  ptnn_trace_image: pinned
 * host copies owned by the handle, in the device's layout -- pos_w [R, n_samples, *row_floats] (the first n_param floats of a row
 * are the vector), rows [R, n_samples, 8] as ptnn_get_trace_rows describes them; allocated on the first call; needs every row
 * resident (trace_capacity 0) and non-compact traces.  ptnn_trace_image_fetch: behind everything queued on the handle so far
 * (ptnn_run returns once the steps are queued), copies rows [step0, step0+nsteps) of every local replica into the images on a
 * second stream and returns a ticket >= 0; marks the rows as fetched.  ptnn_trace_image_wait: blocks until that copy has landed.
 * A run's errors surface at ptnn_sync as always.  ptnn_set_state (a restart) forgets all tickets. */
int ptnn_trace_image(ptnn_handle *h, float **pos_w, int32_t *row_floats, float **rows);
int ptnn_trace_image_fetch(ptnn_handle *h, int step0, int nsteps);
int ptnn_trace_image_wait(ptnn_handle *h, int ticket);
/* num_swap / total_swap_proposals (REG:501-502, 680-688) */
int ptnn_get_swap_stats(ptnn_handle *h, int64_t *num_swap, int64_t *total_proposals, int32_t *rounds_done);
/* src permutation of every completed round, [rounds, R_global] (tests) */
int ptnn_get_swap_log(ptnn_handle *h, int32_t *src, int max_rounds);
/* current chain state per local replica: w [R,P], eta [R], likelihood [R] (tempered, possibly stale: Q12),
 * prior_current [R], num_accepted [R], langevin_count [R] (Langevin steps proposed, REG:347), langevin_accepted [R]
 * (of those, accepted; speculative schedule only); any pointer may be NULL */
int ptnn_get_state(ptnn_handle *h, float *w, float *eta, float *likelihood, float *prior, int32_t *num_accepted,
                   int32_t *langevin_count, int32_t *langevin_accepted);

/* label_swap = 1: label[R_global] = the temperature index every chain slot of the whole ladder holds after the rounds queued so
 * far (identity without label swapping) */
int ptnn_get_labels(ptnn_handle *h, int32_t *label);

/* ---- checkpoint / resume (the reference has none) ----
 * The RNG is counter based, so the state of the chains is small: (w, eta), cached gradient, recorded row, likelihood / prior /
 * counters per replica, posted scalars, swap counters and log, step and round indices.  ptnn_checkpoint_save writes it into a
 * caller buffer of ptnn_checkpoint_size bytes; ptnn_checkpoint_load on a handle created with the same chain configuration
 * (after ptnn_set_data, instead of ptnn_set_state) continues the chains bit for bit.  Trace rows are not part of it: rows up
 * to the checkpoint step stay with whoever fetched them, and ptnn_get_traces of the restored handle refuses them. */
int ptnn_checkpoint_size(ptnn_handle *h, int64_t *bytes);
int ptnn_checkpoint_save(ptnn_handle *h, void *buf, int64_t bytes);
int ptnn_checkpoint_load(ptnn_handle *h, const void *buf, int64_t bytes);

/* ---- the model functions on their own (same device code as the sampler) ---- */
/* Network.evaluate_proposal + likelihood_func + prior_likelihood for n weight vectors w [n,P] (REG:120-134, 200-221;
 * CLS:134-153, 209-230); tau_sq [n] (ignored for CLS, may be NULL).  out [n,8] =
 * {loglik_train (untempered), rmse_train, rmse_test, acc_train, acc_test, prior, loglik_test, 0}. */
int ptnn_evaluate(ptnn_handle *h, const float *w, const float *tau_sq, int n, float *out);
/* Network.langevin_gradient(train, w, depth=1) for n weight vectors (REG:99-118, CLS:114-132) */
int ptnn_langevin_gradient(ptnn_handle *h, const float *w_in, int n, float *w_out);
/* What ONE sequential SGD epoch (langevin_gradient of one chain, REG:99-118) costs on this device, in milliseconds: `reps` epochs
 * back to back on one wavefront, timed inside the kernel with the constant-rate counter (s_memrealtime; the rate comes from
 * hipDeviceAttributeWallClockRate).  An accepted Langevin step makes the next proposal wait for a fresh epoch, so accepted steps x
 * this number is the floor of a swap interval whatever the number of speculative slots (bench.py: roofline.chain).
 * ms_per_epoch[2]: [0] one epoch; [1] wide nets (n_hidden > 64) only: a PAIR of epochs run through one row loop (what two Langevin
 * steps of one speculative window cost together), else 0. */
int ptnn_time_sgd_epoch(ptnn_handle *h, const float *w, int reps, double *ms_per_epoch);
/* What a round of the prefetching-tree schedule cannot do without, timed on the device (in-kernel constant-rate counter, `reps`
 * repetitions): ms[0] = one forward pass over all rows with the likelihood and prior sums by one work-group of the handle's block
 * size (what a node does between its proposal and its record); ms[1] = one {tag, value} granule from one work-group to another,
 * one way (half a round trip between two work-groups of one XCD), through the XCD's L2 when xcd_local != 0 and the two groups
 * report the same XCC id (ms[2] = 1), else through the agent-scope path.  bench.py's roofline.tree floor is made of these. */
int ptnn_time_tree_round(ptnn_handle *h, const float *w, int reps, int xcd_local, double *ms);
/* the random tape of MH step `step` of global replica `replica`: noise [P] normals, scal[3] = {lx, u, n_eta} */
int ptnn_tape(ptnn_handle *h, int replica, int step, float *noise, float *scal);

/* What this handle will launch, as one line of JSON text written into buf (returns its length, or negative): the segment
 * kernel the schedule resolved to for this topology / data set / replica count, its grid, block and dynamic LDS size, the
 * work-groups per replica and speculative slots per round, and what the runtime reports for it (blocks per CU from
 * hipOccupancyMaxActiveBlocksPerMultiprocessor, VGPRs, scratch bytes).  After ptnn_set_data.  The reference has no
 * counterpart (its "schedule" is one OS process per chain, REG:709-712); bench.py and the profiles name kernels with it. */
int ptnn_describe(ptnn_handle *h, char *buf, int nbytes);

/* timing of the dominant kernel, measured with HIP events on the library's stream around every segment launch
 * since the last reset: launches, total milliseconds */
int ptnn_kernel_time(ptnn_handle *h, int reset, int64_t *launches, double *total_ms);

/* cycle sums per phase of the speculative kernel, replica 0 / wave 0; all zero unless the library was built with
 * -DPTNN_STAMPS (diagnostic build, never the product).  160 entries: [0..8] phase sums, [9] rounds, [16+2r], [17+2r] =
 * cycles and rounds of replica r < 64; reading resets. */
int ptnn_debug_stamps(ptnn_handle *h, uint64_t *out16);

/* ---- host-side helper (no GPU): the text dump the result-file layout requires ---- */
/* np.savetxt(path, data[rows, cols], fmt=fmt) with ' ' between columns and '\n' after rows (REG:454-481, 864-868).
 * fmt is one printf floating conversion such as "%.18e", "%1.8f", "%1.2f". */
int ptnn_savetxt(const char *path, const double *data, int64_t rows, int64_t cols, const char *fmt);
/* the same for float32 data as the device's traces are fetched (ptnn_get_traces): row r starts at data + r * row_stride; every
 * value is printed as the double it converts to, exactly as np.savetxt prints a float32 array; append != 0 continues an existing
 * file (a run written in windows).  Both savetxt entry points produce np.savetxt's bytes: values are formatted by exact integer
 * arithmetic (correctly rounded, ties to even, as glibc's printf), printf itself only for formats or magnitudes outside that
 * path, and a row identical to the one before it (a rejected MH step: pos_w[i+1] = pos_w[i], REG:417) reuses that row's text. */
int ptnn_savetxt_f32(const char *path, const float *data, int64_t rows, int64_t cols, int64_t row_stride, const char *fmt, int append);
/* n_files such files (the per-chain files of a window of trace rows, REG:454-481) by `threads` host threads, taken in the order
 * given (put the large ones first); returns when all are written, < 0 with the first failure's message */
int ptnn_savetxt_f32_batch(int n_files, const char *const *paths, const float *const *data, const int64_t *rows, const int64_t *cols,
                           const int64_t *row_stride, const char *const *fmts, int append, int threads);

/* in place: every value as np.loadtxt would read it back after np.savetxt(fmt=fmt) (show_results re-reads the per-chain
 * files, REG:795-831) */
int ptnn_text_round(double *values, int64_t n, const char *fmt);
int ptnn_text_round_f32(const float *in, double *out, int64_t n, const char *fmt);
/* out[p][c * m + t] = pos_w[c][first_row + t][p] with m = n_rows - first_row, as float64: the posterior matrix show_results
 * returns (REG:795-797, 848: every chain's pos_w file read back, burn-in cut, chains side by side, transposed).
 * pos_w [n_chains, n_rows, row_floats >= n_param] float32 (row_floats == n_param as ptnn_get_traces delivers it, the padded row of
 * ptnn_trace_image); out [n_param, n_chains * m]; `threads` host threads. */
int ptnn_posterior_matrix(const float *pos_w, int64_t n_chains, int64_t n_rows, int64_t n_param, int64_t row_floats, int64_t first_row,
                          double *out, int threads);

#ifdef __cplusplus
}
#endif
#endif /* PTNN_H */
