/*
 * libfsmg -- C-ABI of the MI355X-native LSTM-baseline episodic train / eval step.
 *
 * This is the drop-in boundary for the hot path of AI-ON/Few-Shot-Music-Generation
 * (reference tree: /root/reference).  The reference has no FFI of its own: its only
 * device boundary is `self._sess.run(...)` inside the `models/` plugin
 * (src/models/lstm_baseline.py:104,125,150).  The entry points below are what a
 * Python `models.lstm_baseline.LSTMBaseline` plugin binds with ctypes instead of a
 * TensorFlow session; each one names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, opaque handle, caller-allocated buffers, no C++ exceptions cross the boundary;
 *   - every function returns 0 on success or a negative FSMG_ERR_* code;
 *     fsmg_last_error(handle) (or fsmg_last_error(NULL) for create-time errors) has the text;
 *   - a handle is NOT thread-safe; all work of a handle is ordered on ONE HIP stream
 *     (its own, or the caller's when fsmg_config.stream is set);
 *   - "host" pointers are ordinary CPU memory, "device" pointers are HBM addresses on the
 *     handle's device (e.g. torch.Tensor.data_ptr()); token arrays are int32, C-contiguous,
 *     ids in [0, input_size) exactly as the reference's Episode.support / Episode.query
 *     (src/data/episode.py:63-74);
 *   - parameters cross the boundary in the REFERENCE's variable layout (tf names and
 *     shapes, src/models/lstm_baseline.py:39-40,44-49,60-62): embedding [V1,E],
 *     kernel_<l> [(in+H),4H] with gate column blocks i,j,f,o, bias_<l> [4H],
 *     softmax_w [H,V1], softmax_b [V1];  V1 = input_size + 1 (start word).
 *     Inside, they live padded and gate-interleaved (DESIGN.md "HBM layout").
 *   - there is no CPU fallback: without a gfx950 device fsmg_create fails with
 *     FSMG_ERR_NO_DEVICE.
 */
#ifndef FSMG_H
#define FSMG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSMG_VERSION 600 /* 0.6.0 */
/* layout of struct fsmg_config AND of every other struct / flat-buffer layout a caller may hold (struct fsmg_stats, the padded
 * sizes fsmg_debug_dims reports): fsmg_create rejects any other value in .config_version, so a caller built against an older
 * header fails at create time instead of being overrun later.  4 (0.5.0): fsmg_stats grew (steps_skipped_peer_failure in 0.4.0,
 * xov_selfcheck_mismatches now); hidden sizes that no persistent kernel takes at a multiple of 16 pad to a multiple of 64
 * (200 -> 256, not 208: every padded offset behind fsmg_debug_read / fsmg_debug_dims moved in 0.4.0 without a version bump). */
#define FSMG_CONFIG_VERSION 4

enum {
    FSMG_OK = 0,
    FSMG_ERR_INVALID = -1,     /* bad argument / config value                        */
    FSMG_ERR_NO_DEVICE = -2,   /* no usable HIP device (no CPU fallback exists)      */
    FSMG_ERR_HIP = -3,         /* a HIP runtime call failed                          */
    FSMG_ERR_NOMEM = -4,       /* host or device allocation failed                   */
    FSMG_ERR_NAME = -5,        /* unknown parameter / buffer name                    */
    FSMG_ERR_SIZE = -6,        /* element count does not match the named tensor      */
    FSMG_ERR_TOKEN_RANGE = -7, /* a token id was outside [0, input_size)             */
    FSMG_ERR_STATE = -8,       /* call sequence error (e.g. apply without backward)  */
    /* the step was SKIPPED on the device (parameters, Adam state and global_step untouched) and the handle has changed how it runs
     * the next one: repeat the call.  fsmg_train_step / _indexed / fsmg_maml_* repeat it themselves when a loss is read back. */
    FSMG_ERR_TIMEOUT = -9,         /* a persistent kernel's hand-off or gate timed out (its blocks were not co-resident, or two launches that must
                                      run side by side did not): the handle runs one launch per time step for `fallback_steps` steps */
    FSMG_ERR_SOFTMAX_RANGE = -10   /* a row's sum of exp(logit) or its target's exp(logit) left the fp32 range of the shift-free fused softmax:
                                      the handle takes the cross-entropy pass with the shifted softmax from here on (nothing else changes) */
};

enum { FSMG_CLIP_TF1_SLICES = 0, FSMG_CLIP_DENSE = 1 };
/* arithmetic of the dense contractions: AUTO = BX3 (every operand split exactly into three bf16 pieces, the six partial products
 * >= 2^-23 of the fp32 product summed on the bf16 matrix pipe: each product to within half an fp32 ulp, fp32 accumulation:
 * DESIGN.md section 4); F32 = v_mfma_f32_32x32x2_f32 */
enum { FSMG_GEMM_AUTO = 0, FSMG_GEMM_BX3 = 1, FSMG_GEMM_F32 = 2 };
/* order of a pass: AUTO picks from the shapes; SINGLE_STREAM = one stream, serial; TWO_STREAM = projection GEMMs on an
 * auxiliary stream beside the recurrence (eager); XCD_PARTITIONED = the bf16-split recurrence packed on ceil(rows / 16) XCDs, the
 * projection / its weight gradient as work-queue GEMMs on the other XCDs beside it (hidden 512, one layer; AUTO picks it there) */
enum { FSMG_SCHEDULE_AUTO = 0, FSMG_SCHEDULE_SINGLE_STREAM = 1, FSMG_SCHEDULE_TWO_STREAM = 2, FSMG_SCHEDULE_XCD_PARTITIONED = 3 };
/* recurrent kernels: AUTO = the fastest family the shape admits; PER_STEP = one launch per time step; COLUMN_SPLIT = persistent,
 * gate columns over the chip (round 1); XCD_LOCAL = persistent, rows over the XCDs (hidden size 512) */
enum { FSMG_RECURRENCE_AUTO = 0, FSMG_RECURRENCE_PER_STEP = 1, FSMG_RECURRENCE_COLUMN_SPLIT = 2, FSMG_RECURRENCE_XCD_LOCAL = 3 };

typedef struct fsmg_model* fsmg_handle;

/* Model / optimiser configuration == the YAML keys the reference plugin reads
 * (src/models/lstm_baseline.py:21-29,80; src/models/tf_model.py:81). */
typedef struct fsmg_config {
    int32_t input_size;      /* config['input_size']: vocabulary WITHOUT the start word        */
    int32_t max_len;         /* config['max_len'] = T                                         */
    int32_t embedding_size;  /* config['embedding_size'] = E                                  */
    int32_t hidden_size;     /* config['hidden_size'] = H                                     */
    int32_t n_layers;        /* config['n_layers'] = L                                        */
    float lr;                /* config['lr']                                                  */
    float max_grad_norm;     /* config['max_grad_norm']                                       */
    float n_decay;           /* config['n_decay'] (lr halves every n_decay steps, continuous) */
    int32_t clip_norm_mode;  /* FSMG_CLIP_TF1_SLICES (reference behaviour) or FSMG_CLIP_DENSE */
    int32_t device;          /* HIP device ordinal                                            */
    int32_t max_sequences;   /* sequences per training episode (N*(K+Q)): initial activation capacity (grows on demand)
                                and the row count the recurrent kernel family is chosen for (hidden 512, > 64 rows: the
                                bf16-split XCD-local kernels); 0 = 45 */
    int32_t use_graph;       /* 1: replay the per-timestep launch chains as hipGraphs         */
    void* stream;            /* optional caller hipStream_t; NULL = the library creates one   */
    void* state_arena;       /* optional caller-owned DEVICE memory for params+grads+Adam state
                                (fsmg_state_bytes() bytes, 256-B aligned); NULL = hipMalloc  */
    uint64_t state_arena_bytes;
    /* ---- since config version 3: what used to be environment variables read at create time.  The variables still exist as
     * debugging overrides (FSMG_GEMM, FSMG_OVERLAP, FSMG_XCD_OVERLAP, FSMG_PERSISTENT, FSMG_XCD, FSMG_DP_SPLIT, ...: they win) */
    int32_t config_version;     /* must be FSMG_CONFIG_VERSION: a caller built against another header is refused            */
    int32_t gemm;               /* FSMG_GEMM_*                                                                               */
    int32_t schedule;           /* FSMG_SCHEDULE_*                                                                           */
    int32_t recurrence;         /* FSMG_RECURRENCE_*                                                                         */
    int32_t dp_split_backward;  /* 1: fsmg_forward_backward replays two graphs and bucket 0 of the gradient exchange
                                   (softmax gradients) is final behind the first one (fsmg_stream_wait_bucket), which ends
                                   behind the projection gradients; 2: the first graph ends behind the last recurrent
                                   chain instead (a collective started before a chain that needs every CU only delays it)   */
    int32_t reserved[7];        /* zero                                                                                       */
} fsmg_config;

/* ---- lifetime -------------------------------------------------------------------------- */
int fsmg_version(void);
const char* fsmg_last_error(fsmg_handle h);
/* bytes a caller-provided state arena must have for this config */
uint64_t fsmg_state_bytes(const fsmg_config* cfg);
/* replaces TFModel.__init__ (session + graph build, src/models/tf_model.py:80-97) */
int fsmg_create(const fsmg_config* cfg, fsmg_handle* out);
int fsmg_destroy(fsmg_handle h);
int fsmg_synchronize(fsmg_handle h);

/* ---- parameters, optimiser state, checkpoints ------------------------------------------ */
/* replaces the variable initialisation of recover_or_init (src/models/tf_model.py:16-25,127-129):
 * Glorot-uniform for every matrix AND softmax_b, zeros for LSTM biases;
 * also zeroes Adam m/v and global_step. */
int fsmg_init_params(fsmg_handle h, uint64_t seed);
int fsmg_num_params(fsmg_handle h);
/* name (tf variable name without scope), rows, cols (cols = 1 for vectors) of parameter idx */
int fsmg_param_info(fsmg_handle h, int idx, char* name, int name_cap, int64_t* rows, int64_t* cols);
/* replace Saver.restore / Saver.save of a variable (src/models/tf_model.py:96-114); host
 * float32 buffers in the reference layout, count = rows*cols */
int fsmg_set_param(fsmg_handle h, const char* name, const float* host, int64_t count);
int fsmg_get_param(fsmg_handle h, const char* name, float* host, int64_t count);
/* Adam slots of a variable (what a TF checkpoint holds besides weights) */
int fsmg_set_opt_state(fsmg_handle h, const char* name, const float* m, const float* v, int64_t count);
int fsmg_get_opt_state(fsmg_handle h, const char* name, float* m, float* v, int64_t count);
int fsmg_set_step(fsmg_handle h, int64_t global_step);
int fsmg_get_step(fsmg_handle h, int64_t* global_step);
/* gradient of the last backward, reference layout, before clipping (parity tests) */
int fsmg_get_grad(fsmg_handle h, const char* name, float* host, int64_t count);

/* ---- the hot path ---------------------------------------------------------------------- */
/* replaces LSTMBaseline.train (src/models/lstm_baseline.py:89-113): support [N,K,T] and
 * query [N,Q,T] are flattened support-rows-first, shifted against the start word on the
 * device, forward + BPTT + clip_by_global_norm + Adam + global_step++.  *loss receives the
 * mean NLL computed with the PRE-update parameters; passing loss == NULL skips the
 * device->host readback (the loss stays in the handle's ring, see fsmg_read_losses).
 * tokens_on_device != 0: support/query are device pointers (episode pool resident in HBM). */
int fsmg_train_step(fsmg_handle h, const int32_t* support, const int32_t* query,
                    int32_t N, int32_t K, int32_t Q, int32_t tokens_on_device, float* loss);

/* Episode-parallel form of the same step: forward + backward only; the flat fp32
 * gradient buffer (fsmg_grad_buffer) is then summed across ranks by the host (one RCCL
 * all-reduce), and fsmg_apply_update(grad_scale = 1/world) clips and applies Adam identically
 * on every rank. */
int fsmg_forward_backward(fsmg_handle h, const int32_t* support, const int32_t* query,
                          int32_t N, int32_t K, int32_t Q, int32_t tokens_on_device);
/* device address + element count of the flat gradient buffer; the last FSMG_GRAD_TAIL floats
 * are scalars that must be reduced with it: [0] = sum of squared embedding-slice gradients,
 * [1] = loss, [2] = non-zero when a persistent recurrent kernel of this rank timed out (its gradients are garbage):
 * fsmg_apply_update then leaves parameters, Adam state and step counter alone on every rank and, when it reads the
 * loss back, returns FSMG_ERR_TIMEOUT after switching the handle to one launch per time step -- repeat the step;
 * [3] = non-zero when this rank's batch held a token id outside [0, input_size): summed like [2], so EVERY rank skips the
 * update (replicas stay identical) and every rank's read-back returns FSMG_ERR_TOKEN_RANGE;
 * [4] = non-zero when this rank's pass failed on the host before the exchange (library-owned exchange: the rank still joins
 * the collectives so that no peer blocks): every rank skips the update, peers' read-backs return FSMG_ERR_STATE;
 * [5] = non-zero when a row of this rank's logits left the range of the shift-free fused softmax: summed like [2], every rank skips
 * the update, switches to the cross-entropy pass with the shifted softmax and returns FSMG_ERR_SOFTMAX_RANGE -- repeat the step */
#define FSMG_GRAD_TAIL 16
int fsmg_grad_buffer(fsmg_handle h, void** device_ptr, int64_t* count);
int fsmg_apply_update(fsmg_handle h, float grad_scale, float* loss);
/* Bucketed form of the gradient exchange, so that communication overlaps the rest of the backward pass:
 *   bucket 0 = softmax_w + softmax_b gradients (final as soon as the dW GEMM retires, 56 % of the bytes at cfg-B)
 *   bucket 1 = every other gradient, bucket 2 = the FSMG_GRAD_TAIL scalars (both final when backward ends).
 * fsmg_grad_bucket returns the device range of a bucket; fsmg_stream_wait_bucket makes `stream` (the caller's
 * communication hipStream_t) wait until that bucket of the LAST fsmg_forward_backward is final. */
#define FSMG_NUM_BUCKETS 3
int fsmg_grad_bucket(fsmg_handle h, int32_t bucket, void** device_ptr, int64_t* count);
int fsmg_stream_wait_bucket(fsmg_handle h, void* stream, int32_t bucket);

/* The gradient exchange INSIDE the library ("fsmg_allreduce_grads internal to train_step when world > 1").
 * With a communicator attached, fsmg_train_step / fsmg_train_step_indexed / fsmg_maml_step are the episode-parallel step by
 * themselves: forward + backward graph(s) -> ncclAllReduce(SUM) of the three gradient buckets on the library's own
 * communication stream (bucket 0 as soon as the projection gradients are final when fsmg_config.dp_split_backward is set) ->
 * clip + Adam with grad_scale = 1 / world_size; the caller (PyTorch or anything else) only owns the memory.  RCCL is looked up
 * at run time (dlopen of the librccl.so already in the process, else /opt/rocm/lib): no link-time dependency.
 *   fsmg_comm_unique_id        rank 0: a fresh ncclUniqueId to hand to the other ranks by whatever means the job has;
 *   fsmg_comm_init             every rank: ncclCommInitRank on the handle's device (a collective call);
 *   fsmg_comm_attach           instead of the two above: an ncclComm_t the caller already owns (never destroyed by the library);
 *   fsmg_comm_broadcast_state  parameters, Adam slots and global_step of rank `root` to every rank (after init / restore);
 *   fsmg_comm_release          detach (and destroy a communicator the library created).
 * fsmg_forward_backward / fsmg_apply_update stay for callers that own the exchange. */
#define FSMG_COMM_ID_BYTES 128
int fsmg_comm_unique_id(char id[FSMG_COMM_ID_BYTES]);
int fsmg_comm_init(fsmg_handle h, const char id[FSMG_COMM_ID_BYTES], int32_t world_size, int32_t rank);
int fsmg_comm_attach(fsmg_handle h, void* nccl_comm, int32_t world_size, int32_t rank);
int fsmg_comm_broadcast_state(fsmg_handle h, int32_t root);
int fsmg_comm_release(fsmg_handle h);

/* Device-resident episode table.  The reference fills an episode from a host cache of per-song rows
 * (src/data/episode.py:62-74, src/data/dataset.py:187-199); here the packed split -- int32 [n_songs][max_len], the
 * `.npy` sidecars of the split in artist/song order -- is uploaded once and an episode is N*K + N*Q ROW INDICES gathered on
 * the GPU.  table_id in [0, 4) (e.g. 0 train, 1 val, 2 test).  An index outside [0, n_songs) -> FSMG_ERR_TOKEN_RANGE. */
int fsmg_upload_table(fsmg_handle h, int32_t table_id, const int32_t* host_table, int64_t n_songs);
int fsmg_forward_backward_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                                  int32_t N, int32_t K, int32_t Q);
int fsmg_train_step_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                            int32_t N, int32_t K, int32_t Q, float* loss);

/* replaces LSTMBaseline.eval (src/models/lstm_baseline.py:115-133): query-only mean NLL,
 * no state change. */
int fsmg_eval_step(fsmg_handle h, const int32_t* query, int32_t N, int32_t Q,
                   int32_t tokens_on_device, float* nll);
/* n_episodes independent eval calls in one pass: queries [n_episodes,N,Q,T] -> nll[n_episodes]
 * (what train.evaluate's loop over model.eval computes, src/train/train.py:27-33) */
int fsmg_eval_batch(fsmg_handle h, const int32_t* queries, int32_t n_episodes, int32_t N, int32_t Q,
                    int32_t tokens_on_device, float* nll);

/* ---- cfg-E (BASELINE.json configs[4]): MAML-style inner / outer loop, first order.
 * The reference has no code for it (READING_LIST.md:5-7 names the direction); semantics: DESIGN.md "cfg-E", oracle:
 * oracle/lstm_oracle.py maml_step / maml_eval.  theta' starts at theta and takes `inner_steps` steps
 *   theta' <- theta' - inner_lr * clip_by_global_norm(grad of the SUPPORT rows' mean NLL at theta', max_grad_norm);
 * the outer gradient is the gradient of the QUERY rows' mean NLL with respect to theta' (no derivative through the inner
 * steps); theta is restored before anything is updated.
 *   fsmg_maml_forward_backward  leaves that gradient (+ the query loss in the tail) in the flat gradient buffer, theta and
 *                               Adam state untouched: all-reduce it across ranks, then fsmg_apply_update(1/world);
 *   fsmg_maml_step              = the two on one GPU (clip + Adam on theta, global_step++); *loss = query NLL at theta';
 *   fsmg_maml_eval              few-shot evaluation: adapt on the support set, query NLL at theta', no state change. */
int fsmg_maml_forward_backward(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K,
                               int32_t Q, int32_t inner_steps, float inner_lr, int32_t tokens_on_device);
int fsmg_maml_step(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                   int32_t inner_steps, float inner_lr, int32_t tokens_on_device, float* loss);
int fsmg_maml_eval(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                   int32_t inner_steps, float inner_lr, int32_t tokens_on_device, float* nll);
/* the same two on the device-resident split table (fsmg_upload_table): the episode is N*K + N*Q ROW INDICES (host int32) */
int fsmg_maml_forward_backward_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                                       int32_t N, int32_t K, int32_t Q, int32_t inner_steps, float inner_lr);
int fsmg_maml_step_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                           int32_t N, int32_t K, int32_t Q, int32_t inner_steps, float inner_lr, float* loss);

/* replaces LSTMBaseline.sample (src/models/lstm_baseline.py:135-156): greedy argmax decode of
 * `num` tokens from the start word and a zero state (the support set is ignored there). */
int fsmg_sample(fsmg_handle h, int32_t num, int32_t* out_tokens);

/* ---- unigram baseline.  Replaces the graph of UnigramModel (src/models/unigram_model.py:26-39): a
 * word_count variable initialised to alpha = 1, tf.scatter_add of ones, prob = gather(word_count) / reduce_sum(word_count),
 * loss = -mean(log prob).  Counts live on the device as unsigned integers (exact, order-independent atomics) and cross the
 * boundary as float32 like the reference's variable.  `words` is a flat int32 array (host, or device when on_device != 0) of ids
 * in [0, input_size); an id outside -> FSMG_ERR_TOKEN_RANGE (counts untouched by that call's NLL, the update skips the id). */
typedef struct fsmg_unigram* fsmg_unigram_handle;
int fsmg_unigram_create(int32_t input_size, int32_t device, fsmg_unigram_handle* out);
int fsmg_unigram_destroy(fsmg_unigram_handle u);
const char* fsmg_unigram_last_error(fsmg_unigram_handle u);     /* NULL: the text of a failed fsmg_unigram_create */
/* *nll = -mean(log(count[w] / sum(counts))) over the n words, counts as they are (unigram_model.py:35-37) */
int fsmg_unigram_nll(fsmg_unigram_handle u, const int32_t* words, int64_t n, int32_t on_device, float* nll);
/* counts[w] += 1 per word (unigram_model.py:31-33); with loss != NULL the NLL of the same words BEFORE the update is returned
 * (UnigramModel.train fetches both in one sess.run) */
int fsmg_unigram_train(fsmg_unigram_handle u, const int32_t* words, int64_t n, int32_t on_device, float* loss);
int fsmg_unigram_get_counts(fsmg_unigram_handle u, float* host, int64_t count);
int fsmg_unigram_set_counts(fsmg_unigram_handle u, const float* host, int64_t count);
/* argmax of the counts, lowest id on ties (UnigramModel.sample, unigram_model.py:71-78) */
int fsmg_unigram_argmax(fsmg_unigram_handle u, int32_t* word);

/* last n train losses (oldest first), n <= 1024; synchronises the stream */
int fsmg_read_losses(fsmg_handle h, float* out, int32_t n);

/* What the handle has been doing (synchronises the stream).  A train step whose persistent recurrent kernel timed out
 * (its blocks were not co-resident: another workload held the CUs) or whose batch held an out-of-range token is SKIPPED
 * on the device -- parameters, Adam state, global_step untouched -- and tallied here; after a time-out the handle runs
 * `fallback` train steps with one launch per time step and then tries the persistent kernels again.  With loss != NULL
 * fsmg_train_step repeats a timed-out step itself; with loss == NULL the episode stays skipped, so a throughput loop
 * must compare fsmg_get_step / these counters with the number of steps it issued (bench.py does). */
typedef struct fsmg_stats {
    int64_t timeouts;                   /* time-outs noticed by the host (each one starts a fallback period)        */
    int64_t steps_skipped_timeout;      /* train steps the device skipped because of a time-out (own or a peer rank) */
    int64_t steps_skipped_token_range;  /* train steps the device skipped because a token id was out of range        */
    int64_t xcd_launches;               /* launches of the XCD-local persistent kernels (hidden size 512)            */
    int64_t persistent_launches;        /* launches of the column-split persistent kernels                           */
    int64_t step_launches;              /* one-launch-per-time-step recurrent launches                               */
    int32_t persistent_path;            /* 1: persistent kernels are in force right now                              */
    int32_t fallback_steps_left;
    int64_t steps_skipped_peer_failure; /* train steps every rank skipped because one rank failed before the exchange (library-owned exchange) */
    int64_t xov_selfcheck_mismatches;   /* XCD-partitioned order: 16-byte words of the gated projection's logits that differed from the
                                           same GEMM recomputed on the serial path (the self-check of a handle's first passes);
                                           non-zero = that step was skipped and repeated, the handle keeps the serial order        */
    int64_t softmax_range_rows;         /* rows outside the range of the shift-free fused softmax (sum_v exp(logit) within [e^-60, 1e30],
                                           exp(target logit) >= 1e-30): the step that held them was skipped and repeated with the
                                           cross-entropy pass, which the handle keeps from then on */
    int64_t steps_skipped_softmax_range; /* train steps the device skipped for that reason (own rows or a peer rank's: the indicator travels in
                                           the reduced gradient tail, so every rank of a data-parallel job switches in the same step) */
    int32_t aux_stream_tries;           /* second streams fsmg_create drew until one ran BESIDE the handle's own (the process's streams share a few
                                           hardware queues); -1: none did -- the XCD-partitioned / two-stream orders are off for this handle, it runs
                                           the serial order (slower, same results); fsmg_debug_set("reprobe_aux", 1) probes again */
    int32_t reserved0;
} fsmg_stats;
int fsmg_get_stats(fsmg_handle h, fsmg_stats* out);

/* ---- introspection for kernel-level parity tests and bench.py --------------------------- */
/* copy an internal activation buffer of the last forward to the host, float32:
 *   "h<l>" [T+1,B,Hp] (index 0 = zero state), "c<l>" [T+1,B,Hp], "gates<l>" [T,B,4Hp] (packed
 *   gate order, holds dz after a backward), "logits" [T*B,V1p], "lse" [T*B], "ce" [T*B];
 *   rows are TIME-major (row = t*B + b).  count = elements to copy (<= buffer size).
 *   "xcd_bx3" [1]: 1.0 when the handle runs the bf16-split XCD-local recurrent kernels (hidden 512: created for > 64 rows, or for
 *   the XCD-partitioned schedule); "aux_tries" [1]: second streams fsmg_create drew until one ran BESIDE the handle's stream (a process's
 *   streams share GPU_MAX_HW_QUEUES hardware queues; -1: none did and the handle keeps the serial order); "xcd_partitioned" [3]: 1.0 when train passes take the XCD-partitioned order, XCDs the chains occupy, whether the LAST pass took it */
int fsmg_debug_read(fsmg_handle h, const char* what, float* host, int64_t count);
/* run-time knobs of a handle that used to be create-time environment variables (tests, diagnostics):
 *   "chain_spin_limit"  polls before a persistent recurrent kernel gives up (0 forces the time-out path)
 *   "fallback_steps"    train steps on per-step launches after a time-out before the persistent path is tried again
 *   "persistent"        0: one launch per time step instead of the persistent recurrent kernels, 1: back (buffers permitting)
 *   "eager"             0: passes are replayed from hipGraphs wherever fsmg_config.use_graph allows, 1: passes on the persistent
 *                       recurrent kernels are issued eagerly (default)
 *   "inplace_dlogits"   1 (default): a train pass's cross entropy writes dlogits over the logits it has just read ("logits" then
 *                       reads back as dlogits after a train pass), 0: two buffers
 *   "fused_softmax"     1 (default): train passes whose projection weight gradient runs on the 256 x 256-tile kernel never materialise
 *                       dlogits: the projection stores exp(logit), one kernel per pass derives lse / loss / row scales, the two GEMMs
 *                       apply them ("logits" / "dlogits" then read back exp(logit) with the target element reduced by the row sum);
 *                       0: the cross-entropy pass
 *   "upd_split"         1: clip + Adam of an eager pass as two launches, the softmax half on the auxiliary stream beside the next
 *                       step's input phase (bit-identical; measured slower), 0 (default): one launch
 *   "tail_aside"        1 (default): the bandwidth-bound tail of an eager backward pass (deferred slab sums, embedding gradient) on
 *                       the auxiliary stream beside the bottom layer's weight-gradient GEMM (bit-identical), 0: in line
 *   "xov_selfcheck"     XCD-partitioned order: the next `value` train passes recompute the gated projection on the serial path and
 *                       compare the words (default: the first 2 passes of a handle)
 *   "xov_selfcheck_fault" 1: the comparison runs against a buffer that is NOT the recomputed logits (tests of the recovery path)
 *   "xov_selfcheck_every" XCD-partitioned order: besides the first passes, one pass in every `value` is checked the same way for the handle's
 *                       whole life (default 1000: 0.03 % of the training time; 0: never again).  fsmg_debug_read("xov_selfcheck", 3) = [passes
 *                       checked so far, passes that took the order, the period]
 *   "reprobe_aux"       1: a handle whose create-time probe found no second stream running beside its own (fsmg_stats.aux_stream_tries = -1)
 *                       probes again (300 us, then 5 ms per candidate); on success the overlapped tails -- and the XCD-partitioned order where the
 *                       handle was created in the format it needs -- come back
 * Synchronises the stream and drops the captured graphs. */
int fsmg_debug_set(fsmg_handle h, const char* what, int64_t value);
/* shader clock the chip sustains while the handle works: _begin starts a one-wave probe on a stream of its own that compares the
 * shader-clock counter with the constant 100 MHz real-time counter for `microseconds`; issue the work to be measured behind it;
 * _end waits for the probe and returns GHz (bench.py: roofline.clock_ghz -- peaks are quoted at the 2.4 GHz spec clock) */
int fsmg_debug_clock_begin(fsmg_handle h, int32_t microseconds);
int fsmg_debug_clock_end(fsmg_handle h, float* ghz);
/* padded sizes: writes Ep, Hp, V1p, last B, T */
int fsmg_debug_dims(fsmg_handle h, int32_t dims[5]);
/* diagnostics: run ONE instrumented recurrent step kernel (which = 0 forward, 1 backward) at t = T/2 on the
 * buffers of the last forward/backward (clobbers them) and return 8 s_memtime stamps per wave:
 * [0] entry, [1] operands landed, [2] partials in LDS, [3] past the block barrier, [4] done.  cap >= n_blocks*n_waves*8 */
int fsmg_debug_step_profile(fsmg_handle h, int32_t which, uint64_t* stamps, int64_t cap, int32_t* n_blocks,
                            int32_t* n_waves);
/* per-kernel-class HIP-event timing on the handle's stream (disables graph replay while on).
 * classes: "gemm_zx","lstm_fwd","gemm_logits","ce","gemm_dhout","gemm_dw","lstm_bwd",
 * "gemm_dk","gemm_dx","embed_grad","update" */
int fsmg_timing_enable(fsmg_handle h, int32_t on);
/* restrict event timing to ONE kernel class (NULL or "" = all classes): two event records per
 * launch of that class, cheap enough to leave on inside a throughput measurement */
int fsmg_timing_select(fsmg_handle h, const char* kernel_class);
int fsmg_timing_read(fsmg_handle h, const char* kernel_class, double* total_ms, int64_t* launches);
int fsmg_timing_reset(fsmg_handle h);

#ifdef __cplusplus
}
#endif
#endif /* FSMG_H */
