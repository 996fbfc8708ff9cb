/*
 * idiaptts_amd.h -- C ABI of the MI355X-native IdiapTTS hot path (libidiaptts_amd.so).
 *
 * The reference (idiap/IdiapTTS v0.2) has no FFI of its own: the boundary is the Python call
 * surface that today delegates to pyworld / pysptk / bandmat / torch.  Every entry point below
 * cites the reference call site it replaces (paths relative to /root/reference/idiaptts).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures.
 *   - `d_*` parameters are DEVICE pointers (HBM), `h_*` are host pointers.  Buffers are
 *     caller-owned and C-contiguous unless a leading dimension is passed; the library never
 *     keeps a pointer after the call returns.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  GPU entry points are
 *     asynchronous with respect to the host unless stated otherwise.
 *   - return value: 0 = ok, negative = error (ITTS_E_*); itts_last_error() gives a message.
 *   - no hidden global state besides read-only tables cached per device; HIP is not
 *     initialised before the first GPU entry point is called (fork-safe for DataLoader workers,
 *     src/neural_networks/pytorch/ModularModelHandlerPyTorch.py:528-548).
 */
#ifndef IDIAPTTS_AMD_H
#define IDIAPTTS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITTS_OK 0
#define ITTS_E_INVALID (-1)  /* bad argument (shape / size / null pointer)            */
#define ITTS_E_HIP (-2)      /* HIP runtime error, see itts_last_error()               */
#define ITTS_E_UNSUPPORTED (-3)

/* ---- library ------------------------------------------------------------------------- */
int itts_abi_version(void);
const char* itts_last_error(void);
/* number of visible HIP devices (initialises HIP). */
int itts_device_count(void);
/* Bytes of scratch the library holds on the current device (reserved), of which handed out (used),
 * and the amount it may keep between calls (ITTS_POOL_KEEP_GB); any pointer may be NULL. */
int itts_scratch_pool_stats(int64_t* reserved, int64_t* used, int64_t* keep_threshold);

/* The GPU entry points take their scratch from blocks the library keeps per device (stream-ordered:
 * a block handed back on one stream is reused on another only behind an event), up to
 * ITTS_POOL_KEEP_GB (environment, default 64) between calls.  This synchronises the device and hands
 * every idle block back, e.g. between a feature-extraction job and training in one process. */
int itts_release_scratch(void);

/* Deferred reductions (this host thread): while on, itts_linear_fwd_mse / itts_masked_mse /
 * itts_linear_bwd / itts_linear_bwd_weight leave their partial results (loss partial sums, split-K
 * slabs of the weight and bias gradients) in the workspace they were given and queue the reduction
 * they owe; itts_reduce_deferred then runs every queued reduction in ONE launch (bit-identical
 * results) and switches deferral off.  The caller keeps each call's workspace alive and separate
 * until then.  A training step of the dense stack owes five such reductions, each a launch of its own
 * otherwise (FFWrapper.py:63-73 backward + NamedLoss.py:113-117; torch launches one kernel per
 * gradient as well). */
int itts_defer_reductions(int on);
int itts_reduce_deferred(void* stream);

/* ---- gradient exchange of data-parallel training (csrc/collective.cpp) --------------------------
 * Replaces torch.nn.DataParallel's scatter / gather / gradient reduction
 * (src/neural_networks/pytorch/ModularModelHandlerPyTorch.py:732-735, :757-763) by one process per GPU
 * and an in-place all-reduce of a flat device buffer (the gradient arena, or the additive
 * normalisation sums of gen_data) over RCCL / xGMI on the caller's stream.  RCCL is resolved at run
 * time from the copy the process already holds (PyTorch's), else the system's; ITTS_E_UNSUPPORTED when
 * there is none.  A communicator is either the caller's own ncclComm_t or one made here:
 *   itts_comm_unique_id   rank 0 fills 128 bytes, sends them to the others by any means
 *   itts_comm_init_rank   collective over the n_ranks processes (device = the current HIP device)
 *   itts_allreduce_flat   d_buf[0..n) <- reduction over the ranks, in place, asynchronous on `stream`
 *   itts_comm_destroy
 *   itts_comm_version     the library's ncclGetVersion code (e.g. 22707), 0 when no usable RCCL was found: the
 *                         binding passes enum values of the 2.10+ ABI and refuses any other version */
#define ITTS_F32 0
#define ITTS_F64 1
#define ITTS_REDUCE_SUM 0
#define ITTS_REDUCE_MAX 1
#define ITTS_REDUCE_AVG 2
int itts_comm_version(void);
int itts_comm_unique_id(void* h_id128);
int itts_comm_init_rank(const void* h_id128, int n_ranks, int rank, void** comm_out);
int itts_comm_destroy(void* comm);
int itts_allreduce_flat(void* d_buf, int64_t n, int dtype, int op, void* comm, void* stream);

/* ---- integer / scalar helpers (host, no GPU) ------------------------------------------- */
/* pyworld.get_cheaptrick_fft_size(fs, f0_floor=71)  -- src/data_preparation/audio/AudioProcessing.py:60 */
int itts_cheaptrick_fft_size(int fs, double f0_floor);
/* pyworld.get_num_aperiodicities(fs)                -- AudioProcessing.py:71 */
int itts_num_aperiodicities(int fs);
/* pysptk.util.mcepalpha(fs)                         -- AudioProcessing.py:40 */
double itts_mcep_alpha(int fs);
/* number of analysis frames pyworld.wav2world yields: int(1000*n/fs/frame_period)+1 */
int64_t itts_world_num_frames(int64_t n_samples, int fs, double frame_period_ms);
/* samples pyworld.synthesize yields: int(T*frame_period*fs/1000) */
int64_t itts_world_synth_length(int64_t n_frames, int fs, double frame_period_ms);

/* ---- host file I/O of the feature-extraction loop (no GPU; csrc/hostio.cpp) --------------------
 * AudioProcessing.get_raw (src/data_preparation/audio/AudioProcessing.py:107-120) for a batch:
 * mono PCM 8 / 16 / 32-bit or IEEE-float wav files -> float64 samples in [-1, 1] with
 * pre-emphasis raw[i] - p * raw[i-1], files read by n_threads threads.  Two passes: itts_wav_info
 * gives rate and length (ITTS_E_UNSUPPORTED for other layouts: read those another way), the caller
 * sizes one buffer, itts_wav_read_batch fills h_out[h_offsets[i] .. h_offsets[i+1]) with file i. */
int itts_wav_info(const char* h_path, int* fs, int64_t* n_samples);
int itts_wav_read_batch(const char* const* h_paths, int n_files, const int64_t* h_offsets,
                        double preemphasis, double* h_out, int n_threads);
/* The `.npz` archives save_output writes (world/WorldFeatLabelGen.py:1121-1172 through
 * LabelGen._save_to_npz, data_preparation/LabelGen.py:63-101) for a batch of utterances whose
 * features sit in one host matrix h_feat [Ttot, ld] f32 (utterance u = rows h_f_off[u] ..
 * h_f_off[u+1]): archive h_paths[u * n_streams + s] gets stream s = columns h_col0[s] ..., stored
 * as `<key>.npy` (h_parts[s] == 1, width h_width[s]) or as `<key>.npy`, `<key>_deltas.npy`,
 * `<key>_double_deltas.npy` (h_parts[s] == 3, three blocks of h_width[s] columns).  Archives are
 * written to `<path>_tmp` and renamed.  _save_to_npz keeps the other keys of an archive that already
 * exists: with h_needs_merge != NULL ([n_utts * n_streams] bytes) an existing archive holding
 * members this call would not write is left untouched and flagged 1 there (the caller merges it);
 * with NULL existing archives are replaced.  n_threads writer threads. */
int itts_write_feature_archives(const float* h_feat, int64_t ld, const int64_t* h_f_off, int n_utts,
                                const char* const* h_paths, int n_streams, const int* h_col0,
                                const int* h_width, const int* h_parts, const char* const* h_keys,
                                int n_threads, unsigned char* h_needs_merge);

/* Per-item normalisation of the data readers (NpzDataReader.preprocess_sample :347-371): h_out[r][c] =
 * (float)(((double)h_x[r][c] - h_sub[c]) / h_div[c]) -- numpy's `((x - sub) / div).astype(float32)` for float32 x and
 * float64 parameters, bit for bit, in one pass.  Host memory, no GPU. */
int itts_normalise_rows_f32(const float* h_x, int64_t rows, int cols, const double* h_sub,
                            const double* h_div, float* h_out);

/* ---- question labels (host, no GPU; csrc/labels.cpp) ------------------------------------------------
 * HTSLabelNormalisation (src/data_preparation/questions/label_normalisation.py): question-set
 * loading :817-897 (patterns compiled once per question), pattern matching :753-791,
 * load_labels_with_state_alignment :521-666.  Bit-identical to the reference.
 *   itts_questions_load      compiles a `.hed` question file (QS / CQS lines); *handle is freed with
 *                            itts_questions_free; label width = n_binary + n_continuous + 9
 *   itts_questions_vector    question answers of one context string -> h_out [n_binary + n_continuous]
 *   itts_labels_count_frames frames of every state-aligned `.lab` file (int((end - start) / 50000)
 *                            per state line; five state lines [2]..[6] per phone)
 *   itts_labels_generate     frame-level labels of all files into h_out [sum frames, ld_out] f64, file i at
 *                            rows h_frame_off[i] .. h_frame_off[i+1]: question vector of the phone +
 *                            the nine sub-phone features ('full'); files are spread over n_threads */
int itts_questions_load(const char* h_path, void** handle, int* n_binary, int* n_continuous);
void itts_questions_free(void* handle);
int itts_questions_vector(void* handle, const char* h_label, double* h_out);
int itts_labels_count_frames(const char* const* h_paths, int n_files, int64_t* h_frames, int n_threads);
int itts_labels_generate(void* handle, const char* const* h_paths, int n_files,
                         const int64_t* h_frame_off, double* h_out, int64_t ld_out, int n_threads);

/* ---- MLPG (misc/mlpg.py:94-127, bandmat solveh) ----------------------------------------- */
/*
 * Batched maximum-likelihood parameter generation with the reference's three windows
 * ([1], [-.5,0,.5], [1,-2,1]) for U utterances stored back to back.
 *   d_feat      [Ttot, ld_feat] f64; static/delta/delta-delta means of dimension d sit in
 *               columns col0+d, col0+D+d, col0+2D+d          (mlpg.py:119-121)
 *   d_var       [3*D] f64 diagonal of the 3D x 3D covariance (mlpg.py:111-113)
 *   h_offsets   [U+1] frame offsets of the utterances (host), offsets[U] = Ttot
 *   d_out       [Ttot, ld_out] f64; result for dimension d in column ocol0+d
 *   d_scratch   >= itts_mlpg_scratch_bytes(Ttot, D) bytes
 */
int64_t itts_mlpg_scratch_bytes(int64_t t_total, int dim);
int itts_mlpg_generation(const double* d_feat, int64_t ld_feat, int col0, int dim,
                         const double* d_var, const int64_t* h_offsets, int n_utts,
                         double* d_out, int64_t ld_out, int ocol0, void* d_scratch,
                         void* stream);
/* The same with float32 input rows -- the type the acoustic model's output has where mlpg.py:119-121 assigns it
 * into float64 arrays: the conversion happens in the solve's loads (exact), the result is that of
 * itts_mlpg_generation on the widened rows.   d_scratch >= itts_mlpg_scratch_bytes_f32(Ttot, D) bytes. */
int64_t itts_mlpg_scratch_bytes_f32(int64_t t_total, int dim);
int itts_mlpg_generation_f32(const float* d_feat, int64_t ld_feat, int col0, int dim,
                             const double* d_var, const int64_t* h_offsets, int n_utts,
                             double* d_out, int64_t ld_out, int ocol0, void* d_scratch,
                             void* stream);

/* A prepared plan for solving over the SAME utterance layout repeatedly -- the streams of one batch
 * (mlpg.py:94-127 is called once per stream: mcep, lf0, bap), a fixed validation set: the checks of the offsets,
 * the longest length and the one-pass kernel's (start, end) table in launch order are worked out once and kept
 * (the table in page-locked memory the launches read in place), so a planned call does nothing on the host but
 * launch.  The plan must outlive the work queued with it; destroy it after synchronising. */
int itts_mlpg_plan_create(const int64_t* h_offsets, int n_utts, void** plan_out);
void itts_mlpg_plan_destroy(void* plan);
int64_t itts_mlpg_plan_frames(const void* plan);
/* itts_mlpg_generation / itts_mlpg_generation_f32 (feat_is_f32) over the plan's utterances; d_scratch as for those. */
int itts_mlpg_generation_planned(const void* plan, const void* d_feat, int feat_is_f32, int64_t ld_feat, int col0,
                                 int dim, const double* d_var, double* d_out, int64_t ld_out, int ocol0,
                                 void* d_scratch, void* stream);

/* ---- frame utilities (misc/utils.py:40-105) --------------------------------------------- */
/* compute_deltas == np.gradient(x, axis=0) in float32 (utils.py:103-105): part of
 * itts_assemble_cmp_f32 below (static, delta, delta-delta columns of every stream in one pass). */

/* lf0 / V-UV of WorldFeatLabelGen.world_extract_features (world/WorldFeatLabelGen.py:798-802):
 *   lf0 = float32 log(clip(f0, 1e-10)); lf0[lf0 <= log(f0_silence_threshold)] = lf0_zero;
 *   lf0, vuv = interpolate_lin(lf0)
 * for U utterances stored back to back (h_f_off [U+1] frame offsets, each utterance < 2^24
 * frames).  d_f0 [Ttot] f64 -> d_lf0 [Ttot] f32 (interpolated, continuous), d_vuv [Ttot] f32. */
int itts_lf0_vuv(const double* d_f0, const int64_t* h_f_off, int n_utts,
                 double f0_silence_threshold, float lf0_zero, float* d_lf0, float* d_vuv,
                 void* stream);
/* d_x[i] = sqrt(d_x[i]) (IEEE, round to nearest): the amplitude envelope of
 * WorldFeatLabelGen.world_extract_features (world/WorldFeatLabelGen.py:795, np.sqrt(sp)) from CheapTrick's power
 * envelope, before it leaves the device. */
int itts_sqrt_inplace_f64(double* d_x, int64_t n, void* stream);
/* d_x[i] = d_x[i] * d_x[i] (one IEEE product): the power envelope WORLD's synthesis takes from the amplitude envelope
 * of WorldFeatLabelGen.world_features_to_raw (world/WorldFeatLabelGen.py:925, np.square), after the upload. */
int itts_square_inplace_f64(double* d_x, int64_t n, void* stream);

/* interpolate_lin (misc/utils.py:40-86) on float32 contours stored back to back: frames <= 0 are
 * gaps; bit-exact including the reference's quirks (target reached one frame early; a gap whose
 * next voiced frame is the last frame is filled, with that frame, by the last voiced value).
 * d_ip [Ttot] f32 interpolated contour, d_vuv [Ttot] f32 (1 where d_in > 0).  In place
 * (d_ip == d_in) is allowed. */
int itts_interpolate_lin_f32(const float* d_in, const int64_t* h_off, int n_utts, float* d_ip,
                             float* d_vuv, void* stream);
/* Feature matrix of save_output / load (WorldFeatLabelGen.py:1121-1172, :459-573) for U utterances:
 * add_deltas != 0: [sp, d sp, dd sp | lf0, d, dd | vuv | bap, d bap, dd bap], width
 * 3*(n_sp+1+n_bap)+1 (the `.cmp` layout), d = compute_deltas = np.gradient in float32 per
 * utterance (misc/utils.py:103-105), dd = compute_deltas(d); else [sp | lf0 | vuv | bap].
 * d_sp [Ttot, ld_sp] f32, d_lf0 / d_vuv [Ttot] f32, d_bap [Ttot, ld_bap] f32 -> d_out [Ttot, ld_out]. */
int itts_assemble_cmp_f32(const float* d_sp, int64_t ld_sp, int n_sp, const float* d_lf0,
                          const float* d_vuv, const float* d_bap, int64_t ld_bap, int n_bap,
                          const int64_t* h_f_off, int n_utts, int add_deltas, float* d_out,
                          int64_t ld_out, void* stream);
/* Normalisation sums of MeanCovarianceExtractor / MeanStdDevExtractor.add_sample
 * (misc/normalisation/MeanCovarianceExtractor.py:27-31, MeanStdDevExtractor.py) over columns
 * [col0, col0+width) of d_x [n_rows, ld_x] f32, accumulated in fp64 with a fixed summation order:
 * d_sum [width] = sum x; d_second = sum x x^T [width, width] (want_cov != 0; fp64 matrix cores)
 * or sum x^2 [width].  accumulate != 0 adds to the outputs.
 * d_workspace >= itts_feature_stats_workspace_bytes(width, want_cov). */
int64_t itts_feature_stats_workspace_bytes(int width, int want_cov);
int itts_feature_stats(const float* d_x, int64_t ld_x, int64_t n_rows, int col0, int width,
                       int want_cov, int accumulate, double* d_sum, double* d_second,
                       void* d_workspace, void* stream);

/* Row gather of the recurrent layers' packed layout -- pack_padded_sequence / pad_packed_sequence and
 * their gradients (rnn_dyn/RNNWrapper.py:89-102), and the h_{t-1} shift of the weight gradient:
 * d_dst[r, :width] = d_src[d_idx[r], :width] where 0 <= d_idx[r] < n_src, else d_fill_row[:width] (zeros when
 * NULL); columns width .. dst_width - 1 of every row of d_dst are zeroed (16-byte row pitches for the GEMMs). */
int itts_rows_gather_f32(const float* d_src, int64_t ld_src, int64_t n_src, const int64_t* d_idx, int64_t n_out,
                         int width, const float* d_fill_row, float* d_dst, int64_t ld_dst, int dst_width,
                         void* stream);

/* ---- mini-batches on the device (ModularModelHandlerPyTorch.prepare_batch :388-465, sequence_mask :467-491) ----
 * The rows of the utterances of a batch lie back to back in d_src (utterance b: rows d_starts[b] .. d_starts[b] +
 * d_lens[b] - 1, d_starts / d_lens on the device); the padded batch is [n_utts, t_max, width] (batch_first) or
 * [t_max, n_utts, width] with row pitch ld_dst.  Position (b, t) receives row d_starts[b] + t for t < d_lens[b]
 * (and a row inside [0, n_src)), else d_fill_row[:width] (zeros when NULL) -- pad_sequence of the cached rows, or
 * pad_packed_sequence with the layers' value of a padding position as the fill row; the padding position with flat
 * index rep_pos (in the batch's layout; -1: none) receives d_rep_row[:width] instead; d_mask (may be NULL) receives
 * the float sequence mask, one value per position in the batch's layout. */
int itts_batch_pad_gather_f32(const float* d_src, int64_t ld_src, int64_t n_src, const int64_t* d_starts,
                              const int64_t* d_lens, int n_utts, int64_t t_max, int width, int batch_first,
                              const float* d_fill_row, int64_t rep_pos, const float* d_rep_row, float* d_dst,
                              int64_t ld_dst, float* d_mask, void* stream);
/* The adjoint: d_dst[d_starts[b] + t, :width] = padded position (b, t) of d_src for t < d_lens[b]; columns width ..
 * dst_width - 1 of the written rows are zeroed (16-byte row pitches for the GEMMs).  Padding positions are not read,
 * except the one with flat index rep_pos (-1: none), which is written to row rep_dst_row. */
int itts_batch_pack_rows_f32(const float* d_src, int64_t ld_src, const int64_t* d_starts, const int64_t* d_lens,
                             int n_utts, int64_t t_max, int width, int batch_first, float* d_dst, int64_t ld_dst,
                             int dst_width, int64_t rep_pos, int64_t rep_dst_row, void* stream);
/* d_dst[d_dst_starts[b] + t, :width] = d_src[d_src_starts[b] + t, :width] for t < d_lens[b] (t_max >= every length);
 * columns width .. dst_width - 1 of the written rows are zeroed: the valid frames of a mini-batch's utterances out of
 * an arena, back to back in batch order -- the packed batch of the flat feed-forward step
 * (data_preparation/FrameShard.py: gather), one launch instead of a row index built on the host. */
int itts_batch_concat_rows_f32(const float* d_src, int64_t ld_src, int64_t n_src, const int64_t* d_src_starts,
                               const int64_t* d_dst_starts, const int64_t* d_lens, int n_utts, int64_t t_max, int width,
                               float* d_dst, int64_t ld_dst, int dst_width, void* stream);
/* d_out[c] = sum of d_x[(b, t), c] over the padding positions (t >= d_lens[b]) of a padded batch, summed in a fixed
 * order (the gradient that reaches the fill row of itts_batch_pad_gather_f32); d_out[width .. out_width - 1] = 0.
 * d_workspace: itts_batch_pad_colsum_workspace_bytes(n_utts * t_max, width) bytes on the device. */
int64_t itts_batch_pad_colsum_workspace_bytes(int64_t n_rows, int width);
int itts_batch_pad_colsum_f32(const float* d_x, int64_t ld_x, const int64_t* d_lens, int n_utts, int64_t t_max,
                              int width, int batch_first, float* d_out, int out_width, void* d_workspace,
                              void* stream);

/* ---- acoustic model: dense layers (rnn_dyn/FFWrapper.py:63-73 -> torch.nn.Linear + act) --- */
#define ITTS_ACT_NONE 0
#define ITTS_ACT_TANH 1
#define ITTS_ACT_RELU 2
/* Pitch rule for the row-major activations of the four entry points below (x, y, dz, dx, yprev):
 * a row pitch (ld*) that is a multiple of 4 floats on a 16-byte aligned base selects 16-byte loads.
 * If the logical width is not a multiple of 4, the 1-3 floats between the width and the next
 * multiple of 4 are then read as well and must be finite (keep them zero): a padded reduction
 * element meets a zero of the other operand, a padded output column is never stored.
 * y[M,N] = act(x[M,K] @ w[N,K]^T + b[N]); fp32 MFMA. */
int itts_linear_fwd(const float* d_x, int64_t ldx, const float* d_w, const float* d_b,
                    float* d_y, int64_t ldy, int64_t M, int N, int K, int act, void* stream);
/* Output layer of a training step fused with NamedLoss(MSELoss, reduction 'mean_per_frame')
 * (FFWrapper.py:63-73 + loss/NamedLoss.py:70-117): y = x w^T + b is never stored; d_loss [1] receives
 * loss_weight * sum_valid (y - target)^2 / (n_valid * N) and d_dz [M, N] its gradient w.r.t. y.
 * x, w and dz need 16-byte aligned rows (ldx, K, lddz multiples of 4); d_workspace:
 * itts_linear_fwd_mse_workspace_bytes(M, N) bytes. */
int64_t itts_linear_fwd_mse_workspace_bytes(int64_t M, int N);
int itts_linear_fwd_mse(const float* d_x, int64_t ldx, const float* d_w, const float* d_b,
                        const float* d_target, int64_t ldt, const uint8_t* d_row_valid, double n_valid,
                        float loss_weight, int64_t M, int N, int K, float* d_loss, float* d_dz,
                        int64_t lddz, void* d_workspace, void* stream);

/* dz = dy * act'(y)  (elementwise; act' expressed through the layer output y). */
int itts_act_bwd(const float* d_dy, const float* d_y, float* d_dz, int64_t n_elem, int act,
                 void* stream);
/* dx[M,K] = dz[M,N] @ w[N,K]; if d_yprev != NULL the previous layer's act' (through its
 * output yprev[M,K]) is fused into the epilogue: dx *= act'(yprev). */
int itts_linear_bwd_input(const float* d_dz, int64_t lddz, const float* d_w, float* d_dx,
                          int64_t lddx, const float* d_yprev, int64_t ldyp, int act_prev,
                          int64_t M, int N, int K, void* stream);
/* dw[N,K] = dz[M,N]^T @ x[M,K], db[N] = colsum(dz).  Deterministic split-K through
 * d_workspace (>= itts_linear_bwd_weight_workspace_bytes). If accumulate != 0, adds. */
int64_t itts_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K);
int itts_linear_bwd_weight(const float* d_dz, int64_t lddz, const float* d_x, int64_t ldx,
                           float* d_dw, float* d_db, int64_t M, int N, int K,
                           void* d_workspace, int accumulate, void* stream);
/* Both gradients of a layer from dz in one call (autograd's backward of torch.nn.Linear in
 * rnn_dyn/FFWrapper.py:63-73): dw / db as itts_linear_bwd_weight, dx as itts_linear_bwd_input
 * (d_yprev != NULL: times the derivative of the previous layer's activation).  With 16-byte rows
 * the two GEMMs share one launch; the results equal the separate calls bit for bit.  d_workspace
 * as for itts_linear_bwd_weight. */
int itts_linear_bwd(const float* d_dz, int64_t lddz, const float* d_x, int64_t ldx, const float* d_w,
                    float* d_dw, float* d_db, float* d_dx, int64_t lddx, const float* d_yprev,
                    int64_t ldyp, int act_prev, int64_t M, int N, int K, void* d_workspace,
                    int accumulate, void* stream);

/* ---- masked MSE, reduction 'mean_per_frame' (loss/NamedLoss.py:70-117) -------------------- */
/*
 * loss = mean_d( sum_{valid frames} (pred-target)^2 / n_valid ), grad = dloss/dpred.
 * d_row_valid [M] u8 (1 = frame inside the sequence, i.e. the seq mask), n_valid = number of
 * valid frames of the GLOBAL batch (sum of lengths over all ranks when data parallel).
 * d_loss: one f32 (sum over this rank's rows already divided by n_valid*D).
 * d_grad may be NULL (evaluation).  d_workspace >= itts_masked_mse_workspace_bytes(M,D).
 */
int64_t itts_masked_mse_workspace_bytes(int64_t M, int D);
int itts_masked_mse(const float* d_pred, int64_t ldp, const float* d_target, int64_t ldt,
                    const uint8_t* d_row_valid, int64_t M, int D, double n_valid,
                    float loss_weight, float* d_loss, float* d_grad, int64_t ldg,
                    void* d_workspace, void* stream);

/* The other NamedLoss types and reductions (loss/NamedLoss.py:113-131): row-weighted elementwise loss
 *   loss = sum_r w[r] sum_c e(pred[r,c] - target[r,c]),  e = squared (kind 0: MSELoss) or absolute
 *   (kind 1: L1Loss) error;  d_grad = dloss/dpred;  d_elem (optional) = w[r] e per element.
 * The sequence mask and the reduction are folded into d_row_weight [M] f32 by the caller:
 * 'mean_per_frame' mask / (frames * D), 'mean_per_sample' mask / (len_b * B * D), 'mean' mask /
 * (rows * D), 'sum' and 'none' mask.  Rows of weight 0 may hold anything.
 * d_workspace >= itts_masked_mse_workspace_bytes(M, D). */
int itts_weighted_loss(const float* d_pred, int64_t ldp, const float* d_target, int64_t ldt,
                       const float* d_row_weight, int64_t M, int D, int kind, float* d_loss,
                       float* d_grad, int64_t ldg, float* d_elem, int64_t lde, void* d_workspace,
                       void* stream);

/* ---- Adam on a flat fp32 parameter buffer (torch.optim.Adam semantics, ----------------------
 *      ModularModelHandlerPyTorch.py:570-571); grad_scale multiplies the gradient first
 *      (1/world_size after an all-reduce(sum)). */
int itts_adam_step(float* d_param, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   int64_t step, float grad_scale, void* stream);

/* Gradient norm of a flat fp32 buffer for clipping (torch.nn.utils.clip_grad_norm_,
 * ModularModelHandlerPyTorch.py:810-814): norm_kind 2 -> d_accum[0] (+)= sum x^2, norm_kind 0 ->
 * d_accum[0] = max(d_accum[0], max |x|) (infinity norm).  accumulate = 0 overwrites d_accum.
 * Deterministic two-stage reduction; d_workspace >= 4096 bytes. */
int itts_grad_norm_accum(const float* d_x, int64_t n, int norm_kind, float* d_accum,
                         int accumulate, void* d_workspace, void* stream);

/* The whole optimiser tail of one training step in one pass over a flat parameter buffer
 * (ModularModelHandlerPyTorch.py:810-831 + ExponentialMovingAverage.py:32-45):
 *   g = grad * grad_scale
 *   d_norm_accum != NULL: g *= min(1, clip_max_norm / (norm + 1e-6)), norm = sqrt(accum) (kind 2)
 *                         or accum (kind 0) of the scaled gradient    [clip_grad_norm_]
 *   clip_value > 0:       g = clamp(g, -clip_value, clip_value)        [clip_grad_value_]
 *   Adam update as itts_adam_step
 *   d_ema_shadow != NULL: shadow -= (1 - ema_decay) * (shadow - param_new) */
int itts_adam_step_fused(float* d_param, const float* d_grad, float* d_exp_avg,
                         float* d_exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int64_t step, float grad_scale,
                         const float* d_norm_accum, int norm_kind, float clip_max_norm,
                         float clip_value, float* d_ema_shadow, float ema_decay, void* stream);

/* SGD on a flat fp32 buffer (torch.optim.SGD semantics, ModularModelHandlerPyTorch.py:572-573):
 * g += weight_decay*p; with momentum: buf = g on the first step, else momentum*buf +
 * (1-dampening)*g; g = nesterov ? g + momentum*buf : buf; p -= lr*g.  d_momentum_buf may be NULL
 * when momentum == 0. */
int itts_sgd_step(float* d_param, const float* d_grad, float* d_momentum_buf, int64_t n, float lr,
                  float momentum, float dampening, float weight_decay, int nesterov,
                  int first_step, float grad_scale, void* stream);

/* Exponential moving average of the parameters, shadow -= (1-decay)*(shadow - param)
 * (neural_networks/pytorch/ExponentialMovingAverage.py:32-45). */
int itts_ema_update(float* d_shadow, const float* d_param, int64_t n, float decay, void* stream);

/* ---- WORLD analysis, frame-parallel part --------------------------------------------------------
 * Utterances are stored back to back: d_x holds the (pre-emphasised) f64 waveforms, h_x_off[U+1]
 * their sample offsets, h_f_off[U+1] the frame offsets with
 * frames(u) = int(1000*n_u/fs/frame_period)+1 and frame i at time i*frame_period/1000 s
 * (pyworld.wav2world, src/data_preparation/world/WorldFeatLabelGen.py:792-793). */

/* CheapTrick spectral envelope (WORLD cheaptrick.cpp; q1 = -0.15, f0 floor 71 Hz in wav2world)
 * with SPTK mel-cepstral analysis optionally fused behind it, i.e.
 *   sp   = pyworld.cheaptrick(x, f0, t, fs, fft_size=fft_size)            -> d_sp [Ttot, K] f64 power
 *   mcep = pysptk.mcep(sqrt(sp), order, alpha, eps, min_det=0, etype=1, itype=3)
 *          (AudioProcessing.extract_mcep, src/data_preparation/audio/AudioProcessing.py:142-153)
 * d_sp may be NULL when only mcep is wanted (the envelope then never leaves the chip).
 * d_mc_f32 [Ttot, ld_mc] and/or d_mc_f64 [Ttot, order+1]; d_iters [Ttot] Newton steps (optional). */
int itts_cheaptrick_mcep(const double* d_x, const int64_t* h_x_off, const double* d_f0,
                         const int64_t* h_f_off, int n_utts, int fs, double frame_period_ms,
                         int fft_size, double q1, double* d_sp, int order, double alpha, double eps,
                         int miniter, int maxiter, double threshold, float* d_mc_f32, int64_t ld_mc,
                         double* d_mc_f64, int* d_iters, void* stream);

/* pysptk.mcep on a given amplitude spectrum d_amp_sp [T, K] f64 (AudioProcessing.py:146-152). */
int itts_mcep(const double* d_amp_sp, int64_t T, int K, int order, double alpha, double eps,
              int miniter, int maxiter, double threshold, float* d_mc_f32, int64_t ld_mc,
              double* d_mc_f64, int* d_iters, void* stream);

/* pysptk.mgc2sp(mc, alpha, gamma=0, fftlen): d_logamp_f64 [T, fftlen/2+1] = real part (log
 * amplitude); d_amp_f32 = exp(float32(real)) as AudioProcessing.mcep_to_amp_sp (:252-256);
 * d_pow_f64 = float64(d_amp_f32)^2, the power spectrum world_features_to_raw hands to WORLD
 * (WorldFeatLabelGen.py:925). Any of the three outputs may be NULL. */
int itts_mgc2sp(const double* d_mc, int64_t T, int order, double alpha, int fftlen,
                float* d_amp_f32, double* d_logamp_f64, double* d_pow_f64, void* stream);

/* pysptk.mgcep(amp_sp, order, alpha, gamma, eps, min_det=0, etype=1, itype=3) with pysptk's defaults
 * num_recursions = K - 1, otype = 0 (AudioProcessing.extract_mgc, AudioProcessing.py:123-140; the
 * reference uses gamma = -1/3): d_amp_sp [T, K] f64 amplitude spectra (input_is_power != 0: power
 * spectra, e.g. CheapTrick's output) -> mel-generalized cepstra d_mgc_f32 [T, ld_mgc] and / or
 * d_mgc_f64 [T, order+1]; d_iters [T] Newton steps after the initial gamma = -1 step (optional).
 * -1 <= gamma <= 0, order <= 63. */
int itts_mgcep(const double* d_amp_sp, int input_is_power, int64_t T, int K, int order, double alpha,
               double gamma, double eps, int miniter, int maxiter, double threshold,
               float* d_mgc_f32, int64_t ld_mgc, double* d_mgc_f64, int* d_iters, void* stream);
/* pysptk.mgc2sp(mgc, alpha, gamma, fftlen) (AudioProcessing.mgc_to_amp_sp, AudioProcessing.py:259-275):
 * outputs as itts_mgc2sp; gamma = 0 is itts_mgc2sp. */
int itts_mgc2sp_gamma(const double* d_mgc, int64_t T, int order, double alpha, double gamma,
                      int fftlen, float* d_amp_f32, double* d_logamp_f64, double* d_pow_f64,
                      void* stream);

/* pyworld.code_aperiodicity (WorldFeatLabelGen.py:805) / pyworld.decode_aperiodicity (:940-941). */
int itts_code_aperiodicity(const double* d_ap, int64_t T, int fft_size, int fs, double* d_bap_f64,
                           float* d_bap_f32, void* stream);
int itts_decode_aperiodicity(const double* d_bap, int64_t T, int fs, int fft_size, double* d_ap,
                             void* stream);
/* The same for the synthesis that follows (WorldFeatLabelGen.py:940-945): only the rows a voiced pulse can read --
 * frames with f0 > 0 and their two neighbours, d_f0 [T] f64 -- are written; the rest of d_ap is left untouched. */
int itts_decode_aperiodicity_voiced(const double* d_bap, const double* d_f0, int64_t T, int fs,
                                    int fft_size, double* d_ap, void* stream);

/* StoneMask F0 refinement (pyworld.stonemask; second stage of wav2world; also
 * src/data_preparation/world/LF0LabelGen.py:263-264). d_f0_in / d_f0_out [Ttot] f64. */
int itts_stonemask(const double* d_x, const int64_t* h_x_off, const double* d_f0_in,
                   const int64_t* h_f_off, int n_utts, int fs, double frame_period_ms,
                   double* d_f0_out, void* stream);

/* D4C band aperiodicity with LoveTrain V/UV (pyworld.d4c, threshold 0.85 in wav2world).
 * d_ap [Ttot, fft_size/2+1] f64 (may be NULL) and / or the coded band aperiodicity
 * pyworld.code_aperiodicity(ap, fs) as f64 [Ttot, nap] / f32 [Ttot, ld_bap] (may be NULL). */
int itts_d4c(const double* d_x, const int64_t* h_x_off, const double* d_f0, const int64_t* h_f_off,
             int n_utts, int fs, double frame_period_ms, int fft_size, double threshold,
             double* d_ap, double* d_bap_f64, float* d_bap_f32, int64_t ld_bap, void* stream);

/* DIO F0 estimation (pyworld.dio defaults: f0_floor 71, f0_ceil 800, channels_in_octave 2,
 * speed 1, allowed_range 0.1; first stage of wav2world, WorldFeatLabelGen.py:792-793).
 * d_f0 [Ttot] f64. Scratch is taken from the stream-ordered allocator (hipMallocAsync). */
int itts_dio(const double* d_x, const int64_t* h_x_off, const int64_t* h_f_off, int n_utts, int fs,
             double frame_period_ms, double f0_floor, double f0_ceil, double channels_in_octave,
             double allowed_range, double* d_f0, void* stream);

/* Harvest F0 estimation (pyworld.harvest(x, fs, f0_floor=71, f0_ceil=800, frame_period=5): the
 * estimator BASELINE.json's north_star names beside DIO; the reference's own extractor is
 * dio + stonemask, WorldFeatLabelGen.py:792-793).  Frame count per utterance:
 * itts_harvest_num_frames (WORLD GetSamplesForHarvest).  d_f0 [Ttot] f64.  The d_dbg_* pointers
 * (NULL, or device buffers when n_utts == 1) receive intermediate stages on the 1 ms grid
 * T1 = itts_harvest_num_frames(n, fs, 1.0): raw band candidates [channels, T1]; refined candidates
 * and scores after RemoveUnreliableCandidates [T1, max_candidates]; the contour before smoothing
 * [T1].  Synchronises the stream once at the end (event-list overflow check). */
int64_t itts_harvest_num_frames(int64_t n_samples, int fs, double frame_period_ms);
int itts_harvest(const double* d_x, const int64_t* h_x_off, const int64_t* h_f_off, int n_utts,
                 int fs, double frame_period_ms, double f0_floor, double f0_ceil, double* d_f0,
                 double* d_dbg_raw, double* d_dbg_cand, double* d_dbg_score, double* d_dbg_best,
                 void* stream);

/* pyworld.wav2world(x, fs, fft_size=..., frame_period=...) in one call (WorldFeatLabelGen.py:792-793):
 * DIO -> StoneMask -> CheapTrick -> D4C with wav2world's defaults (f0 floor 71 Hz, q1 -0.15, D4C
 * threshold 0.85).  d_f0 [Ttot] f64, d_sp / d_ap [Ttot, fft_size/2+1] f64 (either may be NULL);
 * fft_size <= 0: pyworld.get_cheaptrick_fft_size(fs). */
int itts_wav2world(const double* d_x, const int64_t* h_x_off, const int64_t* h_f_off, int n_utts,
                   int fs, double frame_period_ms, int fft_size, double* d_f0, double* d_sp,
                   double* d_ap, void* stream);

/* WORLD synthesis (pyworld.synthesize(f0, sp, ap, fs, frame_period)) followed by the reference's
 * float32 cast and de-pre-emphasis lfilter([1],[1,-preemphasis]) -- WorldFeatLabelGen.py:943-945.
 * d_f0 [Ttot] f64, d_sp / d_ap [Ttot, fft_size/2+1] f64 (power spectrum / aperiodicity);
 * h_y_off[U+1] sample offsets with y_len(u) = int(T_u*frame_period*fs/1000).
 * d_y_f32 and / or d_y_f64 [Ytot] (f64 = what scipy.signal.lfilter returns). */
int itts_world_synthesize(const double* d_f0, const double* d_sp, const double* d_ap,
                          const int64_t* h_f_off, const int64_t* h_y_off, int n_utts, int fs,
                          double frame_period_ms, int fft_size, double preemphasis, float* d_y_f32,
                          double* d_y_f64, void* stream);
/* The same, with the spectra still being produced when the call is made: everything in front of the pulse kernels (the
 * per-sample phase, the pulse positions, the noise: a third of the launches, bound by latency) needs d_f0 only.  `stream`
 * waits for `envelope_ready_event` (a hipEvent_t recorded behind the producer of d_sp on whatever stream it ran) right
 * before the kernel of the unvoiced pulses, which read the envelope only, and for `aperiodicity_ready_event` (behind
 * the producer of d_ap) before the kernel of the voiced ones.  NULL: that array is complete on `stream`. */
int itts_world_synthesize_after(const double* d_f0, const double* d_sp, const double* d_ap,
                                const int64_t* h_f_off, const int64_t* h_y_off, int n_utts, int fs,
                                double frame_period_ms, int fft_size, double preemphasis,
                                float* d_y_f32, double* d_y_f64, void* stream, void* envelope_ready_event,
                                void* aperiodicity_ready_event);

/* ---- acoustic model: (Bi)LSTM recurrence (torch.nn.LSTM inside rnn_dyn/RNNWrapper.py:45-107) ------
 * Packed rows, the layout pack_padded_sequence produces (RNNWrapper.py:89-92): the B sequences are
 * sorted by decreasing length, frame t of sequence b is packed row d_row_off[t] + b with
 * d_row_off[t] = sum_{t' < t} #{b : len_b > t'}; N = sum of the lengths rows in total, T = the
 * longest length.  Direction d owns column block d of every tensor.
 *   d_gin    [N, ndir*4H]    x W_ih^T + b_ih + b_hh for all frames (one itts_linear_fwd call)
 *   d_whh    [ndir][4H][H]   gate order i, f, g, o
 *   d_h0/c0  [ndir][H] or NULL (zeros)
 *   d_lengths [B] int32 on the device, h_lengths the same values on the host (sorted decreasingly;
 *            the host copy sizes the per-step launches), d_row_off [T] int32 on the device;
 *   d_rev_row [T*B] int32 on the device (needed when ndir == 2): the reverse direction starts at
 *            each sequence's own last frame, at step s it visits packed row
 *            d_rev_row[s*B + b] = d_row_off[len_b - 1 - s] + b (entries with s >= len_b unused)
 *   d_y      [N, ndir*H]
 *   d_gates [N, ndir, H, 4] (i, f, g, o after activation, gate-minor) and d_csave [N, ndir*H] (c_t):
 *            saved for the backward pass, both NULL for inference.  h_{t-1}, which dW_hh needs next
 *            to d_dg, is d_y shifted by one frame along each sequence (h0 at the first frame).
 *   d_hn/d_cn [ndir][B][H]   final states in sorted row order (may be NULL);
 *   d_state >= itts_lstm_state_bytes bytes */
int64_t itts_lstm_state_bytes(int B, int H, int ndir);
int itts_lstm_layer_fwd(const float* d_gin, const float* d_whh, const float* d_h0, const float* d_c0,
                        const int* d_lengths, const int* h_lengths, const int* d_row_off,
                        const int* d_rev_row, int T, int B, int H, int ndir, float* d_y,
                        float* d_gates, float* d_csave, float* d_hn, float* d_cn, void* d_state,
                        void* stream);
/* d_dg [N, ndir*4H] = dLoss/d(pre-activation gates) from d_dy [N, ndir*H]; d_whh as in the
 * forward call.  dW_ih, dW_hh, db and dX follow from d_dg with itts_linear_bwd_weight /
 * itts_linear_bwd_input.  d_dc0 [ndir][B][H] (may be NULL) receives dLoss/d(initial cell state) per
 * packed row (RNNWrapper's train_hidden_init, rnn_dyn/RNNWrapper.py:66-71: sum over the rows);
 * dLoss/d(initial hidden state) of row b is W_hh^T d_dg[first processed frame of b]. */
int itts_lstm_layer_bwd(const float* d_dy, const float* d_whh, const float* d_c0,
                        const float* d_gates, const float* d_csave, const int* h_lengths,
                        const int* d_row_off, const int* d_rev_row, int T, int B, int H, int ndir,
                        float* d_dg, float* d_dc0, void* d_state, void* stream);

/* ---- (Bi)GRU recurrence (torch.nn.GRU behind rnn_dyn/RNNWrapper.py:45-107 for `..GRU..` groups;
 *      gate order r, z, n; packed rows, lengths and row offsets as for the LSTM entry points).
 * d_gin  [N, ndir*3H] = X W_ih^T + b_ih for all frames (one itts_linear_fwd call),
 * d_whh  [ndir][3H][H], d_bhh [ndir][3H] (b_hn sits inside r * (W_hn h + b_hn)), d_h0 [ndir][H] or
 * NULL.  Training saves d_gates [N, ndir, H, 4] = (r, z, n after activation, W_hn h + b_hn) per
 * unit; NULL for inference.  d_hn [ndir][B][H] may be NULL.
 * d_state >= itts_gru_state_bytes(B, H, ndir).
 * Backward takes d_hprev [N, ndir*H] (h_{t-1} of every frame: d_y shifted by one frame along each
 * sequence, h0 at the first frame) and fills d_dgi (gradient wrt d_gin: feeds dX, dW_ih, db_ih) and d_dgh (gradient wrt the
 * hidden projections: feeds dW_hh with d_hprev, db_hh), both [N, ndir*3H].  d_dh0 [ndir][B][H] (may
 * be NULL) receives the direct part dh * z of dLoss/d(initial hidden state) per packed row; the
 * recurrent part of row b is W_hh^T d_dgh[first processed frame of b] (train_hidden_init). */
int64_t itts_gru_state_bytes(int B, int H, int ndir);
int itts_gru_layer_fwd(const float* d_gin, const float* d_whh, const float* d_bhh,
                       const float* d_h0, const int* d_lengths, const int* h_lengths,
                       const int* d_row_off, const int* d_rev_row, int T, int B, int H, int ndir,
                       float* d_y, float* d_gates, float* d_hn, void* d_state, void* stream);
int itts_gru_layer_bwd(const float* d_dy, const float* d_whh, const float* d_gates,
                       const float* d_hprev, const int* h_lengths,
                       const int* d_row_off, const int* d_rev_row, int T, int B, int H, int ndir,
                       float* d_dgi, float* d_dgh, float* d_dh0, void* d_state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IDIAPTTS_AMD_H */
