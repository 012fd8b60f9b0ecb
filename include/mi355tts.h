/*
 * mi355tts.h — C-ABI of libmi355tts.so, the MI355X (gfx950) engine that stands in for the
 * reference's onnxruntime.InferenceSession.run() hot path.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference repo DakeQQ/Text-to-Speech-TTS-ONNX):
 *
 *   mi_bigvgan_forward      ort_session_A.run_with_ort_values([generated_wav], {mel_features})
 *                           BigVGAN/Export_BigVGAN.py:170   (graph = BIGVGAN.forward, :44-49)
 *   mi_f5_preprocess        ort_session_A.run(...)   F5_TTS/F5-TTS-ONNX-Inference.py:247-253
 *                           (graph = F5Preprocess.forward, F5_TTS/Export_F5.py:117-141)
 *   mi_f5_transformer_step  ort_session_B.run(...)   F5_TTS/F5-TTS-ONNX-Inference.py:292-303
 *                           (graph = F5Transformer.forward, F5_TTS/Export_F5.py:167-182)
 *   mi_f5_sample            the whole `for i in range(0, NFE_STEP - 1, FUSE_NFE)` loop, :291-304,
 *                           kept on device (no per-step host round trip)
 *   mi_f5_decode            ort_session_C.run(...)   F5_TTS/F5-TTS-ONNX-Inference.py:306-311
 *                           (graph = F5Decode.forward, F5_TTS/Export_F5.py:193-203)
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; tensors are row-major contiguous with exactly
 *     the ONNX graph layouts.
 *   - `mem` says where caller buffers live: MI_HOST (copied by the engine) or MI_DEVICE (HIP
 *     device pointers on the handle's device, e.g. torch tensor.data_ptr()).
 *   - every function returns 0 on success, a negative MI_E* code on failure; mi_last_error()
 *     gives the message (thread-local).  Constructors return NULL on failure.
 *   - a handle owns its weights, workspace and one HIP stream; calls on one handle are
 *     serialised; different handles may be used concurrently from different threads.
 *   - weights are one flat fp32 blob in the canonical tensor order of
 *     mi355tts/weights.py (bigvgan_spec / f5_packed_spec), PyTorch-native layouts, with the
 *     reference's export-time folds already applied by the packer.
 */
#ifndef MI355TTS_H
#define MI355TTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MI_F32 = 0, MI_F16 = 1, MI_BF16 = 2 };          /* activation / matmul-operand dtype   */
enum { MI_HOST = 0, MI_DEVICE = 1 };                    /* where caller buffers live           */
enum { MI_OK = 0, MI_EINVAL = -1, MI_EHIP = -2, MI_ENOMEM = -3, MI_ESTATE = -4 };

typedef struct mi_bigvgan mi_bigvgan;
typedef struct mi_f5 mi_f5;

/* ---- process-level ------------------------------------------------------------------------ */
int         mi_init(int device);            /* hipSetDevice + context warm-up                  */
int         mi_device_count(void);
const char* mi_last_error(void);
const char* mi_version(void);

/* ---- BigVGAN-v2 vocoder ---------------------------------------------------------------------
 * cfg: int32 array = BigVGANConfig.to_int_array():
 *   [num_mels, initial_channel, n_up, n_kernels, bias_at_final, tanh_at_final, snake_logscale,
 *    rates[n_up], up_kernels[n_up], res_kernels[n_kernels], n_dil, dil[n_kernels][n_dil],
 *    pre_layernorm, speaker_cond]   (the last two are optional; 1,1 = IndexTTS graph F)          */
int64_t     mi_bigvgan_param_count(const int32_t* cfg, int n_cfg);
mi_bigvgan* mi_bigvgan_create(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights,
                              int dtype, int device);
/* same, with the blob in host (MI_HOST) or device (MI_DEVICE) memory, e.g. the buffer an RCCL broadcast filled (the
 * broadcast itself is torch.distributed's: SURVEY.md 8b `mi_bcast_weights` became "hand the engine the device buffer").
 * BigVGAN / GPT re-lay their conv weights out in host code, so a device blob is read back once at load; F5 converts its
 * matrices device-to-device (see mi_f5_create_mem).                                                                  */
mi_bigvgan* mi_bigvgan_create_mem(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights,
                                  int dtype, int device, int mem);
void        mi_bigvgan_destroy(mi_bigvgan* h);
int64_t     mi_bigvgan_out_len(const mi_bigvgan* h, int frames);      /* frames*hop + 30 */
/* mel: (B, num_mels, frames) fp32 channels-first (the ONNX `mel_features` layout; B>1 is this
 * engine's extension).  out: (B, 1, out_len) int16 = trunc(clamp(32767*tanh(.)))               */
int         mi_bigvgan_forward(mi_bigvgan* h, const float* mel, int B, int frames, int16_t* out, int mem);
/* same, but the float waveform before the int16 conversion (tests)                            */
int         mi_bigvgan_forward_f32(mi_bigvgan* h, const float* mel, int B, int frames, float* out, int mem);
/* IndexTTS graph F (speaker-conditioned vocoder): ort_session_F.run_with_ort_values, IndexTTS/Inference_IndexTTS_ONNX.py:787
 * (graph = IndexTTS_F.forward, IndexTTS/Export_IndexTTS.py:300-314).  The handle must be created from a config with the
 * pre_layernorm / speaker_cond flags (BigVGANConfig.indextts()).  latent: `save_hidden_state` (T_codes, gpt_dim) fp32
 * (the last two rows are dropped, like the reference); conds: save_bigvgan_conds_0..n-1 followed by
 * bigvgan_cond_layer_speaker_embedding, concatenated (n_conds floats).  out: int16 (1,1,(T_codes-2)*hop+30).            */
int         mi_bigvgan_forward_latent(mi_bigvgan* h, const float* latent, int T_codes, const float* conds, int64_t n_conds,
                                      int16_t* out, float* out_f32, int mem);
/* unit-level entry (tests): one anti-aliased SnakeBeta Activation1d on x (B,C,T) fp32
 * channels-first host memory; post!=0 selects the pad-15 variant (out T+30).                   */
int         mi_aa_activation1d(const float* x, int B, int C, int T, const float* alpha_log,
                               const float* beta_log, int logscale, int post, int dtype, float* y);
/* unit-level entry (tests): the FUSED kernel of the low-channel stages — Activation1d(SnakeBeta) -> Conv1d(C, C, k, dilation,
 * "same" padding) (+ res): `xt = c(a(x))` / `x = c2(a2(xt)) + x` of AMPBlock1.forward, BigVGAN/modeling_modified/
 * bigvgan.py:132-140.  x, res, y: (B, C, T) fp32 channels-first host memory; w (C, C, k); C % 8 == 0, C <= 96.             */
int         mi_aa_conv1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log, int logscale,
                         const float* w, const float* bias, int k, int dilation, const float* res, int dtype, float* y);
/* unit-level entry (tests): Conv1d / ConvTranspose1d on fp32 channels-first host tensors,
 * executed by the implicit-GEMM MFMA kernel in `dtype`.                                       */
int         mi_conv1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout,
                      int k, int dilation, int padding, int groups, int dtype, float* y);
int         mi_conv_transpose1d(const float* x, int B, int Cin, int T, const float* w, const float* bias,
                                int Cout, int k, int stride, int padding, int dtype, float* y);

/* ---- F5-TTS (DiT CFM sampler + text/mel front end + Vocos/ISTFT back end) ---------------------
 * cfg_i = F5Config.to_int_array() (21 ints: dim, depth, heads, dim_head, ff_mult, mel_dim, text_dim,
 *   text_num_embeds, conv_layers, conv_mult, pos_conv_kernel, pos_conv_groups, freq_embed_dim, nfe_step,
 *   max_signal_length, n_fft, hop_length, sample_rate, vocos_dim, vocos_intermediate, vocos_layers);
 * cfg_f = [cfg_strength, sway_coef] or [cfg_strength, sway_coef, attn_score_scale].  The third value (MI_F16 engines
 * only; absent = 1) selects the attention rounding points of the reference's fp16-transformer export
 * (use_fp16_transformer, F5_TTS/Export_F5.py:20,321-326; F5/fp16/modules.py:467): the caller folds head_dim^-0.25 * 0.1
 * into the q / k projections of the blob (mi355tts.weights.fold_f5 does, from F5Config.ref_fp16_attn), the kernel rounds
 * the q k scores to fp16 and multiplies them by attn_score_scale (= 100) in fp32 before the fp32 softmax.  `dtype` selects the DiT operand type (fp32 residual stream and
 * fp32 softmax/norm statistics always); the front end and the vocoder always run in fp32, like the
 * reference's CPU-pinned graphs A and C (F5-TTS-ONNX-Inference.py:173,214).
 * Batching extension: U utterances of equal max_duration N; tensors gain a leading U axis.        */
int64_t     mi_f5_param_count(const int32_t* cfg_i, int n_i, const float* cfg_f, int n_f);
mi_f5*      mi_f5_create(const int32_t* cfg_i, int n_i, const float* cfg_f, int n_f, const float* weights,
                         int64_t n_weights, int dtype, int device);
/* blob in host or device memory.  MI_DEVICE: every DiT / Vocos matrix (98 % of the blob) is converted to the engine
 * dtype by a device kernel straight out of the caller's buffer — no host staging; only the tensors whose load-time tables
 * are built by host code (time MLP, the two k31 grouped convs, the Vocos embed conv: ~6 M of 351 M floats) are read back. */
mi_f5*      mi_f5_create_mem(const int32_t* cfg_i, int n_i, const float* cfg_f, int n_f, const float* weights,
                             int64_t n_weights, int dtype, int device, int mem);
void        mi_f5_destroy(mi_f5* h);
/* load-time tables (tests): time_expand (nfe, dim), delta_t (nfe-1)   Export_F5.py:153-164          */
int         mi_f5_tables(mi_f5* h, float* time_expand, float* delta_t);
/* What the engine is doing, machine-readably (no reference counterpart: ONNX Runtime has one arithmetic).  Keys:
 *   "f32_arithmetic"     fp32 engines: 0 native fp32 MFMA | 2 fp16 {hi, lo} pairs | 3 three bf16 planes — the form IN USE (the
 *                        config's trailing int 21 selects it; an fp16-pair engine whose weights exceed the fp16 range at load,
 *                        or whose activations do during a call, has switched itself to 3); -1 for 16-bit engines
 *   "saturation_events"  calls on this handle during which a pair operand met the fp16 range limit and that were therefore
 *                        re-run on three bf16 planes (0 or 1: the switch is permanent)
 *   "adaln_fold"         1 when the load-time vectors of the AdaLN fold exist (LayerNorm statistics carried by the GEMM epilogues)
 * Returns the value, or a negative MI_E* code for a null handle / unknown key.                                              */
int64_t     mi_f5_info(mi_f5* h, const char* key);
/* graph A.  audio (L) int16, text_ids (T) int32 (pad value -1 allowed), max_duration N.
 * noise_in != NULL injects the initial noise (N,100); else it is drawn from `seed`.
 * Outputs (any may be NULL): noise (N,100), rope_cos/rope_sin (N,64) [the ONNX graph broadcasts these
 * to (2,16,N,64) and a transposed K copy], cat_mel_text / cat_mel_text_drop (N,612), ref_signal_len.   */
int         mi_f5_preprocess(mi_f5* h, const int16_t* audio, int64_t L, const int32_t* text_ids, int64_t T,
                             int64_t max_duration, const float* noise_in, uint64_t seed, float* noise,
                             float* rope_cos, float* rope_sin, float* cat_mel_text, float* cat_mel_text_drop,
                             int64_t* ref_signal_len, int mem);
/* unit-level entry (tests): STFT-B of graph A alone (F5_TTS/STFT_Process.py:144-157, reflect padding): audio (L) int16 ->
 * spec (L/hop + 1, 2*(n_fft/2+1)) fp32, row f = [re(0..n_fft/2) | im(0..n_fft/2)] of frame f.                          */
int         mi_f5_stft(mi_f5* h, const int16_t* audio, int64_t L, float* spec, int mem);
/* graph B, one call = `fuse` Euler/CFG steps starting at *time_step (host int); noise (U,N,100) is
 * updated in place, *time_step += fuse.  cat_mel_text(_drop): (U,N,612).                              */
int         mi_f5_transformer_step(mi_f5* h, float* noise, const float* cat_mel_text, const float* cat_mel_text_drop,
                                   int U, int64_t N, int32_t* time_step, int fuse, int mem);
/* the whole NFE loop on device: steps k0 .. k0+n_steps-1 (reference: k0 = 0, n_steps = nfe_step-1)   */
int         mi_f5_sample(mi_f5* h, float* noise, const float* cat_mel_text, const float* cat_mel_text_drop, int U,
                         int64_t N, int k0, int n_steps, int mem);
/* tests: one DiT evaluation at grid index k -> pred (2U, N, 100) (cond branch first)                 */
int         mi_f5_dit_eval(mi_f5* h, const float* noise, const float* cat_mel_text, const float* cat_mel_text_drop,
                           int U, int64_t N, int k, float* pred, int mem);
/* graph C.  denoised (U,N,100), ref_signal_len R -> int16 (U, (N-R-1)*hop); out_f32 (optional) is the
 * float signal before clamp/int16.  *out_len receives the per-utterance sample count.                 */
int         mi_f5_decode(mi_f5* h, const float* denoised, int U, int64_t N, int64_t ref_signal_len, int16_t* out,
                         float* out_f32, int64_t* out_len, int mem);
/* A -> loop -> C without leaving the device.  audio (U,L), text_ids (U,T), noise_in (U,N,100) or NULL. */
int         mi_f5_synthesize(mi_f5* h, int U, const int16_t* audio, int64_t L, const int32_t* text_ids, int64_t T,
                             int64_t max_duration, const float* noise_in, uint64_t seed, int16_t* out,
                             int64_t* out_len, int mem);

/* A -> loop, with the generated frames handed on as a vocoder mel instead of going through Vocos: mel_out (U, 100, N - R)
 * fp32, channels first = the `mel_features` layout of mi_bigvgan_forward (BigVGAN-v2 24khz_100band_256x has F5's 100 bands,
 * hop 256, 24 kHz).  The "F5-TTS + BigVGAN" pipeline of BASELINE.json's metric; the reference's bigvgan-type mel front end is
 * F5_TTS/modeling_modified/F5/modules.py:30-72 (its exported graphs use the Vocos pair).  *n_frames receives N - R.     */
int         mi_f5_synthesize_mel(mi_f5* h, int U, const int16_t* audio, int64_t L, const int32_t* text_ids, int64_t T,
                                 int64_t max_duration, const float* noise_in, uint64_t seed, float* mel_out,
                                 int64_t* n_frames, int mem);

/* ---- IndexTTS acoustic GPT-2 (graphs B, C, E + the greedy decode loop) ------------------------------------------
 * Replaces ort_session_B / _C / _E of IndexTTS/Inference_IndexTTS_ONNX.py:619-675 and the loop at :745-783
 * (graph definitions: IndexTTS/Export_IndexTTS.py:203-289).  The KV cache lives in the handle (the reference
 * passes 2*layers growing tensors in and out of every call); mi_gpt_kv_read / _write expose it in the reference's
 * tensor layouts.  cfg = {hidden, layers, heads, inner, mel_codes, text_tokens, max_mel_pos, max_text_pos, max_seq
 * [, max_batch = 1]}.
 * weights: canonical fp32 blob of mi355tts.weights.pack_gpt (Conv1D weights transposed to (out, in); q and k rows
 * pre-scaled by head_dim^-0.25 like Export_IndexTTS.py:257-258).                                                  */
typedef struct mi_gpt mi_gpt;
typedef struct mi_cond mi_cond;
int64_t     mi_gpt_param_count(const int32_t* cfg, int n_cfg);
mi_gpt*     mi_gpt_create(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int dtype, int device);
mi_gpt*     mi_gpt_create_mem(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int dtype, int device,
                              int mem);
void        mi_gpt_destroy(mi_gpt* h);
/* graph B: text_ids (n) -> text_hidden_state (n + 2, hidden): start id 0 / end id 1 added, + position rows. */
int         mi_gpt_text_embed(mi_gpt* h, const int32_t* text_ids, int n, float* out, int mem);
/* graph C: mel embedding of gpt_id + mel position row gen_len -> (hidden). (the caller keeps gen_len + 1) */
int         mi_gpt_mel_embed(mi_gpt* h, int32_t gpt_id, int64_t gen_len, float* out, int mem);
/* history_len := 0 (the reference re-feeds the empty init_past_keys_E / init_past_values_E, :796-800) */
int         mi_gpt_reset(mi_gpt* h);
int64_t     mi_gpt_history_len(mi_gpt* h);
/* graph E: hidden_state (ids_len, hidden) appended after the handle's history; repeat_penality (mel_codes) or NULL
 * (= ones); attention_mask 1 for the prompt pass, 0 for single-token steps.  Outputs: last_hidden_state (hidden),
 * max_logit_id (1), optionally the logits before the penalty multiply (mel_codes).  history_len += ids_len.     */
int         mi_gpt_step(mi_gpt* h, const float* hidden_state, int ids_len, const float* repeat_penality,
                        int attention_mask, float* last_hidden_state, int32_t* max_logit_id, float* logits, int mem);
/* in_key_i / in_value_i / out_key_i / out_value_i of graph E: keys (heads, 64, history), values (heads, history, 64).
 * mi_gpt_kv_write sets history_len := hist (write every layer with the same hist). */
int         mi_gpt_kv_read(mi_gpt* h, int layer, float* keys, float* values, int mem);
int         mi_gpt_kv_write(mi_gpt* h, int layer, const float* keys, const float* values, int hist, int mem);
/* The per-sentence loop (:745-783) on the device: prompt (P, hidden) = graph D's concat; at most max_new tokens
 * (the reference's generate_limit = MAX_GENERATE_LENGTH - P); stops after a token in stop_ids; repeat_value /
 * penalty_range = REPEAT_PENALITY / PENALITY_RANGE.  repeat_penality (mel_codes) is read and written back (the
 * reference carries it across sentences, :685) — NULL starts from ones.  Outputs: tokens (max_new), hidden
 * (max_new, hidden) = the stacked last_hidden_state rows graph F consumes, *n_out (host) = tokens produced.     */
int         mi_gpt_generate(mi_gpt* h, const float* prompt, int P, int max_new, const int32_t* stop_ids, int n_stop,
                            float repeat_value, int penalty_range, float* repeat_penality, int32_t* tokens,
                            float* hidden, int32_t* n_out, int mem);

/* The same loop for nb sentences at once (nb <= the handle's max_batch = optional 10th cfg int): prompt passes run one
 * after the other, then every decode step serves all unfinished sentences with ONE pass over the weights (batched
 * GEMV), each sentence with its own cache, history, penalty vector, stop test and limit.  prompts = the nb prompts
 * concatenated row-wise (sum(prompt_rows), hidden); prompt_rows / max_new / n_out are host arrays of nb ints;
 * repeat_penality (nb, mel_codes) in/out or NULL; tokens (nb, cap), hidden (nb, cap, hidden).  Results per sentence
 * are identical to mi_gpt_generate on that sentence alone (fp32: bit-for-bit the same kernels' arithmetic order). */
int         mi_gpt_generate_batch(mi_gpt* h, int nb, const float* prompts, const int32_t* prompt_rows,
                                  const int32_t* max_new, const int32_t* stop_ids, int n_stop, float repeat_value,
                                  int penalty_range, float* repeat_penality, int32_t* tokens, float* hidden, int cap,
                                  int32_t* n_out, int mem);

/* tuning hook: time `iters` launches of the implicit-GEMM conv kernel on random device data (no host copies);
 * returns average milliseconds per launch in *ms.  x (B,T,Cin), w (N, taps*Cin), out (B,T,N), "same" padding. */
int         mi_bench_conv_gemm(int dtype, int B, int T, int Cin, int N, int taps, int dil, int with_res, int iters,
                               double* ms);

/* ---- IndexTTS graph A: prompt audio -> conditioning (cond.hip) -----------------------------------------------------------
 * Replaces ort_session_A of IndexTTS/Inference_IndexTTS_ONNX.py:700-712 (graph definition: IndexTTS_A,
 * IndexTTS/Export_IndexTTS.py:74-200): mel front end (0.1 s constant noise pad + 'constant'-padded STFT + HTK mel + log) ->
 * Conformer conditioning encoder (rel-pos attention with rel_shift) -> Perceiver resampler => conds_latent (latents, model_dim),
 * the prompt of the acoustic GPT (graph D's first input); ECAPA-TDNN speaker encoder + the 1x1 conditioning convolutions =>
 * conds = [bigvgan_cond_layer_speaker_embedding (voc_initial) | save_bigvgan_conds_0 | ... | _n-1] (mi_bigvgan_forward_latent
 * takes the same vectors with the speaker-embedding layer's LAST: IndexCond.split_conds / graph F's input order).  fp32 engine.  cfg: mi355tts.config.IndexCondConfig.to_int_array(); weights: the packed
 * blob of mi355tts.weights.pack_cond (export-time folds applied).  mel (optional, may be NULL): the (frames, n_mels) log-mel. */
int64_t     mi_indextts_cond_param_count(const int32_t* cfg, int n_cfg);
mi_cond*    mi_indextts_cond_create(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int device);
void        mi_indextts_cond_destroy(mi_cond* h);
int         mi_indextts_cond_run(mi_cond* h, const int16_t* audio, int64_t L, float* conds, float* conds_latent, float* mel,
                                 int mem);

/* Process-wide tuning / A-B switches (tools, tests).  Thread safety: every entry point that DISPATCHES KERNELS (create / run / step /
 * generate / forward ...) holds the shared side of one reader-writer lock for its duration and mi_set_option the exclusive side — a
 * change waits for the calls in flight on other threads (under sustained concurrent inference that wait is unbounded: the lock
 * prefers readers; set options before serving) and is seen as a whole by the calls that start after it; it never alters the dispatch
 * of a call that is running.  mi_*_destroy, mi_last_error, mi_device_count and mi_version read no option and take no lock.
 *  The arithmetic of an fp32 F5 engine is NOT one of them any more: it is a
 * property of the engine (config int 21, F5Config.f32_arithmetic; mi_f5_info reports what runs) — the four arithmetic keys below
 * only set the default of engines created without one.
 *   arithmetic defaults: "gemm_f32_x3" (1: fp32 linear layers from 16-bit partial products, 0: native v_mfma_f32_32x32x2_f32),
 *     "gemm_f32_planes" (2: fp16 {hi, lo * 2^11} pairs, 22-bit operands, |a| <= 65504 — watched, see mi_f5_info; 3: three bf16
 *     planes, exact), "attn_f32_x3" (2: both attention products split, 1: q.k only, 0: native), "attn_f32_planes" (2 | 3);
 *     "gemm_f32_x3p" (0: the round-2 kernel gemm_x3.hip instead of the panel-plane kernel), "gemm_f32_n64_pairs" /
 *     "gemm_f32_gconv" (the position convolution on fp16 pairs / with each operand split once per workgroup),
 *     "gconv_two_taps" (0: re-split the position convolution's weights per launch instead of using the planes split at load),
 *     "gconv16" (0: the 16-bit engines' position convolution on the generic tile kernel instead of gconv16.hip)
 *   dispatch thresholds of the implicit-GEMM launcher: "gemm_big_tile_min", "gemm_n192_min", "gemm_mid_tile_min",
 *     "gemm_dma3_k_min", "gemm_use_dma3", "gemm_use_dma", "gemm_big_tiles", "gemm_n192", "gemm_f32_dma", "gemm_ring4",
 *     "gemm_ring4_max", "gemm_buf", "gemm_f32_small", "gemm_f32_small_max", "gemm_small16_max", "gemm_sk" (stream-K: 0 off, 1 fp32,
 *     2 also 16-bit), "gemm_sk_stages", "gemm_ph8", "gemm_ph8_min_tiles", "gemm_ph8_order", "gemm_ph8_split_max", "gemm_ph8_split_min_nk" (split tail, tests), "gemm_row_split",
 *     "gemm_x3p_grid" (XCD bands: 0 automatic, else 1 / 2 / 4 / 8 row bands), "gemm_x3p_noalign"
 *   attention: "attn_split" (small grids: 0 plain 128-query workgroups, 1 64-query workgroups with the keys split between wave
 *     pairs + key slices, 2 [default] fp32: 128-query workgroups + key slices, 16-bit as 1), "attn_kv_planes" (0: the fp32 kernel splits K / V itself), "attn_z_force" (key slices, tests),
 *     "attn_xcd_map" (1: a head's query tiles on one XCD; bit-identical either way),
 *     "attn_lpt" (1 [default]: the fp32 128-query kernel's key slices are uneven, longest first — attention.hip attn_pick_slices),
 *     "gemm_x3d" (1 [default]: fp32 linear layers on the exact-fit data-parallel kernel gemm_x3d.hip when the output divides into whole rounds
 *     of the CUs), "gemm_x3d_min_eff" (per cent of useful tile area from which it is taken, default 90)
 *   "gpt_mfma_min" (sentences from which mi_gpt_generate_batch runs its linears on MFMA; default 9)
 *   "bigvgan_streams" (default 3): the AMP blocks of a BigVGAN stage on side streams of the handle — 1: one stream; 2: only
 *     the stages whose AMP halves are separate AA and conv launches (C > 96); 3: every stage.  Bit-identical in all three.
 * Removed in round 4 (variants that lost their A/B, nothing used them): "gemm_sk_min_tiles", "gemm_sk_max_tiles", "gemm_sk_qkv32",
 * "gemm_f32_n64_dma", "gemm_n64_dma16", "attn_z_max", "attn_z16_max",
 * "aa_conv_deterministic".  Changing an option invalidates the hipGraphs captured by existing handles.   */
int         mi_set_option(const char* key, int64_t value);
/* PCI bus id of HIP device `device` ("0000:05:00.0") into buf: multi-rank launchers use it to check that every rank drives its
 * own GPU (mi355tts/shard.py assert_one_device_per_rank).  No reference counterpart (the reference is single-device).   */
int         mi_device_pci_bus_id(int device, char* buf, int cap);

/* ---- profiling hooks (bench.py roofline leg) -------------------------------------------------
 * family_mask: bit i enables family i (0 = off, -1 = all).  Every launch of an enabled kernel
 * family is bracketed by HIP events on the handle's own stream; mi_prof_get returns accumulated
 * milliseconds, launches and algorithmic bytes / flops since the last reset.
 * Families (bit): "conv_gemm"(0) "aa_act"(1) "conv_post"(2) "attn"(3) "norm"(4) "other"(5).    */
int         mi_prof_enable(int family_mask);
int         mi_prof_reset(void);
int         mi_prof_get(const char* family, double* ms, int64_t* launches, double* bytes, double* flops);
/* The same accumulators per kernel instantiation (name = the template instantiation rocprofv3 --kernel-trace lists, e.g.
 * "conv_gemm_dma_kernel<float, float, true, 2, 64>"; launches that do not name their kernel are filed under the family
 * name).  index in [0, mi_prof_kernel_count()).                                                                        */
int         mi_prof_kernel_count(void);
int         mi_prof_kernel_get(int index, char* name, int name_cap, char* family, int family_cap, double* ms,
                               int64_t* launches, double* bytes, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* MI355TTS_H */
