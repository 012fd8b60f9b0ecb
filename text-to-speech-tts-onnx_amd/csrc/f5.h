// f5.h — F5-TTS engine object: DiT flow-matching sampler + text/mel front-end + Vocos/ISTFT back-end.
#pragma once
#include "common.h"
#include "f5_kernels.h"

namespace mi {

struct F5Cfg {
    int dim, depth, heads, dim_head, ff_mult, mel, text_dim, vocab, conv_layers, conv_mult, pos_k, pos_g, freq_dim,
        nfe, max_len, n_fft, hop, sr, vd, vi, vlayers;
    float cfg_strength, sway;
    float score_scale = 1.f;   // != 1 (f16 engines only): the reference's fp16-transformer score form, see launch_attention(ref_fp16_scale)
    // optional trailing ints of the config array (mi355tts/config.py F5Config.to_int_array):
    int f32_arith = ARITH_DEFAULT;   // fp32 engines: ARITH_PAIRS | ARITH_BF16X3 | ARITH_NATIVE; ARITH_DEFAULT = the process-wide options (fp16 pairs)
    int mel_type = 0;                // prompt mel front end: 0 vocos (HTK fbank of the magnitude, Export_F5.py:113,125) | 1 bigvgan (slaney, modules.py:30-72)
    int ln_fold = -1;                // AdaLN fold: -1 / 1 on where the kernels support it, 0 off (row-norm launches)
    int ff() const { return dim * ff_mult; }
    int nb() const { return n_fft / 2 + 1; }
    int cat_dim() const { return 2 * mel + text_dim; }     // x | mel | text
    int cond_dim() const { return mel + text_dim; }        // cat_mel_text width
};
F5Cfg parse_f5_cfg(const int32_t* ci, int ni, const float* cf, int nf);
int64_t f5_param_count(const F5Cfg& c);

struct Lin { DevBuf w, b, w3, w3p; int n = 0, k = 0; };      // fp32 engines: w3 = the three bf16 planes of w (gemm_x3.hip), w3p = the same as panel planes (gemm_x3p.hip)

struct F5 {
    F5Cfg cfg;
    int dtype, device;
    hipStream_t stream = nullptr;

    // ---- DiT (engine dtype) ----
    Lin in_proj, gconv1, gconv2, proj_out;
    // row stride of the cat(x, cond, text) buffer = K of in_proj: fp32 engines pad it to whole 64-deep chunks (zero columns)
    // proj_out (dim -> mel = 100 columns: 72 tiles of 64 x 64 at one utterance, each walking all of K with one wave per SIMD) as
    // proj_parts K slices = a grouped launch ([slice][mel][dim / parts] weights, partial sums side by side in `pred`, summed in a fixed
    // order by their consumers): fp32 engines, 4 slices (MI355TTS_PROJ_PARTS=1 at construction: one slice, the A/B switch)
    Lin proj_out_k; int proj_parts = 1; DevBuf pred_sum;
    const float* pred_rows(int U, int N);               // pred as [2U][N][mel] rows (sums the slices when there are any)
    // (MI355TTS_CAT_PAD=0 at construction: unpadded, the A/B switch)
    int cat_pad = 0;
    int cat_ld() const { return cfg.cat_dim() + cat_pad; }
    struct Block { Lin qkv, o, ff1, ff2; };
    std::vector<Block> blocks;
    DevBuf mod;             // fp32 [nfe][depth*6d + 2d]  AdaLN modulation, hoisted out of the loop
    long mod_ld = 0;
    DevBuf delta_t;         // fp32 [nfe-1]
    DevBuf rope_cos, rope_sin;   // fp32 [max_len][dim_head] (rounded through fp16 like the export)
    DevBuf rope_pack;            // the same values as (cos, sin) half pairs, [max_len][dim_head / 2]: 4x fewer table bytes for the QKV epilogue
    std::vector<float> h_time_expand, h_delta;

    // ---- front end (fp32) ----
    DevBuf text_emb, text_pos, stft_w, fbank;
    struct TextBlock { DevBuf dw_w, dw_b, ln_w, ln_b, grn_g, grn_b; Lin pw1, pw2; };
    std::vector<TextBlock> tblocks;

    // ---- Vocos + ISTFT (fp32) ----
    Lin v_embed, v_head, istft;
    DevBuf v_norm_w, v_norm_b, v_fnorm_w, v_fnorm_b, wsi;
    struct VBlock { DevBuf dw_w, dw_b, n_w, n_b; Lin pw1, pw2; };
    std::vector<VBlock> vblocks;

    // ---- HIP graphs of the sampling loop, keyed by (U, N, k0, nsteps) ----
    struct GraphEntry { hipGraphExec_t exec = nullptr; int uses = 0; };
    std::map<std::vector<int>, GraphEntry> graphs;
    bool use_graph = true;          // MI355TTS_NO_GRAPH=1 disables
    long graph_epoch = 0;           // option_epoch() the cached graphs were captured under
    void drop_graphs();
    void recover();          // after an error on this handle: drain the stream, drop graphs, re-zero the hand-off flags / tickets
    void steps_eager(int U, int N, int k0, int nsteps);

    // ---- workspace ----
    SkWorkspace sk;          // stream-K partial-tile slots of this handle's stream
    DevBuf attn_ws, attn_cnt; // key-sliced fp32 attention: partial (m, l, O) per 64-query tile and slice, ticket counters
    long attn_ws_floats = 0, attn_cnt_n = 0;
    int ws_U = 0, ws_N = 0;
    int np = 3;              // planes per operand of the panel-plane GEMMs, fixed when the weights are split (x3p_planes())
    ArithOverride arith;     // this engine's fp32 arithmetic: every call on the handle runs under ArithScope(arith)
    int arith_kind = ARITH_DEFAULT;
    DevBuf Ap, Ap2;          // fp32 engines: the A operand of the big linear layers as panel planes (gemm_x3p.hip): dim / ff columns
    // ---- AdaLN fold (dit_eval; gemm_epilogue.h) ----
    bool fold_built = false; // the load-time vectors exist (dim >= 1024, dim % 128 == 0, cfg.ln_fold != 0)
    DevBuf ApN;              // fp32 engines: x o (1 + scale) of the residual row as panel planes (16-bit engines: rows in Ub)
    DevBuf ln_stats;         // [rows][dim / 32][2] partial (sum, M2 about the block mean) per residual row
    DevBuf ln_fin;           // 16-bit engines: [rows][2] finished (rstd, mean * rstd) (launch_ln_finalize: one tiny launch per norm)
    DevBuf ln_tab;           // [nfe][depth][ W_qkv (1 + sc_a) : 3d | W_qkv sh_a + b : 3d | W_ff1 (1 + sc_m) : ff | W_ff1 sh_m + b : ff ]
    long ln_ld = 0, ln_blk = 0;
    DevBuf d_sat;            // fp16-pair producers raise word 0 when an operand met the fp16 range limit (x3_split.h sat_publish)
    long sat_events = 0;     // calls on this handle that tripped it (and were re-run on three bf16 planes)
    void build_ln_tables(int i, DevBuf& G, DevBuf& S, DevBuf& tmpw);
    void set_arith(int kind);                 // (re)split the big matrices for ARITH_PAIRS / ARITH_BF16X3 / ARITH_NATIVE
    bool take_saturation();                   // after a stream synchronisation: did a producer flag a saturated operand?  (clears it)
    void finish_call();                       // ... plus the stream-K watchdog and the text-id flag: every C-ABI entry ends with it
    DevBuf d_noise, d_cmt, d_cmtd, cat, h32, hT, c1, X, Ub, qb, kb, vb, Ob, Hff, pred;
    DevBuf p_audio, p_pad, p_spec, p_mag, p_mel, p_ids, p_tid, p_err, p_tx, p_ty, p_ty2, p_ss;
    std::vector<float> h_noise;
    DevBuf v_h, v_z, v_z2, v_s, v_c, v_fr, v_outf, v_outi;

    F5(const F5Cfg& c, const float* w, int64_t nw, int dt, int dev, int mem = MI_HOST);   // w: host or device blob
    ~F5();
    void ensure_workspace(int U, int N);
    void gemm(int dt, const void* x, long xb, long xr, int K, const Lin& L, void* out, int odt, long ob, long orr,
              int B, int M, int act = ACT_NONE, const void* res = nullptr, const float* gate = nullptr);
    // fills d_noise, d_cmt, d_cmtd for U utterances (asynchronous on `stream`); returns ref_signal_len
    int preprocess(int U, const int16_t* audio, long L, const int32_t* text_ids, int T, int N,
                   const float* noise_in, uint64_t seed, int mem);
    void stft(const int16_t* audio_dev, int U, long L, int pad = -1);     // -> p_spec [u][frame][re | im]; pad: reflect padding each side (-1: n_fft / 2)
    void check_text_ids();
    void load_cond(const float* noise, const float* cmt, const float* cmtd, int U, int N, int mem);
    void build_cat_cond(int U, int N);
    void dit_eval(int U, int N, int k);                 // pred <- DiT(noise, cond, t_k)
    void steps(int U, int N, int k0, int nsteps);       // Euler/CFG updates k0 .. k0+nsteps-1 (in d_noise)
    long decode(const float* denoised_dev, int U, int N, int R, float* out_f, int16_t* out_i);  // device outputs
};

}  // namespace mi
