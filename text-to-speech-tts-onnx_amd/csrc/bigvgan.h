// bigvgan.h — BigVGAN-v2 engine object (see bigvgan.hip).
#pragma once
#include "common.h"

namespace mi {

struct BigVGANCfg {
    int num_mels = 0, c0 = 0, n_up = 0, n_kernels = 0, bias_final = 0, tanh_final = 1, logscale = 1, n_dil = 0;
    int pre_ln = 0, cond = 0;          // IndexTTS graph F: LayerNorm in front, speaker-conditioning biases
    int hop = 1;
    std::vector<int> rates, up_k, res_k;
    std::vector<std::vector<int>> dil;
};
BigVGANCfg parse_bigvgan_cfg(const int32_t* c, int n);
int64_t bigvgan_param_count(const BigVGANCfg& g);

struct ConvW { DevBuf w, b; std::vector<float> hb; DevBuf b_eff; };   // hb / b_eff: bias + per-call speaker conditioning
struct SnakeP { DevBuf alpha, inv_beta; };
struct AmpBlock { int k = 3; std::vector<ConvW> c1, c2; std::vector<SnakeP> acts; };
struct Stage { int cin = 0, cout = 0, u = 1, k = 2; ConvW up; std::vector<AmpBlock> blocks; };

struct BigVGAN {
    BigVGANCfg cfg;
    int dtype, device;
    hipStream_t stream = nullptr;
    int mel_pad = 0;
    ConvW pre;
    std::vector<Stage> stages;
    SnakeP post_act;
    DevBuf post_w;
    float post_bias = 0.f;
    // workspace
    int ws_B = 0, ws_F = 0;
    DevBuf bIN, bX, bT1, bT2, bP, bQ, d_mel, d_out_f32, d_out_i16;
    // the AMP blocks of a stage (one per resblock kernel size) only share their input X and the accumulation into IN: blocks
    // 1.. run on side streams with scratch buffers of their own (option "bigvgan_streams", bigvgan.hip body())
    static constexpr int MAX_SIDE = 3;
    hipStream_t side[MAX_SIDE] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_x = nullptr, ev_last[MAX_SIDE + 1] = {nullptr, nullptr, nullptr, nullptr};
    DevBuf sT1[MAX_SIDE], sT2[MAX_SIDE], sP[MAX_SIDE], sQ[MAX_SIDE];
    hipStream_t ls = nullptr;      // launch stream of conv / aa / aa_conv (null: `stream`)
    void ensure_side(int n);

    BigVGAN(const BigVGANCfg& g, const float* w, int64_t nw, int dt, int dev);
    ~BigVGAN();
    void ensure_workspace(int B, int F);
    void conv(const ConvW& cw, const void* x, void* out, int B, int T, int Cin, int Cout, int k, int dil,
              const void* res, float alpha, int accumulate);
    void aa(const SnakeP& sp, const void* x, void* y, int B, int T, int C, int post);
    void aa_conv(const SnakeP& sp, const ConvW& cw, const void* x, void* out, int B, int T, int C, int k, int dil,
                 const void* res, float alpha, int accumulate);
    int fused_max_c = 96;
    bool use_fused = true;     // MI355TTS_NO_FUSED_AA=1 selects the unfused AA + conv path (A/B and debugging)
    void run(const float* mel, int B, int F, float* out_f32, int16_t* out_i16, int mem);
    // IndexTTS graph F: latent (T_codes, num_mels) channels-last fp32, conds = [cond_0 .. cond_{n_up-1}, cond_pre] concatenated
    void run_latent(const float* latent, int T_codes, const float* conds, long n_conds, float* out_f32, int16_t* out_i16, int mem);
    void body(const void* x0, int B, int F, const float* const* cond_ptrs, float* out_f32, int16_t* out_i16, int mem);
    long total_cond() const;
    DevBuf ln_w, ln_b, d_latent;
};

void unit_aa_activation1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log,
                          int logscale, int post, int dtype, float* y);
void unit_aa_conv1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log, int logscale,
                    const float* w, const float* bias, int k, int dil, const float* res, int dtype, int repeat, float* y);
void unit_conv1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int k, int dil,
                 int padding, int groups, int dtype, float* y);
void unit_conv_transpose1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int k,
                           int stride, int padding, int dtype, float* y);

bool bigvgan_set_option(const char* key, long v);   // "bigvgan_streams"

// runtime.hip
const std::string& last_error();
void prof_enable(unsigned mask);
void prof_reset();
int prof_family(const char* name);
void prof_get(int fam, double* ms, int64_t* launches, double* bytes, double* flops);
int prof_kernel_count();
bool prof_kernel_get(int idx, std::string* name, int* fam, double* ms, int64_t* launches, double* bytes, double* flops);

}  // namespace mi
