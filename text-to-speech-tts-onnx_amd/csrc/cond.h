// cond.h — IndexTTS graph A engine (cond.hip): prompt audio -> conds_latent (GPT conditioning) + BigVGAN conditioning vectors.
#pragma once
#include "common.h"
#include <vector>

namespace mi {

struct CondCfg {
    int n_fft = 1024, hop = 256, mel = 100, sr = 24000, audio_pad = 2400, max_len = 4096;
    int d = 512, heads = 8, lin = 2048, blocks = 6, kern = 15;
    int D = 1280, latents = 32, pdepth = 2, pheads = 8, pdh = 64, pmult = 2;
    int att = 128, r2scale = 8, se = 128, emb = 512, voc0 = 1536;
    std::vector<int> sch, sk, sd, vch;      // ECAPA channels / kernels / dilations ; vocoder stage channels
    float ln_eps = 1e-5f, bn_eps = 1e-5f;
    int dk() const { return d / heads; }
    int inner() const { return pheads * pdh; }
    int ffi() const { return (int)((long)D * pmult * 2 / 3); }
    int f2() const { return (mel - 3) / 2 + 1; }
    int ncond() const { int n = voc0; for (int c : vch) n += c; return n; }
    long frames(long L) const { return (L + audio_pad) / hop + 1; }
};
CondCfg parse_cond_cfg(const int32_t* a, int n);
int64_t cond_param_count(const CondCfg& c);

struct Cond;
Cond* cond_create(const CondCfg& c, const float* w, int64_t nw, int device);      // host blob (mi355tts.weights.pack_cond order)
void cond_destroy(Cond* e);
// audio int16 (L,) host or device -> conds [ncond] = cond_layer | conds_0 | ... ; latent [latents][D]   (host or device outputs)
void cond_run(Cond* e, const int16_t* audio, long L, float* conds, float* latent, float* mel_out /* optional [frames][mel] */, int mem);
int cond_device(const Cond* e);

}  // namespace mi
