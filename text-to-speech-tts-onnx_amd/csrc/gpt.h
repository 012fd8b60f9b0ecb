// gpt.h — IndexTTS acoustic GPT-2 decoder (graphs B, C, E of IndexTTS/Export_IndexTTS.py:203-289) with the KV cache,
// the repeat-penalty vector and the greedy decode loop (Inference_IndexTTS_ONNX.py:745-783) resident on the device.
#pragma once
#include "common.h"
#include "f5_kernels.h"

namespace mi {

struct GptCfg {
    int hidden, layers, heads, inner, mel_codes, text_tokens, max_mel_pos, max_text_pos, max_seq;
    int max_batch = 1;          // sentences decoded together (slots); optional 10th cfg int
    int head_dim() const { return hidden / heads; }
};
GptCfg parse_gpt_cfg(const int32_t* ci, int ni);
int64_t gpt_param_count(const GptCfg& c);
bool gpt_set_option(const char* key, long v);     // "gpt_mfma_min": sentences from which the batched step uses MFMA

// device-side decode state (one int32 array; kernels read it so that a decode step has no host-dependent argument
// and can be captured once into a hipGraph)
enum { GS_HIST = 0, GS_TOKEN = 1, GS_GEN_LEN = 2, GS_NDEC = 3, GS_RESET = 4, GS_DONE = 5, GS_NSTOP = 6, GS_RANGE = 7,
       GS_UPDATE_PEN = 8, GS_LIMIT = 9, GS_STOP0 = 10, GS_WORDS = 16 };

struct Gpt {
    GptCfg cfg;
    int dtype, device;
    hipStream_t stream = nullptr;

    struct GLin { DevBuf w, b; int n = 0, k = 0; };
    struct Layer { DevBuf ln1_w, ln1_b, ln2_w, ln2_b; GLin qkv, proj, fc, fc2; };
    std::vector<Layer> L;
    DevBuf text_emb, text_pos, mel_emb, mel_pos;       // fp32 tables
    DevBuf lnf_w, lnf_b, fn_w, fn_b;
    GLin head;

    DevBuf kc, vc;            // [slot][layer][head][max_seq][D] in the engine dtype
    DevBuf X, xn, qkv, att, ff;                       // prompt-pass scratch (max_seq rows), shared by the slots
    DevBuf logits, last, pen, toks, hid, state;       // per slot: [slot][codes] / [slot][h] / [slot][max_seq](x h) / words
    DevBuf Xd, xnd, qkvd, attd, ffd, zd;              // batched decode step: one row per slot
    int MBp = 1;              // max_batch rounded up to a batched-GEMV template width
    std::map<int, hipGraphExec_t> batch_graphs;       // decode step over nb slots, keyed by nb
    DevBuf io_a, io_b;        // host<->device staging
    DevBuf rep_dev;           // REPEAT_PENALITY as a device scalar (read by gpt_pick_kernel, also inside replayed graphs)
    void set_rep_value(float v);
    int history = 0;          // host mirror of state[GS_HIST] (valid outside generate())

    hipGraphExec_t step_graph = nullptr;
    bool use_graph = true;
    long graph_epoch = 0;     // option_epoch() the captured graphs were taken under
    void check_graph_epoch();

    Gpt(const GptCfg& c, const float* w, int64_t nw, int dt, int dev);
    ~Gpt();

    void text_embed(const int32_t* ids_dev, int n, float* out_dev);                 // graph B (n + 2 rows)
    void mel_embed(int32_t id, long gen_len, float* out_dev);                        // graph C
    void reset();
    // graph E on rows new positions whose hidden states are already in X[0..rows): fills last / logits / state token
    void forward_rows(int rows, int flag, int slot = 0);
    void set_state(const std::vector<int32_t>& words, int slot = 0);
    std::vector<int32_t> get_state(int slot = 0);
    void decode_batch_eager(int nb);                                                 // one token for slots 0..nb-1
    void decode_batch_steps(int nb, int n);
    void gemv_b(const GLin& l, const void* x, int nb, void* out, int odt, int act, const float* res, void* kcl, void* vcl);
    size_t slot_cache_elems() const { return (size_t)cfg.layers * cfg.hidden * cfg.max_seq; }
    void decode_step_eager();                                                        // C (from state) + E + bookkeeping
    void decode_steps(int n);                                                        // n graph replays
    void kv_read(int layer, float* keys_dev, float* values_dev);                      // slot 0
    void kv_write(int layer, const float* keys_dev, const float* values_dev, int hist);
    void linear(const GLin& l, const void* x, int rows, void* out, int odt, int act, const float* res);
    void gemv(const GLin& l, const void* x, const float* ln_w, const float* ln_b, void* out, int odt, int act,
              const float* res, void* kcl, void* vcl, int slot = 0, const float* pre_w = nullptr, const float* pre_b = nullptr,
              float* pre_out = nullptr);
};

}  // namespace mi
