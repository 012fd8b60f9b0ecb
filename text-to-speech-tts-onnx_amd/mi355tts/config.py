"""Model configuration records for the two hot-path model families.

The reference keeps these numbers in un-vendored upstream config files
(BigVGAN ``config.json`` read at /root/reference BigVGAN/Export_BigVGAN.py:53,
F5 ``F5TTS_v1_Base.yaml`` read at F5_TTS/Export_F5.py:207, Vocos ``config.yaml``
read at F5_TTS/Export_F5.py:389) and in module-level constants
(F5_TTS/Export_F5.py:43-59).  Here they are explicit dataclasses; the C-ABI
receives them as flat int arrays (see include/mi355tts.h).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List, Optional, Tuple


@dataclass
class BigVGANConfig:
    """BigVGAN-v2 generator hyper-parameters (bigvgan_v2_24khz_100band_256x defaults).

    Mirrors the fields the reference generator reads from ``h``
    (BigVGAN/modeling_modified/bigvgan.py:260-382).
    """
    num_mels: int = 100
    upsample_initial_channel: int = 1536
    upsample_rates: Tuple[int, ...] = (4, 4, 2, 2, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (8, 8, 4, 4, 4, 4)
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    use_bias_at_final: bool = False      # v2 checkpoints: conv_post has no bias
    use_tanh_at_final: bool = True       # forced on by Export_BigVGAN.py:21,41
    snake_logscale: bool = True
    sampling_rate: int = 24000
    # IndexTTS graph F variant (IndexTTS/Export_IndexTTS.py:292-314): the input is the GPT latent sequence, channels-last
    # (T, num_mels=gpt_dim), passed through gpt.final_norm (LayerNorm, affine) first; speaker-conditioning vectors are
    # added after conv_pre and after every upsampler; int16 = trunc(clamp(tanh(.), -1, 1) * 32767).
    pre_layernorm: bool = False
    speaker_cond: bool = False
    ln_eps: float = 1e-5

    @property
    def num_upsamples(self) -> int:
        return len(self.upsample_rates)

    @property
    def num_kernels(self) -> int:
        return len(self.resblock_kernel_sizes)

    @property
    def hop(self) -> int:
        h = 1
        for u in self.upsample_rates:
            h *= u
        return h

    def stage_channels(self, i: int) -> int:
        return self.upsample_initial_channel // (2 ** (i + 1))

    def out_len(self, frames: int) -> int:
        # post activation pads 15 on each side of both FIRs => +30 samples
        # (BigVGAN/modeling_modified/bigvgan.py:370,381-382)
        return frames * self.hop + 30

    def to_int_array(self) -> List[int]:
        """Flat encoding consumed by mi_bigvgan_create (include/mi355tts.h)."""
        out = [self.num_mels, self.upsample_initial_channel, self.num_upsamples, self.num_kernels,
               int(self.use_bias_at_final), int(self.use_tanh_at_final), int(self.snake_logscale)]
        out += list(self.upsample_rates)
        out += list(self.upsample_kernel_sizes)
        out += list(self.resblock_kernel_sizes)
        nd = len(self.resblock_dilation_sizes[0])
        out.append(nd)
        for d in self.resblock_dilation_sizes:
            assert len(d) == nd
            out += list(d)
        out += [int(self.pre_layernorm), int(self.speaker_cond)]
        return out

    @staticmethod
    def indextts() -> "BigVGANConfig":
        """IndexTTS-1.5 vocoder (graph F).  The hyper-parameters live in the un-vendored IndexTTS config.yaml; the
        channel ladder [768..24] is corroborated by IndexTTS/modeling_modified/filter.py:85."""
        return BigVGANConfig(num_mels=1280, upsample_initial_channel=1536, upsample_rates=(4, 4, 4, 4, 2, 2),
                             upsample_kernel_sizes=(8, 8, 4, 4, 4, 4), use_bias_at_final=True, pre_layernorm=True,
                             speaker_cond=True)

    @staticmethod
    def small() -> "BigVGANConfig":
        """Tiny 2-stage generator used by the golden fixtures."""
        return BigVGANConfig(num_mels=16, upsample_initial_channel=32, upsample_rates=(4, 2),
                             upsample_kernel_sizes=(8, 4), use_bias_at_final=True)


# F5Config.f32_arithmetic -> the engine's ARITH_* code (csrc/common.h)
F32_ARITHMETIC = {None: -1, "native-fp32-mfma": 0, "fp16x2-pairs": 2, "bf16x3": 3}
F32_ARITHMETIC_NAMES = {0: "native-fp32-mfma", 2: "fp16x2-pairs", 3: "bf16x3"}
MEL_SPEC_TYPES = {"vocos": 0, "bigvgan": 1}


@dataclass
class F5Config:
    """F5-TTS v1 Base (DiT) + Vocos-mel-24khz + STFT constants.

    DiT arch numbers are *not* in the reference tree; they are
    corroborated by F5_TTS/Optimize_ONNX.py:53-54 (heads 16, hidden 1024) and
    F5_TTS/Export_F5.py:65 (text dim 512).
    """
    # DiT
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    dim_head: int = 64
    ff_mult: int = 2
    mel_dim: int = 100
    text_dim: int = 512
    text_num_embeds: int = 2545          # vocab.txt lines; embedding has +1 rows
    conv_layers: int = 4
    conv_mult: int = 2
    pos_conv_kernel: int = 31
    pos_conv_groups: int = 16
    freq_embed_dim: int = 256
    # sampler (F5_TTS/Export_F5.py:43-59)
    nfe_step: int = 32
    cfg_strength: float = 2.0
    sway_coef: float = -1.0
    max_signal_length: int = 4096
    fuse_step: int = 1                   # FUSE_NFE: Euler steps per F5_Transformer.run (Export_F5.py:167-182, host-side only)
    # use_fp16_transformer (Export_F5.py:20,321-326; F5/fp16/modules.py:467): q / k projections carry an extra x0.1 each, the
    # q k scores are rounded to fp16 and multiplied by 100 in fp32 before the fp32 softmax.  f16 engines only.
    ref_fp16_attn: bool = False
    # fp32 engines: how fp32 products are formed on the matrix cores — a property of the ENGINE (two engines with different
    # arithmetic coexist in one process).  None = the library default (fp16 pairs; mi_set_option can change the default for
    # tools).  "fp16x2-pairs": operands as fp16 {hi, lo} pairs, 22 significant bits, |a| <= 65504 (an engine whose weights or
    # activations leave that range switches itself to "bf16x3", see F5Engine.info()); "bf16x3": exact three-way bf16 split, whole
    # fp32 exponent range; "native-fp32-mfma": v_mfma_f32_32x32x2_f32.
    f32_arithmetic: Optional[str] = None
    # prompt mel front end (modeling_modified/F5/modules.py:30-72 vs Export_F5.py:113,125): "vocos" | "bigvgan"
    mel_spec_type: str = "vocos"
    # AdaLN fold (LayerNorm statistics carried by the GEMM epilogues instead of row-norm launches): None / True = wherever the
    # kernels support it (16-bit engines finish the row statistics with one tiny launch per norm), False = row-norm launches
    adaln_fold: Optional[bool] = None
    # STFT / mel
    n_fft: int = 1024
    hop_length: int = 256
    sample_rate: int = 24000
    # Vocos
    vocos_dim: int = 512
    vocos_intermediate: int = 1536
    vocos_layers: int = 8

    @property
    def ff_dim(self) -> int:
        return self.dim * self.ff_mult

    @property
    def n_freq(self) -> int:
        return self.n_fft // 2 + 1

    def ref_frames(self, n_samples: int) -> int:
        """Mel frames of a prompt of n_samples (= ref_signal_len): vocos-type front end L // hop + 1 (reflect pad n_fft / 2,
        Export_F5.py:122-125); bigvgan-type (L + (n_fft - hop) - n_fft) // hop + 1 (center=False, modules.py:54-68)."""
        if self.mel_spec_type == "bigvgan":
            return (n_samples - self.hop_length) // self.hop_length + 1
        return n_samples // self.hop_length + 1

    def to_int_array(self) -> List[int]:
        return [self.dim, self.depth, self.heads, self.dim_head, self.ff_mult, self.mel_dim, self.text_dim,
                self.text_num_embeds, self.conv_layers, self.conv_mult, self.pos_conv_kernel,
                self.pos_conv_groups, self.freq_embed_dim, self.nfe_step, self.max_signal_length,
                self.n_fft, self.hop_length, self.sample_rate, self.vocos_dim, self.vocos_intermediate,
                self.vocos_layers,
                F32_ARITHMETIC[self.f32_arithmetic], MEL_SPEC_TYPES[self.mel_spec_type],
                -1 if self.adaln_fold is None else int(bool(self.adaln_fold))]

    @property
    def attn_score_scale(self) -> float:
        return 100.0 if self.ref_fp16_attn else 1.0

    def to_float_array(self) -> List[float]:
        return [self.cfg_strength, self.sway_coef] + ([self.attn_score_scale] if self.ref_fp16_attn else [])

    @staticmethod
    def small() -> "F5Config":
        """Reduced-depth model for golden fixtures.  dim/heads stay 1024/16 because the
        reference hard-wires ``.view(2, -1, heads, head_dim)`` shapes only through these."""
        return F5Config(dim=128, depth=2, heads=2, dim_head=64, text_dim=64, text_num_embeds=40,
                        conv_layers=2, pos_conv_groups=2, vocos_dim=64, vocos_intermediate=128,
                        vocos_layers=2, nfe_step=6)


@dataclass
class IndexGPTConfig:
    """IndexTTS-1.5 acoustic GPT-2 (graphs B..E, IndexTTS/Export_IndexTTS.py:203-289).

    The numbers live in the un-vendored IndexTTS ``config.yaml``; the export reads them from the loaded model
    (Export_IndexTTS.py:326-331) and the driver hard-wires the special ids (Inference_IndexTTS_ONNX.py:36-39,
    680: start mel token 8192, stop 8193)."""
    hidden: int = 1280
    layers: int = 24
    heads: int = 20
    inner: int = 5120                   # GPT2Config.n_inner default = 4 * hidden
    mel_codes: int = 8194               # gpt.number_mel_codes (8192 codes + start + stop)
    text_tokens: int = 12001            # text_embedding rows
    max_mel_pos: int = 803              # mel_pos_embedding rows
    max_text_pos: int = 603             # text_pos_embedding rows
    max_seq: int = 1024                 # KV-cache capacity of this engine (>= MAX_GENERATE_LENGTH = 800)
    max_batch: int = 1                  # sentences decoded together by generate_batch (engine extension; <= 16)
    ln_eps: float = 1e-5
    start_mel_token: int = 8192
    stop_mel_token: int = 8193
    max_generate_length: int = 800      # Inference_IndexTTS_ONNX.py:37
    repeat_penalty: float = 0.7         # :38
    penalty_range: int = 10             # :39

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def to_int_array(self) -> List[int]:
        return [self.hidden, self.layers, self.heads, self.inner, self.mel_codes, self.text_tokens, self.max_mel_pos,
                self.max_text_pos, self.max_seq, self.max_batch]

    @staticmethod
    def small() -> "IndexGPTConfig":
        """Reduced model for the golden fixture (head_dim stays 64)."""
        return IndexGPTConfig(hidden=128, layers=2, heads=2, inner=512, mel_codes=50, text_tokens=40, max_mel_pos=40,
                              max_text_pos=24, max_seq=64, start_mel_token=48, stop_mel_token=49,
                              max_generate_length=40)


@dataclass
class IndexCondConfig:
    """IndexTTS-1.5 graph A (IndexTTS/Export_IndexTTS.py:74-200): mel front end -> Conformer conditioning encoder -> Perceiver
    resampler (conds_latent) and ECAPA-TDNN speaker encoder -> BigVGAN conditioning vectors.

    The architecture numbers live in the un-vendored IndexTTS ``config.yaml`` / ``indextts`` package (gpt.condition_module,
    gpt.model_dim, bigvgan.speaker_embedding_dim ...); the defaults are the published IndexTTS-1.5 values — "parity
    unpinned" for them, like the other un-vendored configs.  The export fixes n_fft / hop / mels / rate at
    Export_IndexTTS.py:38-50 and the 0.1 s random ``audio_pad`` at :94."""
    n_fft: int = 1024
    hop: int = 256
    n_mels: int = 100
    sample_rate: int = 24000
    audio_pad: int = 2400               # int(sample_rate * 0.1) samples of N(0,1) prepended to the prompt (:94, :133)
    max_signal_len: int = 4096          # pos_enc.pe rows kept (:87)
    # Conformer conditioning encoder (wenet-style, input_layer conv2d2, rel_pos attention, no macaron)
    enc_dim: int = 512
    enc_heads: int = 8
    enc_linear: int = 2048
    enc_blocks: int = 6
    enc_kernel: int = 15
    ln_eps: float = 1e-5
    # Perceiver resampler
    model_dim: int = 1280
    latents: int = 32
    perc_depth: int = 2
    perc_heads: int = 8
    perc_dim_head: int = 64
    perc_mult: int = 2
    # ECAPA-TDNN speaker encoder (speechbrain-style) inside the vocoder + the conditioning 1x1 convs
    spk_channels: Tuple[int, ...] = (512, 512, 512, 512, 1536)
    spk_kernels: Tuple[int, ...] = (5, 3, 3, 3, 1)
    spk_dilations: Tuple[int, ...] = (1, 2, 3, 4, 1)
    spk_att: int = 128
    spk_res2net_scale: int = 8
    spk_se: int = 128
    spk_embed: int = 512
    bn_eps: float = 1e-5
    voc_initial: int = 1536             # bigvgan.cond_layer out channels (= upsample_initial_channel)
    voc_channels: Tuple[int, ...] = (768, 384, 192, 96, 48, 24)     # bigvgan.conds[i] out channels

    @property
    def enc_dk(self) -> int:
        return self.enc_dim // self.enc_heads

    @property
    def perc_inner(self) -> int:
        return self.perc_heads * self.perc_dim_head

    @property
    def perc_ff(self) -> int:
        return int(self.model_dim * self.perc_mult * 2 / 3)

    @property
    def sub_freq(self) -> int:          # mel bins left by the k3 s2 Conv2d
        return (self.n_mels - 3) // 2 + 1

    def frames(self, audio_len: int) -> int:
        return (audio_len + self.audio_pad) // self.hop + 1

    def enc_len(self, audio_len: int) -> int:
        return (self.frames(audio_len) - 3) // 2 + 1

    def to_int_array(self) -> List[int]:
        # the device engine computes with the epsilons of the published modules (LayerNorm / BatchNorm1d defaults); they are not
        # part of the int array, so a config that changes them must be refused rather than silently ignored
        if self.ln_eps != 1e-5 or self.bn_eps != 1e-5:
            raise ValueError("IndexCondConfig: the HIP engine is built for ln_eps = bn_eps = 1e-5")
        a = [self.n_fft, self.hop, self.n_mels, self.sample_rate, self.audio_pad, self.max_signal_len, self.enc_dim, self.enc_heads,
             self.enc_linear, self.enc_blocks, self.enc_kernel, self.model_dim, self.latents, self.perc_depth, self.perc_heads,
             self.perc_dim_head, self.perc_mult, self.spk_att, self.spk_res2net_scale, self.spk_se, self.spk_embed, self.voc_initial,
             len(self.spk_channels)]
        a += list(self.spk_channels) + list(self.spk_kernels) + list(self.spk_dilations)
        a += [len(self.voc_channels)] + list(self.voc_channels)
        return a

    @staticmethod
    def small() -> "IndexCondConfig":
        """Reduced model for the golden fixture (head dims stay 64 / 32)."""
        return IndexCondConfig(audio_pad=300, max_signal_len=256, enc_dim=64, enc_heads=2, enc_linear=96, enc_blocks=2, enc_kernel=15,
                               model_dim=128, latents=6, perc_depth=2, perc_heads=2, perc_dim_head=32, perc_mult=2,
                               spk_channels=(32, 32, 32, 32, 96), spk_att=16, spk_res2net_scale=4, spk_se=8, spk_embed=24,
                               voc_initial=32, voc_channels=(16, 8))


def as_dict(cfg) -> dict:
    return asdict(cfg)
