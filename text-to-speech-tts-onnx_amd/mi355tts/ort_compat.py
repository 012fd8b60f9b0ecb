"""Drop-in stand-in for the slice of the ``onnxruntime`` Python API the reference drivers use.

    import mi355tts.ort_compat as onnxruntime

then the bodies of /root/reference F5_TTS/F5-TTS-ONNX-Inference.py (:152-311) and
BigVGAN/Export_BigVGAN.py (:77-177) run unchanged with model paths pointing at ``*.mi355.json``
manifests written by :func:`save_model`.  Surface mirrored:

    set_seed, SessionOptions (+add_session_config_entry), GraphOptimizationLevel, ExecutionMode,
    InferenceSession(path, sess_options=, providers=, provider_options=)
        .get_inputs() / .get_outputs() -> objects with .name/.type/.shape ; ._inputs_meta ; .get_providers()
        .run(output_names, {name: ndarray}) -> [ndarray]
        .run_with_ort_values(output_names, {name: OrtValue}) -> [OrtValue]
        .io_binding() / .run_with_iobinding(binding)
    OrtValue.ortvalue_from_numpy(arr, device_type, device_id) / .numpy() / .update_inplace / .shape / .data_type / .device_name
        device_type "cuda": the value lives in HBM; graph B (run_with_iobinding / run_with_ort_values) and BigVGAN
        (run_with_ort_values) then pass device pointers through the C-ABI and write bound outputs in place

Graph tensor names, dtypes and dynamic axes are those of the exports (F5_TTS/Export_F5.py:294-305,
354-365, 409-414; BigVGAN/Export_BigVGAN.py:65-70).  Errors raise (InvalidArgument / Fail), like ORT.
Execution is always the MI355X engine ("MI355XExecutionProvider"); there is no CPU provider.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from .config import BigVGANConfig, F5Config

_SEED = [9527]
_SEED_DRAWS = [0]          # preprocess calls since set_seed (onnxruntime's generator advances with every draw)
PROVIDER = "MI355XExecutionProvider"


class InvalidArgument(ValueError):
    pass


class Fail(RuntimeError):
    pass


def set_seed(seed: int) -> None:
    _SEED[0] = int(seed)
    _SEED_DRAWS[0] = 0


def get_available_providers() -> List[str]:
    return [PROVIDER]


class GraphOptimizationLevel:
    ORT_DISABLE_ALL, ORT_ENABLE_BASIC, ORT_ENABLE_EXTENDED, ORT_ENABLE_ALL = 0, 1, 2, 99


class ExecutionMode:
    ORT_SEQUENTIAL, ORT_PARALLEL = 0, 1


class SessionOptions:
    """Attribute bag: the engine has no graph optimiser or thread pools to configure."""

    def __init__(self):
        self.log_severity_level = 2
        self.log_verbosity_level = 0
        self.inter_op_num_threads = 0
        self.intra_op_num_threads = 0
        self.enable_cpu_mem_arena = True
        self.execution_mode = ExecutionMode.ORT_SEQUENTIAL
        self.graph_optimization_level = GraphOptimizationLevel.ORT_ENABLE_ALL
        self.config_entries: Dict[str, str] = {}

    def add_session_config_entry(self, key: str, value: str) -> None:
        if not isinstance(key, str) or not isinstance(value, str):
            raise InvalidArgument("session config entries are (str, str)")
        self.config_entries[key] = value


class NodeArg:
    def __init__(self, name: str, type_: str, shape):
        self.name, self.type, self.shape = name, type_, list(shape)

    def __repr__(self):
        return f"NodeArg(name='{self.name}', type='{self.type}', shape={self.shape})"


_DEVICE_TYPES = ("cuda", "rocm", "hip", "gpu")


def _unbroadcast(a: np.ndarray):
    """(base, index) of a numpy view with stride-0 axes: `base` = the view with those axes reduced to length 1 — what has to
    cross PCIe; the caller expands it again on the device (graph A's RoPE tables are (N, 64) tables broadcast to (2, 16, N, 64):
    9.2 MB each as a dense copy, 288 KB as what they are)."""
    idx = tuple(slice(0, 1) if (st == 0 and n > 1) else slice(None) for st, n in zip(a.strides, a.shape))
    return a[idx]


class OrtValue:
    """``onnxruntime.OrtValue``: host (numpy) or DEVICE-resident.  ``ortvalue_from_numpy(arr, "cuda", id)`` copies the array into
    HBM (a torch-ROCm tensor holds it; torch is I/O plumbing here) as the reference's io-binding branch does
    (F5-TTS-ONNX-Inference.py:257-266); sessions hand the device pointer to the C-ABI (``MI_DEVICE``) and write bound outputs in
    place.  ``numpy()`` of a device value is a D2H copy."""

    def __init__(self, arr: Optional[np.ndarray] = None, device_type: str = "cpu", device_id: int = 0, tensor=None):
        self._host, self._t, self._device, self._device_id = arr, tensor, device_type, device_id
        self._mirror = None          # host copy of a small device tensor (time_step): saves a D2H per graph-B call
        self._rope_ok = None         # (N, which) once the session has recognised a RoPE table in this value

    @staticmethod
    def ortvalue_from_numpy(arr, device_type: str = "cpu", device_id: int = 0) -> "OrtValue":
        arr = np.asarray(arr)
        dt = (device_type or "cpu").lower()
        if dt not in _DEVICE_TYPES:
            return OrtValue(arr, "cpu", device_id)
        import torch
        base = _unbroadcast(arr)
        base = np.ascontiguousarray(base) if base.flags.writeable else np.array(base, order="C")     # (torch refuses read-only views)
        t = torch.from_numpy(base).to(torch.device("cuda", int(device_id)))
        if base.shape != arr.shape:
            t = t.expand(*arr.shape)
        v = OrtValue(None, "cuda", int(device_id), tensor=t)
        if arr.size <= 16:
            v._mirror = np.array(arr, copy=True)
        return v

    @staticmethod
    def ortvalue_from_shape_and_type(shape, element_type=np.float32, device_type: str = "cpu", device_id: int = 0) -> "OrtValue":
        return OrtValue.ortvalue_from_numpy(np.zeros(tuple(shape), dtype=element_type), device_type, device_id)

    @staticmethod
    def _from_tensor(t) -> "OrtValue":
        return OrtValue(None, "cuda", t.device.index or 0, tensor=t)

    def is_device(self) -> bool:
        return self._t is not None

    def numpy(self) -> np.ndarray:
        if self._t is not None:
            return self._t.cpu().numpy()
        return self._host

    # (the name the host-only round-4 class used; _KVRef overrides it)
    @property
    def _arr(self):
        return self.numpy()

    @_arr.setter
    def _arr(self, v):
        self._host, self._t, self._mirror = np.asarray(v), None, None

    def update_inplace(self, np_arr) -> None:
        np_arr = np.asarray(np_arr)
        if self._t is not None:
            import torch
            self._t.copy_(torch.from_numpy(np.ascontiguousarray(np_arr)).to(self._t.device))
            self._mirror = np.array(np_arr, copy=True) if np_arr.size <= 16 else None
            self._rope_ok = None
        else:
            self._host[...] = np_arr

    def shape(self):
        return list(self._t.shape) if self._t is not None else list(self._host.shape)

    def data_type(self) -> str:
        dt = str(self._t.dtype).replace("torch.", "") if self._t is not None else str(self._host.dtype)
        return {"float32": "tensor(float)", "float16": "tensor(float16)", "int16": "tensor(int16)", "int32": "tensor(int32)",
                "int64": "tensor(int64)", "int8": "tensor(int8)"}.get(dt, f"tensor({dt})")

    def data_ptr(self) -> int:
        return self._t.data_ptr() if self._t is not None else self._host.ctypes.data

    def device_name(self) -> str:
        return self._device


class IOBinding:
    def __init__(self, session):
        self._s, self._in, self._out, self._results = session, {}, {}, []

    def bind_ortvalue_input(self, name: str, ortvalue: OrtValue):
        self._in[name] = ortvalue

    def bind_ortvalue_output(self, name: str, ortvalue: OrtValue):
        self._out[name] = ortvalue

    def get_outputs(self):
        return self._results

    def copy_outputs_to_cpu(self):
        return [r.numpy() for r in self._results]


# ---------------------------------------------------------------------------------------------------
# model manifests
# ---------------------------------------------------------------------------------------------------
def save_model(path: str, graph: str, cfg, weights_file: str, dtype: str = "f32") -> str:
    """Write a ``*.mi355.json`` manifest.  `weights_file` is a ``.npy`` holding the canonical fp32 blob
    (mi355tts.weights.pack_bigvgan / pack_f5); several graphs may share one file."""
    from dataclasses import asdict
    if graph not in _GRAPHS:
        raise InvalidArgument(f"unknown graph {graph}")
    man = {"format": "mi355tts-1", "graph": graph, "dtype": dtype, "config": asdict(cfg),
           "weights": os.path.relpath(weights_file, os.path.dirname(os.path.abspath(path)))}
    with open(path, "w") as f:
        json.dump(man, f, indent=1)
    return path


_ENGINES: Dict[tuple, object] = {}


def _engine(kind: str, cfg, wfile: str, dtype: str, device: int):
    key = (kind, os.path.abspath(wfile), dtype, device, bool(getattr(cfg, "ref_fp16_attn", False)))
    if key not in _ENGINES:
        blob = np.load(wfile, mmap_mode="r")
        if kind == "f5":
            from .f5 import F5Engine
            _ENGINES[key] = F5Engine(cfg, blob=np.asarray(blob), dtype=dtype, device=device)
        elif kind == "gpt":
            from .indextts import IndexGPT
            _ENGINES[key] = IndexGPT(cfg, blob=np.asarray(blob), dtype=dtype, device=device)
        elif kind == "cond":
            from .indextts import IndexCond
            _ENGINES[key] = IndexCond(cfg, blob=np.asarray(blob), device=device)
        else:
            from .bigvgan import BigVGANVocoder
            _ENGINES[key] = BigVGANVocoder(cfg, blob=np.asarray(blob), dtype=dtype, device=device)
    return _ENGINES[key]


def register_engine(kind: str, cfg, weights_file: str, dtype: str, device: int, engine) -> None:
    """Hand an engine that already exists (e.g. built from a blob in HBM) to the sessions that will be opened on manifests naming
    `weights_file` — the file itself is then never read (bench_detail.py times the façade this way without writing 1.4 GB)."""
    _ENGINES[(kind, os.path.abspath(weights_file), dtype, device, bool(getattr(cfg, "ref_fp16_attn", False)))] = engine


def _tname(dtype: str) -> str:
    return {"f32": "tensor(float)", "f16": "tensor(float16)", "bf16": "tensor(bfloat16)"}[dtype]


def _graph_io(graph: str, cfg, dtype: str):
    """(inputs, outputs) as NodeArg lists, names/ranks/dynamic axes as in the reference's exports."""
    if graph == "BigVGAN":
        return ([NodeArg("mel_features", "tensor(float)", [1, cfg.num_mels, "mel_features_len"])],
                [NodeArg("generated_wav", "tensor(int16)", [1, 1, "generated_len"])])
    if graph == "IndexTTS_A":          # IndexTTS/Export_IndexTTS.py:337-355
        outs = [NodeArg(f"save_bigvgan_conds_{i}", "tensor(float)", [1, ch, 1]) for i, ch in enumerate(cfg.voc_channels)]
        outs.append(NodeArg("bigvgan_cond_layer_speaker_embedding", "tensor(float)", [1, cfg.voc_initial, 1]))
        outs.append(NodeArg("conds_latent", "tensor(float)", [1, "ref_signal_len", cfg.model_dim]))
        return [NodeArg("audio", "tensor(int16)", [1, 1, "audio_len"])], outs
    if graph == "IndexTTS_F":          # IndexTTS/Export_IndexTTS.py:497-520
        ins = [NodeArg(f"save_bigvgan_conds_{i}", "tensor(float)", [1, cfg.stage_channels(i), 1])
               for i in range(cfg.num_upsamples)]
        ins.append(NodeArg("bigvgan_cond_layer_speaker_embedding", "tensor(float)", [1, cfg.upsample_initial_channel, 1]))
        ins.append(NodeArg("save_hidden_state", "tensor(float)", ["kv_seq_len", cfg.num_mels]))
        return ins, [NodeArg("generated_wav", "tensor(int16)", [1, 1, "generated_len"])]
    if graph in _GPT_GRAPHS:           # IndexTTS/Export_IndexTTS.py:352-482
        ft, h = "tensor(float)", cfg.hidden
        if graph == "IndexTTS_B":
            return ([NodeArg("text_ids", "tensor(int32)", [1, "text_ids_len"])],
                    [NodeArg("text_hidden_state", ft, [1, "text_ids_len_plus_2", h])])
        if graph == "IndexTTS_C":
            return ([NodeArg("gpt_ids", "tensor(int32)", [1, 1]), NodeArg("kv_seq_len", "tensor(int64)", [1])],
                    [NodeArg("gpt_hidden_state", ft, [1, 1, h]), NodeArg("next_kv_seq_len", "tensor(int64)", [1])])
        if graph == "IndexTTS_D":
            return ([NodeArg("embed_x", ft, [1, "embed_x_len", h]), NodeArg("embed_y", ft, [1, "embed_y_len", h]),
                     NodeArg("embed_z", ft, [1, "embed_z_len", h])],
                    [NodeArg("concat_hidden_state", ft, [1, "concat_len", h]), NodeArg("concat_len", "tensor(int64)", [1])])
        L, H, D = cfg.layers, cfg.heads, cfg.head_dim
        ins = [NodeArg(f"in_key_{i}", ft, [H, D, "history_len"]) for i in range(L)]
        ins += [NodeArg(f"in_value_{i}", ft, [H, "history_len", D]) for i in range(L)]
        ins += [NodeArg("history_len", "tensor(int64)", [1]), NodeArg("repeat_penality", ft, [1, cfg.mel_codes]),
                NodeArg("ids_len", "tensor(int64)", [1]), NodeArg("hidden_state", ft, [1, "ids_len", h]),
                NodeArg("attention_mask", "tensor(int8)", [1])]
        outs = [NodeArg(f"out_key_{i}", ft, [H, D, "history_len_plus_ids_len"]) for i in range(L)]
        outs += [NodeArg(f"out_value_{i}", ft, [H, "history_len_plus_ids_len", D]) for i in range(L)]
        outs += [NodeArg("kv_seq_len", "tensor(int64)", [1]), NodeArg("last_hidden_state", ft, [1, h]),
                 NodeArg("max_logit_id", "tensor(int32)", [1, 1])]
        return ins, outs
    H, D, M, cd = cfg.heads, cfg.dim_head, cfg.mel_dim, cfg.mel_dim + cfg.text_dim
    # the engine keeps graph I/O in fp32 whatever the DiT operand dtype — except for the reference's fp16-transformer export
    # (cfg.ref_fp16_attn; Export_F5.py:139-140,198-199,348-349), whose graphs exchange noise / RoPE tables / cat_mel_text /
    # denoised as float16: the sessions then declare, produce and accept float16 (values rounded exactly where that export
    # rounds them: at the graph boundaries)
    ft = "tensor(float16)" if getattr(cfg, "ref_fp16_attn", False) else "tensor(float)"
    cond = [NodeArg("noise", ft, [1, "max_duration", M]), NodeArg("rope_cos_q", ft, [2, H, "max_duration", D]),
            NodeArg("rope_sin_q", ft, [2, H, "max_duration", D]), NodeArg("rope_cos_k", ft, [2, H, D, "max_duration"]),
            NodeArg("rope_sin_k", ft, [2, H, D, "max_duration"]), NodeArg("cat_mel_text", ft, [1, "max_duration", cd]),
            NodeArg("cat_mel_text_drop", ft, [1, "max_duration", cd])]
    if graph == "F5_Preprocess":
        return ([NodeArg("audio", "tensor(int16)", [1, 1, "audio_len"]), NodeArg("text_ids", "tensor(int32)", [1, "text_ids_len"]),
                 NodeArg("max_duration", "tensor(int64)", [1])], cond + [NodeArg("ref_signal_len", "tensor(int64)", [])])
    if graph == "F5_Transformer":
        return (cond + [NodeArg("time_step", "tensor(int32)", [1])],
                [NodeArg("denoised", ft, [1, "max_duration", M]), NodeArg("time_step", "tensor(int32)", [1])])
    if graph == "F5_Decode":
        return ([NodeArg("denoised", ft, [1, "max_duration", M]), NodeArg("ref_signal_len", "tensor(int64)", [])],
                [NodeArg("output_audio", "tensor(int16)", [1, 1, "generated_len"])])
    raise InvalidArgument(graph)


_GPT_GRAPHS = ("IndexTTS_B", "IndexTTS_C", "IndexTTS_D", "IndexTTS_E")
_GRAPHS = ("BigVGAN", "IndexTTS_A", "IndexTTS_F", "F5_Preprocess", "F5_Transformer", "F5_Decode") + _GPT_GRAPHS


class _KVRef(OrtValue):
    """out_key_i / out_value_i of graph E: a reference to the cache held by the engine instead of a copy (the
    reference round-trips 2 * layers growing tensors through every call, Inference_IndexTTS_ONNX.py:765-766).  Fed back
    as in_key_i / in_value_i it costs nothing; ``.numpy()`` materialises it while it is still current."""

    def __init__(self, sess, layer: int, is_value: bool, length: int, epoch: int):
        self._s, self._layer, self._is_value, self._len, self._epoch = sess, layer, is_value, length, epoch
        self._device, self._device_id = "cpu", 0
        self._host = self._t = self._mirror = self._rope_ok = None

    def current(self) -> bool:
        return self._epoch == self._s._kv_epoch and self._len == self._s._eng.history_len

    def numpy(self) -> np.ndarray:
        if not self.current():
            raise Fail("this out_key/out_value refers to a KV-cache state the session has moved past")
        k, v = self._s._eng.kv_read(self._layer)
        return v if self._is_value else k

    @property
    def _arr(self):
        return self.numpy()

    def shape(self):
        c = self._s._cfg
        return [c.heads, self._len, c.head_dim] if self._is_value else [c.heads, c.head_dim, self._len]


class InferenceSession:
    def __init__(self, path_or_bytes, sess_options: Optional[SessionOptions] = None, providers: Optional[Sequence] = None,
                 provider_options: Optional[Sequence[dict]] = None, **kwargs):
        if not isinstance(path_or_bytes, (str, os.PathLike)) or not os.path.exists(path_or_bytes):
            raise Fail(f"Load model from {path_or_bytes} failed: file does not exist (expected a *.mi355.json manifest)")
        with open(path_or_bytes) as f:
            man = json.load(f)
        if man.get("format") != "mi355tts-1" or man.get("graph") not in _GRAPHS:
            raise InvalidArgument(f"{path_or_bytes}: not an mi355tts manifest")
        self._graph, self._dtype = man["graph"], man.get("dtype", "f32")
        self._opts = sess_options or SessionOptions()
        device = 0
        for po in (provider_options or []):
            if isinstance(po, dict) and "device_id" in po:
                device = int(po["device_id"])
        c = dict(man["config"])
        if self._graph in ("BigVGAN", "IndexTTS_F"):
            for k in ("upsample_rates", "upsample_kernel_sizes", "resblock_kernel_sizes"):
                c[k] = tuple(c[k])
            c["resblock_dilation_sizes"] = tuple(tuple(d) for d in c["resblock_dilation_sizes"])
            self._cfg = BigVGANConfig(**c)
        elif self._graph in _GPT_GRAPHS:
            from .config import IndexGPTConfig
            self._cfg = IndexGPTConfig(**c)
        elif self._graph == "IndexTTS_A":
            from .config import IndexCondConfig
            for k in ("spk_channels", "spk_kernels", "spk_dilations", "voc_channels"):
                c[k] = tuple(c[k])
            self._cfg = IndexCondConfig(**c)
        else:
            self._cfg = F5Config(**c)
        wfile = os.path.join(os.path.dirname(os.path.abspath(path_or_bytes)), man["weights"])
        kind = ("bigvgan" if self._graph in ("BigVGAN", "IndexTTS_F") else "gpt" if self._graph in _GPT_GRAPHS else
                "cond" if self._graph == "IndexTTS_A" else "f5")
        self._eng = None if self._graph == "IndexTTS_D" else _engine(kind, self._cfg, wfile, self._dtype, device)
        self._kv_epoch = 0
        self._rope_tabs, self._rope_seen = {}, {}
        self._inputs, self._outputs = _graph_io(self._graph, self._cfg, self._dtype)
        self._inputs_meta, self._outputs_meta = self._inputs, self._outputs

    # ---- metadata ---------------------------------------------------------------------------------
    def get_inputs(self):
        return self._inputs

    def get_outputs(self):
        return self._outputs

    def get_providers(self):
        return [PROVIDER]

    def io_binding(self):
        return IOBinding(self)

    # ---- execution ----------------------------------------------------------------------------------
    def run(self, output_names, input_feed: Dict[str, np.ndarray], run_options=None, _keep_refs: bool = False):
        names = [o.name for o in self._outputs]
        want = list(output_names) if output_names else names
        for n in want:
            if n not in names:
                raise InvalidArgument(f"Invalid output name: {n}")
        need = [i.name for i in self._inputs]
        missing = [n for n in need if n not in input_feed]
        if missing:
            raise InvalidArgument(f"Required inputs ({missing}) are missing from input feed ({list(input_feed)}).")
        extra = [n for n in input_feed if n not in need]
        if extra:
            raise InvalidArgument(f"Invalid input name: {extra[0]}")
        if self._graph == "IndexTTS_E":
            res = self._run_gpt_e(input_feed)
            return [res[n] if _keep_refs or not isinstance(res[n], OrtValue) else res[n].numpy() for n in want]
        res = self._run(input_feed)
        return [res[n] for n in want]

    def run_with_ort_values(self, output_names, input_feed: Dict[str, OrtValue], run_options=None):
        if self._graph == "IndexTTS_E":            # KV tensors stay references
            outs = self.run(output_names, dict(input_feed), _keep_refs=True)
            return [o if isinstance(o, OrtValue) else OrtValue(o) for o in outs]
        if self._graph in ("F5_Transformer", "BigVGAN") and input_feed and all(isinstance(v, OrtValue) and v.is_device() for v in input_feed.values()):
            self._check_names(output_names, input_feed)
            res = self._run_f5_transformer_device(input_feed, {}) if self._graph == "F5_Transformer" else self._run_bigvgan_device(input_feed)
            return [res[n] for n in (list(output_names) if output_names else [o.name for o in self._outputs])]
        outs = self.run(output_names, {k: v.numpy() for k, v in input_feed.items()})
        return [OrtValue(o) for o in outs]

    def run_with_iobinding(self, binding: IOBinding, run_options=None):
        names = list(binding._out) or [o.name for o in self._outputs]
        if self._graph == "F5_Transformer" and binding._in and all(v.is_device() for v in binding._in.values()) and \
                all(v.is_device() for v in binding._out.values()):
            # the reference's io-binding loop (F5-TTS-ONNX-Inference.py:268-288): every operand already lives in HBM, the bound
            # outputs alias inputs 0 and 7 — device pointers cross the C-ABI, nothing visits the host
            self._check_names(names, binding._in)
            res = self._run_f5_transformer_device(binding._in, binding._out)
            binding._results = [res[n] for n in names]
            return
        outs = self.run(list(binding._out) or None, {k: v.numpy() for k, v in binding._in.items()})
        binding._results = []
        for n, o in zip(names, outs):
            if n in binding._out:                       # the reference aliases outputs onto input buffers
                tgt = binding._out[n]
                if tgt.is_device():
                    if list(tgt._t.shape) == list(o.shape):
                        tgt.update_inplace(o)
                    else:                               # a device-bound output of another shape is re-allocated ON the device (it stays a device value)
                        import torch
                        tgt._t = torch.from_numpy(np.ascontiguousarray(o)).to(tgt._t.device)
                        tgt._mirror = np.array(o, copy=True) if o.size <= 16 else None
                    binding._results.append(tgt)
                    continue
                elif tgt._host.shape == o.shape and tgt._host.dtype == o.dtype:
                    tgt._host[...] = o
                    binding._results.append(tgt)
                    continue
                tgt._arr = o
                binding._results.append(tgt)
                continue
            binding._results.append(OrtValue(o))

    def _check_names(self, output_names, input_feed):
        names = [o.name for o in self._outputs]
        for n in (output_names or []):
            if n not in names:
                raise InvalidArgument(f"Invalid output name: {n}")
        need = [i.name for i in self._inputs]
        missing = [n for n in need if n not in input_feed]
        if missing:
            raise InvalidArgument(f"Required inputs ({missing}) are missing from input feed ({list(input_feed)}).")
        extra = [n for n in input_feed if n not in need]
        if extra:
            raise InvalidArgument(f"Invalid input name: {extra[0]}")

    # ---- device-resident execution -------------------------------------------------------------------------------
    def _dev_tensor(self, v: OrtValue, name: str, dtype, ndim: int):
        t = v._t
        if t.dtype != dtype:
            raise InvalidArgument(f"Unexpected input data type. Actual: ({t.dtype}) , expected: ({dtype}) for input {name}")
        if t.dim() != ndim:
            raise InvalidArgument(f"Invalid rank for input: {name} Got: {t.dim()} Expected: {ndim}")
        if t.device.index != self._eng.device:
            raise InvalidArgument(f"{name} lives on cuda:{t.device.index}, the session runs on cuda:{self._eng.device}")
        return t

    def _run_f5_transformer_device(self, feed: Dict[str, OrtValue], bound_out: Dict[str, OrtValue]):
        """graph B with every operand in HBM (Export_F5.py:167-182): the sampler state `noise` is advanced IN PLACE in the output
        value — the input value itself when the caller bound it as the output (the reference's aliasing), else a device copy."""
        import torch
        e, cfg = self._eng, self._cfg
        if getattr(cfg, "ref_fp16_attn", False):      # the float16-I/O export keeps its host path (values are rounded at the graph edges)
            # ... but the io-binding contract holds here too: a BOUND output is written in place (the reference binds outputs 0 / 1
            # onto inputs 0 / 7 and relies on exactly that to advance the sampler, F5-TTS-ONNX-Inference.py:268-288)
            outs = self.run(None, {k: v.numpy() for k, v in feed.items()})
            res = {}
            for o, a in zip(self._outputs, outs):
                tgt = bound_out.get(o.name)
                if tgt is None:
                    res[o.name] = OrtValue.ortvalue_from_numpy(a, "cuda", e.device)
                    continue
                if tgt._t is None or list(tgt._t.shape) != list(a.shape) or str(tgt._t.dtype).replace("torch.", "") != str(a.dtype):
                    tgt._t = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda", e.device))
                    tgt._host, tgt._device = None, "cuda"
                    tgt._mirror = np.array(a, copy=True) if a.size <= 16 else None
                else:
                    tgt.update_inplace(a)
                res[o.name] = tgt
            return res
        noise = self._dev_tensor(feed["noise"], "noise", torch.float32, 3)
        cmt = self._dev_tensor(feed["cat_mel_text"], "cat_mel_text", torch.float32, 3)
        cmtd = self._dev_tensor(feed["cat_mel_text_drop"], "cat_mel_text_drop", torch.float32, 3)
        tsv = feed["time_step"]
        self._dev_tensor(tsv, "time_step", torch.int32, 1)
        U, N, M = noise.shape
        cd = cfg.mel_dim + cfg.text_dim
        if M != cfg.mel_dim or tuple(cmt.shape) != (U, N, cd) or tuple(cmtd.shape) != (U, N, cd):
            raise InvalidArgument(f"noise must be (U, N, {cfg.mel_dim}) and cat_mel_text(_drop) (U, N, {cd})")
        for n in ("rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k"):
            v = feed[n]
            if v._rope_ok != (N, n):                    # recognised once per value: the tables never change between the 31 calls
                t = v._t
                if t.dim() != 4:
                    raise InvalidArgument(f"Invalid rank for input: {n}")
                got = (t[0, 0] if n.endswith("_q") else t[0, 0].T).float().cpu().numpy()
                want = self._rope_table(N, "cos" in n)
                if got.shape != want.shape or np.abs(got - want).max() > 2e-3:
                    raise InvalidArgument(f"{n}: not the RoPE table of F5_Preprocess (this engine regenerates the tables on "
                                          f"the device and cannot honour modified ones)")
                v._rope_ok = (N, n)
        if tsv._mirror is None:
            tsv._mirror = tsv._t.cpu().numpy()
        k = int(tsv._mirror.reshape(-1)[0])
        fuse = max(1, int(getattr(cfg, "fuse_step", 1)))
        out_x = bound_out.get("denoised")
        if out_x is None:
            out_x = OrtValue._from_tensor(noise.clone())
        elif out_x is not feed["noise"]:
            if list(out_x._t.shape) != [U, N, M] or out_x._t.dtype != torch.float32:
                out_x._t = torch.empty_like(noise)
            out_x._t.copy_(noise)
        x = out_x._t
        if not (x.is_contiguous() and cmt.is_contiguous() and cmtd.is_contiguous()):
            raise InvalidArgument("device-resident graph-B operands must be contiguous")
        new_k = e.transformer_step_device(x, cmt, cmtd, k, fuse)
        out_t = bound_out.get("time_step")
        if out_t is None:
            out_t = OrtValue.ortvalue_from_numpy(np.array([new_k], dtype=np.int32), "cuda", e.device)
        else:
            out_t._t.fill_(new_k)
            out_t._mirror = np.array([new_k], dtype=np.int32)
        return {"denoised": out_x, "time_step": out_t}

    def _run_bigvgan_device(self, feed: Dict[str, OrtValue]):
        """ort_session_A.run_with_ort_values([generated_wav], {mel_features: <device value>}) — BigVGAN/Export_BigVGAN.py:170."""
        import torch
        mel = feed["mel_features"]._t
        if mel.dtype not in (torch.float32, torch.float16) or mel.dim() != 3:
            raise InvalidArgument("mel_features must be a rank-3 float tensor")
        if mel.device.index != self._eng.device:
            raise InvalidArgument(f"mel_features lives on cuda:{mel.device.index}, the session runs on cuda:{self._eng.device}")
        if mel.shape[1] != self._cfg.num_mels or mel.shape[2] < 1:
            raise InvalidArgument(f"mel_features must be (B, {self._cfg.num_mels}, frames >= 1), got {tuple(mel.shape)}")
        out = self._eng.run_torch(mel.float().contiguous())
        return {"generated_wav": OrtValue._from_tensor(out)}

    def _chk(self, feed, name, dtype, ndim):
        a = np.asarray(feed[name])
        if a.dtype != dtype:
            raise InvalidArgument(f"Unexpected input data type. Actual: ({a.dtype}) , expected: ({np.dtype(dtype)}) for input {name}")
        if ndim is not None and a.ndim != ndim:
            raise InvalidArgument(f"Invalid rank for input: {name} Got: {a.ndim} Expected: {ndim}")
        return a

    def _rope_table(self, N: int, cos: bool) -> np.ndarray:
        if (N, cos) in self._rope_tabs:
            return self._rope_tabs[(N, cos)]
        if len(self._rope_tabs) > 8:
            self._rope_tabs.clear()
        self._rope_tabs[(N, cos)] = t = self._rope_table_build(N, cos)
        return t

    def _rope_table_build(self, N: int, cos: bool) -> np.ndarray:
        D = self._cfg.dim_head
        inv = (1.0 / (10000.0 ** (np.arange(0, D, 2, dtype=np.float32) / D))).astype(np.float32)
        ang = np.repeat(np.outer(np.arange(N, dtype=np.float32), inv), 2, axis=1)
        return (np.cos(ang) if cos else np.sin(ang)).astype(np.float16).astype(np.float32)

    def _run_gpt_e(self, feed):
        """graph E (IndexTTS/Export_IndexTTS.py:270-289).  Feed values are numpy arrays, OrtValues or the _KVRef
        objects a previous call returned."""
        e, c = self._eng, self._cfg
        val = lambda x: x.numpy() if isinstance(x, OrtValue) and not isinstance(x, _KVRef) else x
        hs = np.asarray(val(feed["hidden_state"]))
        if hs.dtype != np.float32 or hs.ndim != 3:
            raise InvalidArgument("hidden_state must be a rank-3 float tensor")
        ids_len = int(np.asarray(val(feed["ids_len"])).reshape(-1)[0])
        hist = int(np.asarray(val(feed["history_len"])).reshape(-1)[0])
        if ids_len != hs.shape[1]:
            raise InvalidArgument(f"ids_len ({ids_len}) does not match hidden_state rows ({hs.shape[1]})")
        flag = int(np.asarray(val(feed["attention_mask"])).reshape(-1)[0])
        pen = np.asarray(val(feed["repeat_penality"]), dtype=np.float32)
        k0 = feed["in_key_0"]
        if isinstance(k0, _KVRef):
            refs = [feed[f"in_key_{i}"] for i in range(c.layers)] + [feed[f"in_value_{i}"] for i in range(c.layers)]
            if not all(isinstance(r, _KVRef) and r._s is self and r.current() for r in refs):
                raise Fail("in_key/in_value references are stale (not the outputs of this session's latest call)")
            if hist != e.history_len:
                raise InvalidArgument(f"history_len ({hist}) does not match the cache ({e.history_len})")
        else:
            keys = [np.asarray(val(feed[f"in_key_{i}"])) for i in range(c.layers)]
            vals = [np.asarray(val(feed[f"in_value_{i}"])) for i in range(c.layers)]
            if keys[0].ndim != 3 or keys[0].shape[2] != hist:
                raise InvalidArgument(f"history_len ({hist}) does not match in_key_0 {keys[0].shape}")
            if hist == 0:
                e.reset()
            else:
                e.kv_write(keys, vals)
        kv, last, tok = e.step(hs, pen, attention_mask=flag)
        self._kv_epoch += 1
        out = {"kv_seq_len": kv, "last_hidden_state": last, "max_logit_id": tok}
        n = int(kv[0])
        for i in range(c.layers):
            out[f"out_key_{i}"] = _KVRef(self, i, False, n, self._kv_epoch)
            out[f"out_value_{i}"] = _KVRef(self, i, True, n, self._kv_epoch)
        return out

    def _run(self, feed) -> Dict[str, np.ndarray]:
        g, e = self._graph, self._eng
        if g == "IndexTTS_A":
            conds, lat = e.run(self._chk(feed, "audio", np.int16, 3))
            outs, cond0 = e.split_conds(conds)
            res = {f"save_bigvgan_conds_{i}": o for i, o in enumerate(outs)}
            res["bigvgan_cond_layer_speaker_embedding"] = cond0
            res["conds_latent"] = lat[None]
            return res
        if g == "IndexTTS_B":
            return {"text_hidden_state": e.text_embed(self._chk(feed, "text_ids", np.int32, 2))}
        if g == "IndexTTS_C":
            hs, nxt = e.mel_embed(self._chk(feed, "gpt_ids", np.int32, 2), self._chk(feed, "kv_seq_len", np.int64, 1))
            return {"gpt_hidden_state": hs, "next_kv_seq_len": nxt}
        if g == "IndexTTS_D":
            from .indextts import IndexGPT
            cat, n = IndexGPT.concat(*[self._chk(feed, k, np.float32, 3) for k in ("embed_x", "embed_y", "embed_z")])
            return {"concat_hidden_state": cat, "concat_len": n}
        if g == "BigVGAN":
            mel = np.asarray(feed["mel_features"])
            if mel.dtype not in (np.float32, np.float16) or mel.ndim != 3:
                raise InvalidArgument("mel_features must be a rank-3 float tensor")
            return {"generated_wav": e.run(mel.astype(np.float32))}
        if g == "IndexTTS_F":
            conds = [self._chk(feed, a.name, np.float32, 3) for a in self._inputs[:-1]]
            lat = self._chk(feed, "save_hidden_state", np.float32, 2)
            return {"generated_wav": e.run_latent(lat, conds)}
        h16 = bool(getattr(self._cfg, "ref_fp16_attn", False))       # the fp16-transformer export's float16 graph I/O
        fdt = np.float16 if h16 else np.float32
        if g == "F5_Preprocess":
            audio = self._chk(feed, "audio", np.int16, 3)
            ids = self._chk(feed, "text_ids", np.int32, 2)
            md = self._chk(feed, "max_duration", np.int64, 1)
            seed = _SEED[0] + _SEED_DRAWS[0]              # ORT's generator advances: repeated runs draw fresh noise
            _SEED_DRAWS[0] += 1
            o = e.preprocess(audio, ids, md, noise=None, seed=seed)
            if h16:
                o = {k: (np.ascontiguousarray(v).astype(np.float16) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v)
                     for k, v in o.items()}
            return o
        if g == "F5_Transformer":
            noise = self._chk(feed, "noise", fdt, 3).astype(np.float32, copy=False)
            cmt = self._chk(feed, "cat_mel_text", fdt, 3).astype(np.float32, copy=False)
            cmtd = self._chk(feed, "cat_mel_text_drop", fdt, 3).astype(np.float32, copy=False)
            ts = self._chk(feed, "time_step", np.int32, 1)
            # the engine keeps the RoPE tables on the device (the reference re-feeds 37 MB of them per call); a feed that is
            # not graph A's table (Export_F5.py:107-112: cos/sin of n * 10000^(-2j/64), rounded through fp16) is rejected
            N = noise.shape[1]
            for n in ("rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k"):
                a = np.asarray(feed[n])
                if a.ndim != 4:
                    raise InvalidArgument(f"Invalid rank for input: {n}")
                want = self._rope_table(N, "cos" in n)
                got = a[0, 0] if n.endswith("_q") else a[0, 0].T
                # the loop feeds the same table 31 times: a buffer recognised once is re-checked by a checksum over ALL of it (one
                # pass over 288 KB; buffer identity + the last row would miss an in-place edit elsewhere in the table: ADVICE r5)
                key = (n, a.ctypes.data, a.shape, a.strides, a.dtype.str, float(np.asarray(got, dtype=np.float32).sum(dtype=np.float64)),
                       float(np.abs(np.asarray(got, dtype=np.float32)).sum(dtype=np.float64)))
                if self._rope_seen.get(n) == key and got.shape == want.shape:
                    continue
                self._rope_seen[n] = key
                if got.shape != want.shape or np.abs(got.astype(np.float32) - want).max() > 2e-3:
                    raise InvalidArgument(f"{n}: not the RoPE table of F5_Preprocess (this engine regenerates the tables on "
                                          f"the device and cannot honour modified ones)")
            x, t = e.transformer_step(noise, cmt, cmtd, ts, fuse=max(1, int(getattr(self._cfg, "fuse_step", 1))))
            return {"denoised": x.astype(fdt) if h16 else x, "time_step": t}
        if g == "F5_Decode":
            den = self._chk(feed, "denoised", fdt, 3).astype(np.float32)
            rsl = int(np.asarray(feed["ref_signal_len"]))
            return {"output_audio": e.decode(den, rsl)}
        raise Fail(g)


# ---------------------------------------------------------------------------------------------------
# one-call convenience: lines :223-311 of F5-TTS-ONNX-Inference.py on the device
# ---------------------------------------------------------------------------------------------------
def synthesize(engine, ref_audio_i16: np.ndarray, ref_text: str, gen_text: str, vocab: Dict[str, int], *,
               speed: float = 1.0, seed: int = 9527, noise: Optional[np.ndarray] = None) -> np.ndarray:
    """reference audio (int16 mono 24 kHz) + texts in, int16 waveform (1,1,n) out."""
    from . import text as T
    audio = np.ascontiguousarray(np.asarray(ref_audio_i16).reshape(-1), dtype=np.int16)
    N = T.max_duration(audio.size, ref_text, gen_text, engine.cfg.hop_length, speed)
    ids = T.list_str_to_idx(T.convert_char_to_pinyin([ref_text + gen_text]), vocab)
    return engine.synthesize(audio[None], ids, N, noise=noise, seed=seed)
