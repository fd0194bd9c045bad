"""Checkpoint ingest: everything the reference's ``load_model`` accepts as a LOCAL checkpoint
(``whisperlivekit/whisper/__init__.py:466-596``) turned into what this backend uploads - ``(ModelDims, state dict in
the openai parameter names, alignment-head pairs or None)``:

* one file: ``.pt`` / ``.bin`` (``torch.load``) or ``.safetensors`` (``_load_checkpoint``, ``:394-425``);
* a directory: ``model.safetensors`` / ``pytorch_model.bin`` / ``*.pt`` / another ``*.safetensors`` (that priority), or a
  sharded checkpoint - through its ``*.index.json`` weight map or the ``name-00001-of-0000N.{safetensors,bin}`` pattern,
  ``adapter_*`` files never taken for the model (``model_paths.py:69-132``, ``_load_sharded_checkpoint`` ``:428-462``);
* layouts: openai (``{"dims", "model_state_dict"}`` or a bare state dict with openai names), HuggingFace
  ``WhisperForConditionalGeneration`` names (``_convert_hf_state_dict``, ``:163-252``), MLX names (``mlp1`` / ``mlp2``,
  ``_convert_mlx_state_dict``, ``:255-271``; its ``alignment_heads`` tensor is handed back as pairs, ``:587-592``);
* dimensions: the checkpoint's ``dims`` or the ``config.json`` beside it, native or HuggingFace keys
  (``_infer_dims_from_config``, ``:106-160``);
* a PEFT LoRA adapter directory (``adapter_config.json`` + ``adapter_model.safetensors`` / ``.bin``): every ``B A`` pair
  scaled by ``lora_alpha / r`` and added to the weight it addresses (``_apply_lora_adapter``, ``:337-391``).

Not here, on purpose: downloads (official model names, Hub repository ids - no network on a GPU box; the caller resolves a
local path first, ``model_paths.resolve_model_path``) and the Whisper ``nn.Module`` itself (the tensors go to
``engine.pack_state_dict``).  Pure host logic: ``torch`` only for ``torch.load`` and the LoRA matrix product.
"""
from __future__ import annotations

import json
import os
import re
from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple

from .dims import ModelDims

# ---- names ------------------------------------------------------------------------------------------------------------
_HF_ATTN = {"q_proj": "query", "k_proj": "key", "v_proj": "value", "out_proj": "out"}
_HF_BLOCK_LEAVES = {                               # HuggingFace sub-module of a layer -> openai sub-module of a block
    "self_attn_layer_norm": "attn_ln", "encoder_attn_layer_norm": "cross_attn_ln", "final_layer_norm": "mlp_ln",
    "fc1": "mlp.0", "fc2": "mlp.2",
}
_HF_TOP = {
    "encoder.embed_positions.weight": "encoder.positional_embedding",
    "decoder.embed_positions.weight": "decoder.positional_embedding",
    "encoder.layer_norm.weight": "encoder.ln_post.weight", "encoder.layer_norm.bias": "encoder.ln_post.bias",
    "decoder.layer_norm.weight": "decoder.ln.weight", "decoder.layer_norm.bias": "decoder.ln.bias",
}
_LAYER = re.compile(r"^(encoder|decoder)\.layers\.(\d+)\.(.+)$")


def hf_to_openai_name(key: str) -> Optional[str]:
    """``model.decoder.layers.3.encoder_attn.k_proj.weight`` -> ``decoder.blocks.3.cross_attn.key.weight``; None for a
    tensor the openai module tree has no place for (``proj_out.weight`` is the tied embedding: not under ``model.``)."""
    if not key.startswith("model."):
        return None
    sub = key[len("model."):]
    m = _LAYER.match(sub)
    if m:
        side, idx, rest = m.groups()
        head, _, tail = rest.partition(".")
        block = f"{side}.blocks.{idx}"
        if head in ("self_attn", "encoder_attn"):
            proj, _, leaf = tail.partition(".")
            if proj not in _HF_ATTN:
                return None
            attn = "attn" if head == "self_attn" else "cross_attn"
            return f"{block}.{attn}.{_HF_ATTN[proj]}" + (f".{leaf}" if leaf else "")
        if head in _HF_BLOCK_LEAVES:
            if head.endswith("layer_norm") and tail not in ("weight", "bias"):
                return None
            return f"{block}.{_HF_BLOCK_LEAVES[head]}.{tail}" if tail else None
        return None
    if sub.startswith("encoder.conv") or sub.startswith("decoder.conv"):
        return sub
    if sub in _HF_TOP:
        return _HF_TOP[sub]
    if sub.startswith("decoder.embed_tokens."):
        return sub.replace("embed_tokens", "token_embedding", 1)
    return None


def convert_hf_state_dict(sd: Mapping[str, Any]) -> Dict[str, Any]:
    """HuggingFace names -> openai names; a state dict without ``model.`` keys is returned as it is, and so is one of which
    nothing could be mapped (the reference's behaviour: the failure then surfaces when the tensors are consumed)."""
    if not any(k.startswith("model.") for k in sd):
        return dict(sd)
    out = {}
    for k, v in sd.items():
        name = hf_to_openai_name(k)
        if name:
            out[name] = v
    return out if out else dict(sd)


def convert_mlx_state_dict(sd: Mapping[str, Any]) -> Dict[str, Any]:
    if not any("mlp1" in k or "mlp2" in k for k in sd):
        return dict(sd)
    return {k.replace(".mlp1.", ".mlp.0.").replace(".mlp2.", ".mlp.2."): v for k, v in sd.items() if k != "alignment_heads"}


# ---- dimensions -------------------------------------------------------------------------------------------------------
_NATIVE = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab", "n_text_ctx", "n_text_state",
           "n_text_head", "n_text_layer")


def dims_from_config(cfg: Mapping[str, Any]) -> Optional[ModelDims]:
    if all(k in cfg for k in _NATIVE):
        return ModelDims(**{k: int(cfg[k]) for k in _NATIVE})
    try:
        return ModelDims(n_mels=cfg["num_mel_bins"], n_audio_ctx=cfg["max_source_positions"], n_audio_state=cfg["d_model"],
                         n_audio_head=cfg["encoder_attention_heads"],
                         n_audio_layer=cfg.get("encoder_layers") or cfg["num_hidden_layers"], n_vocab=cfg["vocab_size"],
                         n_text_ctx=cfg["max_target_positions"], n_text_state=cfg["d_model"],
                         n_text_head=cfg["decoder_attention_heads"], n_text_layer=cfg["decoder_layers"])
    except KeyError:
        return None


def infer_dims_from_config(path: str) -> Optional[ModelDims]:
    """``config.json`` inside the directory, or beside the file."""
    cfg = os.path.join(path if os.path.isdir(path) else os.path.dirname(path), "config.json")
    if not os.path.isfile(cfg):
        return None
    with open(cfg, "r", encoding="utf-8") as fh:
        return dims_from_config(json.load(fh))


# ---- files ------------------------------------------------------------------------------------------------------------
_SHARD = re.compile(r"^(.+)-(\d{5})-of-(\d{5})\.(safetensors|bin)$")
_INDEXES = ("model.safetensors.index.json", "pytorch_model.bin.index.json")


def collect_checkpoint_files(directory: str) -> List[str]:
    """The file(s) of the PyTorch checkpoint in ``directory``: the shards named by an index file, else a complete
    ``-0000i-of-0000N`` group, else the best single file (model.safetensors > pytorch_model.bin > *.pt > *.safetensors)."""
    for index in _INDEXES:
        p = os.path.join(directory, index)
        if os.path.isfile(p):
            try:
                with open(p, "r", encoding="utf-8") as fh:
                    names = sorted(set(json.load(fh).get("weight_map", {}).values()))
            except (ValueError, OSError):
                names = []
            shards = [os.path.join(directory, n) for n in names if os.path.isfile(os.path.join(directory, n))]
            if shards:
                return shards
    groups: Dict[Tuple[str, str, int], List[Tuple[int, str]]] = {}
    single: Dict[int, str] = {}
    for name in os.listdir(directory):
        full = os.path.join(directory, name)
        if not os.path.isfile(full) or name.startswith("adapter_"):
            continue
        m = _SHARD.match(name)
        if m:
            base, idx, total, ext = m.groups()
            groups.setdefault((base, ext, int(total)), []).append((int(idx), full))
            continue
        ext = os.path.splitext(name)[1].lower()
        rank = 0 if name == "model.safetensors" else 1 if name == "pytorch_model.bin" else 2 if ext == ".pt" else \
            3 if ext == ".safetensors" else None
        if rank is not None:
            single[rank] = full
    for (_, _, total), shards in groups.items():
        if len(shards) == total:
            return [p for _, p in sorted(shards)]
    return [single[min(single)]] if single else []


def load_tensor_file(path: str) -> Any:
    if path.lower().endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    import torch
    # tensors and plain containers only: a checkpoint file is data, not code (the reference's bare torch.load would also run
    # whatever a pickle asks for; an export that needs that has to be re-saved as tensors first)
    return torch.load(path, map_location="cpu", weights_only=True)


# ---- LoRA -------------------------------------------------------------------------------------------------------------
def _lora_target(module: str) -> str:
    """PEFT module path -> HuggingFace weight key (``base_model.model.model.decoder...`` -> ``model.decoder....weight``)."""
    if module.startswith("base_model."):
        module = module[len("base_model."):]
    if module.startswith("model.model."):
        module = module[len("model."):]
    if not module.startswith("model."):
        module = "model." + module
    return module + ".weight"


def apply_lora_adapter(sd: Dict[str, Any], lora_dir: str) -> int:
    """Merge a PEFT LoRA adapter into ``sd`` (openai names) in place; returns the number of weights touched."""
    import torch
    cfg_path = os.path.join(lora_dir, "adapter_config.json")
    if not os.path.isfile(cfg_path):
        raise FileNotFoundError(f"Missing adapter_config.json inside {lora_dir}")
    with open(cfg_path, "r", encoding="utf-8") as fh:
        cfg = json.load(fh)
    if cfg.get("peft_type") != "LORA":
        raise ValueError("Only LoRA adapters are supported.")
    r, alpha = cfg.get("r"), cfg.get("lora_alpha") or cfg.get("alpha")
    if not r or not alpha:
        raise ValueError("LoRA config must include `r` and `lora_alpha`.")
    safe, binf = os.path.join(lora_dir, "adapter_model.safetensors"), os.path.join(lora_dir, "adapter_model.bin")
    if os.path.isfile(safe):
        adapter = load_tensor_file(safe)
    elif os.path.isfile(binf):
        adapter = load_tensor_file(binf)
    else:
        raise FileNotFoundError(f"No adapter weights found under {lora_dir}. Expected adapter_model.safetensors or adapter_model.bin.")
    pairs: Dict[str, Dict[str, Any]] = {}
    for key, t in adapter.items():
        for tag in ("A", "B"):
            suffix = f".lora_{tag}.weight"
            if key.endswith(suffix):
                pairs.setdefault(key[:-len(suffix)], {})[tag] = t
    if not pairs:
        raise ValueError(f"No LoRA tensors found in {lora_dir}")
    for module, ab in pairs.items():
        if "A" not in ab or "B" not in ab:
            raise ValueError(f"Incomplete LoRA tensors for module '{module}'")
        target = hf_to_openai_name(_lora_target(module))
        if target is None:
            raise KeyError(f"Failed to map LoRA module '{module}' into Whisper state dict.")
        if target not in sd:
            raise KeyError(f"LoRA module '{module}' mapped to '{target}', but the base model has no such parameter.")
        base = torch.as_tensor(sd[target])
        delta = (ab["B"] @ ab["A"]) * (alpha / r)
        sd[target] = base + delta.to(dtype=base.dtype)
    return len(pairs)


# ---- the whole ingest -------------------------------------------------------------------------------------------------
def load_whisper_checkpoint(path: str, lora_path: Optional[str] = None) -> Tuple[ModelDims, Dict[str, Any], Optional[List[Tuple[int, int]]]]:
    """``load_model(path, lora_path=...)`` without the module: -> (dims, state dict in openai names, alignment-head pairs the
    checkpoint itself carries - MLX exports - or None)."""
    if os.path.isdir(path):
        files = collect_checkpoint_files(path)
        if not files:
            raise RuntimeError(f"No PyTorch checkpoint found in directory {path}. Expected .pt, .bin, or .safetensors file(s).")
    elif os.path.isfile(path):
        files = [path]
    else:
        raise RuntimeError(f"Model {path} not found (a local file or directory is expected here)")
    checkpoint: Any = {}
    if len(files) == 1:
        checkpoint = load_tensor_file(files[0])
    else:
        for f in files:
            part = load_tensor_file(f)
            if isinstance(part, dict):
                checkpoint.update(part)
    dims_cfg = checkpoint.get("dims") if isinstance(checkpoint, dict) else None
    sd = checkpoint["model_state_dict"] if isinstance(checkpoint, dict) and "model_state_dict" in checkpoint else checkpoint
    if not isinstance(sd, dict):
        raise ValueError(f"{path}: not a Whisper checkpoint (no state dict inside)")
    heads = None
    if "alignment_heads" in sd:
        h = sd["alignment_heads"]
        heads = [(int(a), int(b)) for a, b in (h.tolist() if hasattr(h, "tolist") else h)]
    sd = convert_mlx_state_dict(convert_hf_state_dict(sd))
    if lora_path:
        apply_lora_adapter(sd, lora_path)
    if dims_cfg is not None:
        dims = ModelDims(**{k: int(v) for k, v in dict(dims_cfg).items()})
    else:
        dims = infer_dims_from_config(path)
        if dims is None:
            raise RuntimeError("Could not determine model dimensions. Ensure the checkpoint includes 'dims' or a HuggingFace "
                               "config.json is present.")
    return dims, sd, heads
