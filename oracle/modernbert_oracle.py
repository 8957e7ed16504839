"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under ``open_provence_amd/`` may import this file.

A plain torch-CPU restatement of the forward that the reference delegates to third-party code:

* ``OpenProvenceModel.forward``            -> /root/reference/open_provence/modeling_open_provence_standalone.py:1666-1739
* ``OpenProvenceHead.forward``             -> same file :434-448  (Linear(H, 2) on every token)
* HF ``ModernBertForSequenceClassification`` (transformers 5.15.0 in this image; the reference pins
  4.57.1 in uv.lock:3640-3641 -- the source of that version is NOT under /root/reference, so the
  published algorithm is restated here and anchored on the reference's own call sites
  standalone.py:1341,1686-1695):
    embeddings + LayerNorm           modeling_modernbert.py:52-71
    GeGLU MLP                        :74-91
    RoPE tables / rotate_half        :94-219
    attention (eager form)           :166-185, :222-301
    encoder layer wiring             :304-333   (attn_norm is Identity on layer 0)
    model loop + final_norm          :409-478
    prediction head + classifier     :481-490, :569-622
    sliding-window mask |q-k| <= local_attention//2 intersected with padding:
                                     masking_utils.py:141-158, configuration_modernbert.py:159-162

Parity pin: the reference's own tests hold NO golden vectors for encoder numerics, so
this restatement is pinned against outputs of the reference itself run in the build container:
``tests/golden/make_golden.py`` imports the real ``OpenProvenceModel`` (+ HF ModernBERT) and stores its
inputs/outputs; ``tests/test_oracle_golden.py`` checks this file against them to <= 2e-5.

The arithmetic is written with explicit matmul / softmax so every line can be compared with a kernel;
``attn="sdpa"`` switches the attention core to ``scaled_dot_product_attention`` (what the reference
executes on a CPU device) for the timed CPU baseline in bench.py.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Mapping

import torch
import torch.nn.functional as F


@dataclass
class OracleOutput:
    ranking_logits: torch.Tensor  # [B, num_labels]
    pruning_logits: torch.Tensor  # [B, L, 2]
    hidden_states: list[torch.Tensor] | None  # N+1 tensors; the last one is post-final_norm (HF ties it)


def _layer_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight


def _gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rope_tables(head_dim: int, theta: float, positions: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """cos/sin of shape [L, head_dim], fp32, ``emb = cat(freqs, freqs)`` (modeling_modernbert.py:117-163)."""

    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = positions.to(torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def _key(prefix: str, name: str) -> str:
    return f"{prefix}{name}"


def oracle_forward(
    state: Mapping[str, torch.Tensor],
    dims,
    input_ids: torch.Tensor,
    attention_mask: torch.Tensor | None = None,
    *,
    dtype: torch.dtype = torch.float32,
    attn: str = "eager",
    return_hidden: bool = False,
    prune_pre_final_norm: bool = False,
) -> OracleOutput:
    """Padded-batch forward: ``input_ids[B, L]`` int64, ``attention_mask[B, L]`` (1 = real token).

    ``dims`` is an ``open_provence_amd.config.EncoderDims`` (duck-typed: only attributes are read).
    ``dtype=torch.float64`` gives a higher-precision run used to measure how far the fp32 oracle itself
    is from exact arithmetic.
    """

    pre = "ranking_model." if any(k.startswith("ranking_model.") for k in state) else ""

    def W(name: str) -> torch.Tensor:
        return state[_key(pre, name)].to(dtype)

    B, L = input_ids.shape
    H, nh = dims.hidden_size, dims.num_heads
    hd = H // nh
    eps = float(dims.norm_eps)
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    key_ok = attention_mask.to(torch.bool)  # [B, L]

    positions = torch.arange(L)
    cos_g, sin_g = rope_tables(hd, dims.global_rope_theta, positions)
    cos_l, sin_l = rope_tables(hd, dims.local_rope_theta, positions)
    cos_g, sin_g, cos_l, sin_l = (t.to(dtype) for t in (cos_g, sin_g, cos_l, sin_l))

    dist = (positions[:, None] - positions[None, :]).abs()
    full_mask = key_ok[:, None, None, :].expand(B, 1, L, L)
    local_mask = full_mask & (dist <= dims.half_window)[None, None, :, :]
    neg = torch.finfo(dtype).min

    x = _layer_norm(W("model.embeddings.tok_embeddings.weight")[input_ids], W("model.embeddings.norm.weight"), eps)
    hidden = [x] if return_hidden else None
    scale = hd**-0.5

    for i in range(dims.num_layers):
        p = f"model.layers.{i}."
        is_global = bool(dims.layer_is_global[i])
        h = x if i == 0 else _layer_norm(x, W(p + "attn_norm.weight"), eps)
        qkv = h @ W(p + "attn.Wqkv.weight").T  # [B, L, 3H], rows ordered [q | k | v] (view(...,3,nh,hd))
        qkv = qkv.view(B, L, 3, nh, hd)
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))  # [B, nh, L, hd]
        cos, sin = (cos_g, sin_g) if is_global else (cos_l, sin_l)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        mask = full_mask if is_global else local_mask
        if attn == "sdpa":
            ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=scale)
        else:
            scores = (q @ k.transpose(2, 3)) * scale
            scores = scores.masked_fill(~mask, neg)
            probs = torch.softmax(scores, dim=-1)
            ctx = probs @ v
        ctx = ctx.transpose(1, 2).reshape(B, L, H)
        x = x + ctx @ W(p + "attn.Wo.weight").T
        h = _layer_norm(x, W(p + "mlp_norm.weight"), eps)
        a, g = (h @ W(p + "mlp.Wi.weight").T).chunk(2, dim=-1)  # [input half ; gate half]
        x = x + (_gelu_erf(a) * g) @ W(p + "mlp.Wo.weight").T
        if return_hidden and i != dims.num_layers - 1:
            hidden.append(x)

    last = _layer_norm(x, W("model.final_norm.weight"), eps)
    if return_hidden:
        hidden.append(last)  # HF ties hidden_states[-1] to the post-final_norm output

    if dims.classifier_pooling == "mean":
        m = attention_mask.to(dtype)
        pooled = (last * m[..., None]).sum(dim=1) / m.sum(dim=1, keepdim=True)
    else:
        pooled = last[:, 0]
    pooled = _layer_norm(_gelu_erf(pooled @ W("head.dense.weight").T), W("head.norm.weight"), eps)
    ranking_logits = pooled @ W("classifier.weight").T + W("classifier.bias")

    pw = state["pruning_head.classifier.weight"].to(dtype)
    pb = state["pruning_head.classifier.bias"].to(dtype)
    # standalone.py:1695 feeds outputs.hidden_states[-1] to the head: the final_norm output under transformers >= 5
    # (utils/output_capturing.py:269-277 ties it to last_hidden_state), the last layer's un-normalised output under
    # the 4.x line the reference pins (its ModernBertModel.forward appended the tuple entry before final_norm).
    pruning_logits = (x if prune_pre_final_norm else last) @ pw.T + pb
    return OracleOutput(ranking_logits=ranking_logits, pruning_logits=pruning_logits, hidden_states=hidden)


def keep_probabilities(pruning_logits: torch.Tensor) -> torch.Tensor:
    """softmax(pruning_logits)[..., 1] in fp32, as the reference post-processing does (standalone.py:2918-2920)."""

    return torch.softmax(pruning_logits.to(torch.float32), dim=-1)[..., 1]


def ranking_scores(ranking_logits: torch.Tensor) -> torch.Tensor:
    """sigmoid of the first label (standalone.py:2913-2916)."""

    logits = ranking_logits.to(torch.float32)
    return torch.sigmoid(logits[..., 0])
