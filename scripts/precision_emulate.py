#!/usr/bin/env python
"""CPU emulation of MFMA operand-split schemes (design study for the gfx950 kernels, not product code).

Restates the forward with every contraction routed through ``mm(left, right, family)`` so that the operand roundings
of a candidate scheme can be applied exactly (products and sums in float64: what is measured is the operand
representation, not the accumulation order).  Compared against the same forward with exact operands, on the golden
fixtures' inputs and weights (optionally rounded to bf16 first = what a bf16 checkpoint holds).

Schemes (left = activation-side operand, right = weight / key / value):
  exact          no rounding
  bf16           single pass, both operands RNE bf16
  bf16x2         left = hi + lo (bf16 pair), right bf16        (today's kernel set for bf16 checkpoints)
  bf16x3         both operands as bf16 pairs, lo x lo dropped  (today's all-terms set)
  f16            single pass fp16
  f16+f8         left = fp16 hi + e4m3 lo (x 2^S), right fp16 for the hi product and e4m3 for the lo product
  f16+2f8        as f16+f8 plus  e4m3(left) x e4m3(lo(right))  (fp32-valued weights)
  f16+alo16+f8w  f16+2f8 with lo(left) as an fp16 fragment; f16+f8a+wlo16: with lo(right) as one; f16x3: both

    python scripts/precision_emulate.py [--fixtures g0c_hd64_synth,g1_xsmall] [--bf16-weights]
"""

from __future__ import annotations

import argparse
import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from helpers import dims_from_meta, load_golden, state_from_fixture  # noqa: E402

F8 = torch.float8_e4m3fn
LO_SHIFT = 12  # lo planes are multiplied by 2^LO_SHIFT before the e4m3 conversion (hardware: MX block scale 2^-12)


def rne(x: torch.Tensor, dtype) -> torch.Tensor:
    return x.to(torch.float32).to(dtype).to(torch.float64)


def f8(x: torch.Tensor, shift: int = 0) -> torch.Tensor:
    s = 2.0**shift
    y = (x.to(torch.float32) * s).clamp(-448.0, 448.0).to(F8).to(torch.float64)
    return y / s


def split_pair(x: torch.Tensor, dtype):
    hi = rne(x, dtype)
    return hi, x.to(torch.float64) - hi


def make_mm(scheme: str, families: set[str] | None = None):
    def mm(left: torch.Tensor, right_t: torch.Tensor, family: str) -> torch.Tensor:
        """left [..., K] @ right_t [..., K, N] with the scheme's operand roundings."""
        a = left.to(torch.float64)
        b = right_t.to(torch.float64)
        sch = scheme if (families is None or family in families) else "exact"
        if sch == "exact":
            return a @ b
        if sch == "bf16":
            return rne(a, torch.bfloat16) @ rne(b, torch.bfloat16)
        if sch == "f16":
            return rne(a, torch.float16) @ rne(b, torch.float16)
        if sch == "bf16x2":
            ah, al = split_pair(a, torch.bfloat16)
            return (ah + rne(al, torch.bfloat16)) @ rne(b, torch.bfloat16)
        if sch == "bf16x3":
            ah, al = split_pair(a, torch.bfloat16)
            bh, bl = split_pair(b, torch.bfloat16)
            al, bl = rne(al, torch.bfloat16), rne(bl, torch.bfloat16)
            return ah @ bh + al @ bh + ah @ bl
        if sch.split(":")[0] in ("f16+f8", "f16+2f8", "f16+f8s"):
            # optional ":LS/WS/AS": shift of the lo planes, of the e4m3 weight plane, of the e4m3 activation plane (2^x)
            base, _, opts = sch.partition(":")
            ls, ws, as_ = (int(v) for v in opts.split("/")) if opts else (LO_SHIFT, 0, 0)
            ah, al = split_pair(a, torch.float16)
            bh, bl = split_pair(b, torch.float16)
            out = ah @ bh + f8(al, ls) @ f8(bh, ws)
            if base == "f16+2f8":
                out = out + f8(ah, as_) @ f8(bl, ls + ws)
            return out
        if sch in ("f16+alo16+f8w", "f16+f8a+wlo16", "f16x3"):
            # the lo part of ONE side (or both) as an fp16 fragment instead of e4m3: one more 16-bit MFMA for that term
            ah, al = split_pair(a, torch.float16)
            bh, bl = split_pair(b, torch.float16)
            left = rne(al, torch.float16) @ bh if sch != "f16+f8a+wlo16" else f8(al, LO_SHIFT) @ f8(bh, 0)
            right = ah @ rne(bl, torch.float16) if sch != "f16+alo16+f8w" else f8(ah, 0) @ f8(bl, LO_SHIFT)
            return ah @ bh + left + right
        if sch in ("f16+alo2f8", "f16+alo_exact_wf8", "f16+alof8_wexact"):
            # diagnostics of the lo(left) x e4m3(right) term: which rounding is its error -- lo(left)'s or right's?
            ah, al = split_pair(a, torch.float16)
            bh, bl = split_pair(b, torch.float16)
            if sch == "f16+alo2f8":      # lo(left) as TWO e4m3 planes (a second K = 128 product on the same e4m3(right))
                l1 = f8(al, LO_SHIFT)
                left = (l1 + f8(al - l1, LO_SHIFT + 4)) @ f8(bh, 0)
            elif sch == "f16+alo_exact_wf8":
                left = al @ f8(bh, 0)
            else:
                left = f8(al, LO_SHIFT) @ bh
            return ah @ bh + left + f8(ah, 0) @ f8(bl, LO_SHIFT)
        if sch == "bf16+f8":
            ah, al = split_pair(a, torch.bfloat16)
            return ah @ rne(b, torch.bfloat16) + f8(al, 9) @ f8(rne(b, torch.bfloat16))
        raise ValueError(sch)

    return mm


def layer_norm(x, w, eps):
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rope_tables(hd, theta, L):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().double(), emb.sin().double()


def rot(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def forward(state, dims, ids, mask, mm, resid_dtype=torch.float32):
    pre = "ranking_model." if any(k.startswith("ranking_model.") for k in state) else ""
    W = lambda n: state[pre + n].double()  # noqa: E731
    B, L = ids.shape
    H, nh = dims.hidden_size, dims.num_heads
    hd = H // nh
    eps = float(dims.norm_eps)
    ok = mask.bool()
    cg, sg = rope_tables(hd, dims.global_rope_theta, L)
    cl, sl = rope_tables(hd, dims.local_rope_theta, L)
    pos = torch.arange(L)
    dist = (pos[:, None] - pos[None, :]).abs()
    full = ok[:, None, None, :].expand(B, 1, L, L)
    local = full & (dist <= dims.half_window)[None, None]
    x = layer_norm(W("model.embeddings.tok_embeddings.weight")[ids], W("model.embeddings.norm.weight"), eps)
    scale = hd**-0.5
    for i in range(dims.num_layers):
        p = f"model.layers.{i}."
        glob = bool(dims.layer_is_global[i])
        h = x if i == 0 else layer_norm(x, W(p + "attn_norm.weight"), eps)
        qkv = mm(h, W(p + "attn.Wqkv.weight").T, "wqkv").view(B, L, 3, nh, hd)
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))
        c, s = (cg, sg) if glob else (cl, sl)
        q = (q * c + rot(q) * s) * scale
        k = k * c + rot(k) * s
        sc = mm(q, k.transpose(2, 3), "qk")
        sc = sc.masked_fill(~(full if glob else local), -1e300)
        pr = torch.softmax(sc, dim=-1)
        ctx = mm(pr, v, "pv").transpose(1, 2).reshape(B, L, H)
        x = x + mm(ctx, W(p + "attn.Wo.weight").T, "attn_out")
        x = x.to(resid_dtype).double()
        h = layer_norm(x, W(p + "mlp_norm.weight"), eps)
        a, g = mm(h, W(p + "mlp.Wi.weight").T, "wi").chunk(2, dim=-1)
        x = x + mm(gelu(a) * g, W(p + "mlp.Wo.weight").T, "mlp_out")
        x = x.to(resid_dtype).double()
    last = layer_norm(x, W("model.final_norm.weight"), eps)
    pooled = last[:, 0] if dims.classifier_pooling != "mean" else (last * mask[..., None]).sum(1) / mask.sum(1, keepdim=True)
    pooled = layer_norm(gelu(pooled @ W("head.dense.weight").T), W("head.norm.weight"), eps)
    rank = pooled @ W("classifier.weight").T + W("classifier.bias")
    prune = last @ state["pruning_head.classifier.weight"].double().T + state["pruning_head.classifier.bias"].double()
    return rank, prune


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixtures", default="g0c_hd64_synth,g1_xsmall,g7_xsmall_refinit")
    ap.add_argument("--bf16-weights", action="store_true")
    ap.add_argument("--schemes", default="bf16,f16,bf16x2,bf16x3,f16+f8,f16+2f8,bf16+f8")
    ap.add_argument("--families", default="", help="comma list: apply the scheme to these families only (others exact)")
    ap.add_argument("--gemm-only", action="store_true", help="attention (qk, pv) stays bf16x3 whatever the scheme")
    ap.add_argument("--override", default="", help="per-family scheme on top of --schemes, e.g. 'wqkv=f16+f8;wi=f16+f8' "
                    "(several alternatives separated by '|': each is run against every scheme)")
    args = ap.parse_args()
    torch.set_num_threads(8)
    for name in args.fixtures.split(","):
        arrays, meta = load_golden(name)
        dims = dims_from_meta(meta)
        state = state_from_fixture(arrays, meta)
        if args.bf16_weights:
            state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.dim() == 2 and "embeddings" not in k and "classifier" not in k and "head" not in k else v)
                     for k, v in state.items()}
        ids = torch.from_numpy(arrays["input_ids"]).long()
        mask = torch.from_numpy(arrays["attention_mask"]).long()
        m = mask.bool()
        ref_rank, ref_prune = forward(state, dims, ids, mask, make_mm("exact"), resid_dtype=torch.float64)
        for sch, ovr in ((a, b) for a in args.schemes.split(",") for b in args.override.split("|")):
            fams = set(args.families.split(",")) if args.families else None
            mm = make_mm(sch, fams)
            if ovr:
                table = {k: make_mm(v) for k, v in (item.split("=") for item in ovr.split(";"))}
                mm = lambda a, b, f, base=mm, table=table: table.get(f, base)(a, b, f)  # noqa: E731
                sch = f"{sch} [{ovr}]"
            if args.gemm_only:
                inner, att = mm, make_mm("bf16x3")
                mm = lambda a, b, f, inner=inner, att=att: (att if f in ("qk", "pv") else inner)(a, b, f)  # noqa: E731
            rank, prune = forward(state, dims, ids, mask, mm)
            ep = float((prune - ref_prune)[m].abs().max())
            er = float((rank - ref_rank).abs().max())
            kp = torch.sigmoid(prune[..., 1] - prune[..., 0])
            kr = torch.sigmoid(ref_prune[..., 1] - ref_prune[..., 0])
            ek = float((kp - kr)[m].abs().max())
            print(f"{name:20s} {'bf16w' if args.bf16_weights else 'fp32w'} {sch:10s} prune {ep:.2e} rank {er:.2e} keep {ek:.2e}", flush=True)


if __name__ == "__main__":
    main()
