"""Power-of-two operand scales of the 2 x fp16 / 3-product split (csrc/split.h: SplitH2), derived from the WEIGHTS alone.

fp16 has five exponent bits, so every operand of a SplitH2 product is multiplied by an exact 2^e before it is split, with
the contract ``|x| * 2^e <= 2^15`` for every value the operand can take (fp16 overflows at 65 504; the factor two is the
margin for fp32 rounding in the bounds below).  Nothing here looks at activations: the bounds are theorems about the
network (models/transformer.py:74-90), so no input can overflow an operand and there is no run-time flag.

  * a LayerNorm output  y = gamma * n + beta  has  sum n_j^2 = d var / (var + eps) <= d = 256, hence |n_j| <= 16:
        |y_j| <= 16 |gamma_j| + |beta_j|                                        (block inputs x, the tail's m1)
  * a bias-free Linear of a LayerNorm output, row w:  w . y = (w * gamma) . n + w . beta, and |n|_2 <= 16:
        |w . y| <= 16 |w * gamma|_2 + |w . beta|                                 (q, k, v; the FFN's hidden units)
  * the attention output of a row is a convex combination of value rows (the weights Q'.K' are non-negative, the 1e-6 in
    the denominator only shrinks it; models/transformer.py:38-42):  |att_j| <= max |v_j| <= the bound on v;
  * relu and elu + 1 do not increase a bound by more than 1.

The bound fixes the exponent: e = floor(log2(2^15 / bound)).  On the HIGH side it is clamped at E_MAX (a smaller exponent only
gives headroom away: safe); on the LOW side it is NOT clamped -- a bound above 2^(15 - E_MIN) = 2^39 has no exponent in range
that keeps the contract, so exp_for raises ScaleRangeError and the model routes such weights to the scale-free bf16 x 3 split
(gemm_backend "x3"; scream_amd/model.py).  Values below 2^-3 / 2^e lose relative precision (the second fp16 plane goes
subnormal), i.e. operand values more than 2^18 below their bound -- 2^-40 of the operand's range in absolute terms.  Weight
matrices take e from their largest |element|.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

TOP = 2.0 ** 15
E_MIN, E_MAX = -24, 24  # keeps 2^(e_a + e_w) squared, times a variance, far inside fp32 (csrc/tail_split.hip: tail_scales)


class ScaleRangeError(ValueError):
    """No power-of-two scale inside the kernels' checked range keeps |x| 2^e <= 2^15 for this operand: the fp16 x 2 split cannot
    carry these weights.  PointTransformer falls back to gemm_backend 'x3' (bf16 x 3, scale free) when it sees this."""


def exp_for(bound: float) -> int:
    """Largest e <= E_MAX with bound * 2^e <= 2^15; bound = 0 -> E_MAX.  Raises ScaleRangeError for a non-finite bound or one
    that would need e < E_MIN (bound > 2^39): clamping there would silently break the contract (fp16 inf / NaN in the products)."""
    b = float(bound)
    if not math.isfinite(b):
        raise ScaleRangeError("non-finite weights: no fp16 operand scale exists (use gemm_backend='x3')")
    if b <= 0.0:
        return E_MAX
    e = math.floor(math.log2(TOP / b))
    while b * 2.0 ** e > TOP:  # (log2 rounded up across a power of two)
        e -= 1
    if e < E_MIN:
        raise ScaleRangeError("operand bound %.3g needs the exponent %d < %d: outside the range the fp16 x 2 kernels were checked "
                              "for (use gemm_backend='x3')" % (b, e, E_MIN))
    return min(E_MAX, e)


def w_exp(W: torch.Tensor) -> int:
    return exp_for(W.detach().double().abs().max().item())


def ln_bound(gamma: torch.Tensor, beta: torch.Tensor) -> float:
    """max_j |LayerNorm(.)_j|."""
    return (16.0 * gamma.detach().double().abs() + beta.detach().double().abs()).max().item()


def lin_bound(W: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor] = None) -> float:
    """max_row |W[row] . LayerNorm(.)| (+ |bias|)."""
    W, g, b = W.detach().double(), gamma.detach().double(), beta.detach().double()
    out = 16.0 * (W * g).norm(dim=1) + (W @ b).abs()
    if bias is not None:
        out = out + bias.detach().double().abs()
    return out.max().item()


def tail_exps(Wm, W1, W2, g1, b1, v_bound: float, q_bound: float) -> Dict[str, int]:
    """Exponents of the layer tail (scream_tail_exps_t) from its weights, a bound on the value rows the attention mixes and a bound
    on the query projection q (Q' = elu(q) + 1 <= 1 + max(q, 0) is an fp16 x 2 operand of the attention apply since round 4)."""
    e_att = exp_for(v_bound)
    e_q = exp_for(1.0 + q_bound)
    e_m1 = exp_for(ln_bound(g1, b1))
    e_h = exp_for(lin_bound(W1, g1, b1))
    e_wm, e_w1, e_w2 = w_exp(Wm), w_exp(W1), w_exp(W2)
    # the accumulator units 2^(e_w + e_a) must stay within what the kernel's LayerNorm arithmetic was checked for
    # (csrc/tail_split.hip: tail_scales accepts |e_w + e_a| <= 44).  Too HIGH: give headroom away (safe).  Too LOW: no remedy
    # inside the fp16 split -- raise here, with a message, instead of SCREAM_EINVAL from scream_pack_tail
    e_att = min(e_att, 40 - e_wm)
    e_h = min(e_h, 40 - e_w2)
    for name, e in (("merge", e_wm + e_att), ("FFN-down", e_w2 + e_h)):
        if e < -44:
            raise ScaleRangeError("the %s accumulators would be in units of 2^%d: outside the range the layer-tail kernel's LayerNorm "
                                  "arithmetic was checked for (use gemm_backend='x3')" % (name, e))
    return {"e_att": e_att, "e_wm": e_wm, "e_m1": e_m1, "e_w1": e_w1, "e_h": e_h, "e_w2": e_w2, "e_q": e_q}


def layer_exps(m, in_q, in_kv) -> Dict[str, int]:
    """Exponents of one MHAttention block.  m: parameter holder (q_proj, k_proj, v_proj, merge, mlp, norm1, norm2);
    in_q / in_kv: (gamma, beta) of the LayerNorm that produced the query-side / key-value-side input."""
    ex = tail_exps(m.merge.weight, m.mlp[0].weight, m.mlp[2].weight, m.norm1.weight, m.norm1.bias, lin_bound(m.v_proj.weight, *in_kv),
                   lin_bound(m.q_proj.weight, *in_q))
    ex.update(e_xq=exp_for(ln_bound(*in_q)), e_xkv=exp_for(ln_bound(*in_kv)),
              # the projection's K^T V epilogue: K' = elu(k) + 1 <= 1 + max(k, 0), V = v
              e_k=exp_for(1.0 + lin_bound(m.k_proj.weight, *in_kv)), e_v=exp_for(lin_bound(m.v_proj.weight, *in_kv)))
    return ex
