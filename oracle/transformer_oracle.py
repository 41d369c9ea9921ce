"""TEST INFRASTRUCTURE (see oracle/criterion_oracle.c header): only tests/ may import this.

CPU restatement (torch float64, gradients by torch autograd) of the reference's Transformer block, arch token `TR`:

  block, parameters and their order   recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:41-95
  forward (post-LayerNorm, layer drop) TransformerCPC.cpp:153-182
  mlp (no dropout inside)              TransformerCPC.cpp:97-101
  selfAttention                        TransformerCPC.cpp:117-151 (q / sqrt(d); position table tiled over heads x batch)

fl::multiheadAttention and fl::relativePositionEmbeddingRotate are Flashlight functions, NOT vendored in /root/reference
(flashlight v0.3 fl/contrib/modules and fl/autograd/Functions.cpp): restated here from their published algorithm --
scores = q k^T + rotate(E q^T)[n .. n + T - 1]^T with n = (2 csz - 1) / 2, softmax over keys, times v; the rotate pads each
query's column of position scores with T zeros, re-reads the buffer with a row pitch one shorter, and so shifts column i
down by i rows.  PARITY UNPINNED for this function: the reference holds no golden vector or test for it; the restatement
is cross-checked against the closed form  rel[i][j] = q_i . E[j - i + csz - 1]  (zero outside the table) in
tests/test_oracle_nn.py, its position-free core is checked against torch's scaled_dot_product_attention (with and without a
key-padding mask), and the GPU path is held to this oracle.
"""
import math

import torch
import torch.nn.functional as F


def relative_position_rotate(ps):
    """ps [..., T, d0]: position scores of query i against table row r (the reference's (d0, T, .) array, d0 fastest)
    -> [..., T, d0 + T - 1] with out[i][r'] = ps[i][r' - i] (0 outside), by the reference's pad / re-pitch trick"""
    T, d0 = ps.shape[-2], ps.shape[-1]
    lead = ps.shape[:-2]
    padded = torch.cat([ps, ps.new_zeros(lead + (T, T))], dim=-1)          # join(0, data, zeros(T, T, .))
    flat = padded.reshape(lead + ((d0 + T) * T,))[..., :(T + d0 - 1) * T]  # moddims + rows(0, (T + d0 - 1) T - 1)
    return flat.reshape(lead + (T, d0 + T - 1))                            # moddims(d0 + T - 1, T, .)


def key_lengths(input_sizes, t_in, t_k):
    """valid keys per utterance as the reference builds its padding mask: forwardSequentialModuleWithPadMask
    (recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:58-81): n_b = ceil(size_b * T / max size), mask[t][b] = t < n_b
    on the T frames of the padded input; TransformerCPC.cpp:138-144 resizes it to the block's frames with af::resize
    ([UNVENDORED] ArrayFire; nearest neighbour: source index round(j * T / T_k), clamped to T - 1) and adds log(mask) to the
    scores.  float32 arithmetic as ArrayFire's."""
    import numpy as np
    sz = np.asarray(input_sizes, np.float32)
    nb = np.ceil(sz * np.float32(t_in) / sz.max())
    xf = np.float32(t_in) / np.float32(t_k)
    xs = (np.arange(t_k, dtype=np.float32) * xf).astype(np.float32)
    lo = np.floor(xs)
    src = np.minimum(lo + ((xs - lo) >= 0.5), t_in - 1)          # C round(): halves away from zero
    mask = src[None, :] < nb[:, None]
    assert all((m[:-1] >= m[1:]).all() for m in mask)                                # monotone: a prefix is valid
    return mask.sum(axis=1).astype(np.int32)


def attention(q, k, v, E, nheads, attn_mask=None, key_len=None):
    """q (already scaled), k, v [B][T][C]; E [2 csz - 1][d] or None -> [B][T][C]; attn_mask [B][heads][T][T]: the dropout
    multiplier of the attention probabilities (0 or 1 / (1 - p)), None in evaluation mode"""
    B, T, C = q.shape
    d = C // nheads
    split = lambda z: z.reshape(B, T, nheads, d).permute(0, 2, 1, 3)       # moddims(T, d, heads * B): head h = features h d ..
    qh, kh, vh = split(q), split(k), split(v)
    scores = qh @ kh.transpose(-1, -2)
    if E is not None:
        n = E.shape[0] // 2
        rot = relative_position_rotate(qh @ E.t())
        scores = scores + rot[..., n:n + T]
    if key_len is not None:    # log(padMask): -inf on the padded keys of each utterance
        pad = torch.arange(T)[None, :] >= torch.as_tensor(key_len)[:, None]
        scores = scores.masked_fill(pad[:, None, None, :], float("-inf"))
    attn = torch.softmax(scores, dim=-1)
    if attn_mask is not None:
        attn = attn * attn_mask
    return (attn @ vh).permute(0, 2, 1, 3).reshape(B, T, C)


def tr_block(x, params, nheads, csz, attn_mask=None, f=1.0, key_len=None):
    """x [B][T][C]; params in the reference's params() order and memory layouts: position table [d][2 csz - 1] (ArrayFire
    (2 csz - 1, d), column-major; absent when csz == 0), then w1, w2, wq, wk, wv, wf as (W [in][out], b [out]) pairs,
    then the (gamma, beta) pairs of norm1 and norm2.  f: the layer-drop factor of this step (0 = block dropped)"""
    B, T, C = x.shape
    d = C // nheads
    i = 0
    E = None
    if csz > 0:
        E = params[0].t()
        i = 1
    w1, b1, w2, b2, wq, bq, wk, bk, wv, bv, wf, bf, g1, g2 = params[i:i + 14]
    lin = lambda z, w, b: z @ w + b
    ln = lambda z, gb: F.layer_norm(z, (C,), eps=1e-5) * gb[0] + gb[1]
    q = lin(x, wq, bq) / math.sqrt(d)
    o = lin(attention(q, lin(x, wk, bk), lin(x, wv, bv), E, nheads, attn_mask, key_len), wf, bf)
    h = ln(f * o + x, g1)
    return ln(f * lin(torch.relu(lin(h, w1, b1)), w2, b2) + h, g2)


def tr_param_shapes(C, mlp, nheads, csz):
    d = C // nheads
    out = [("tr.pos", (d, 2 * csz - 1))] if csz > 0 else []
    for a, b in ((C, mlp), (mlp, C), (C, C), (C, C), (C, C), (C, C)):
        out += [("linear.w", (a, b)), ("linear.b", (b,))]
    return out + [("ln", (2,)), ("ln", (2,))]
