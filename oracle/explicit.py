"""The LSTM and encoder-layer arithmetic of the hot path spelled out step by step
(TEST INFRASTRUCTURE ONLY).

`oracle/models.py` delegates this arithmetic to `nn.LSTM` / `nn.TransformerEncoderLayer`
exactly as the reference does (models/AttnCut.py:8-10).  The functions here restate what
those modules compute, with plain tensor algebra, so that
  * the semantics the HIP kernels must reproduce are written down in one place
    (gate order i,f,g,o; h0=c0=0; [forward | backward] concatenation; attention over the
    LIST axis at each position; post-norm residual blocks; LN eps 1e-5), and
  * tests can run them in float64 to bound the fp32 error of both torch-CPU and the HIP path.
They are checked against the stock modules in tests/test_oracle_golden.py.
"""
import math

import torch


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one LSTM layer.  x (B,S,I) -> h (B,S,H).

    gates = x_t W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh, split in the order i, f, g, o;
    c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g);  h_t = sigmoid(o) tanh(c_t).
    """
    n_list, n_pos, _ = x.shape
    hid = w_hh.shape[1]
    h = x.new_zeros(n_list, hid)
    c = x.new_zeros(n_list, hid)
    out = x.new_zeros(n_list, n_pos, hid)
    steps = range(n_pos - 1, -1, -1) if reverse else range(n_pos)
    for t in steps:
        gates = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        gi, gf, gg, go = gates.split(hid, dim=1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        out[:, t] = h
    return out


def bilstm(x, sd, prefix, num_layers=2):
    """Stacked bidirectional LSTM from an `nn.LSTM` state_dict (keys `<prefix>weight_ih_l0[_reverse]`...)."""
    h = x
    for layer in range(num_layers):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            p = [sd[f"{prefix}{n}_l{layer}{suffix}"].to(x.dtype)
                 for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            outs.append(lstm_direction(h, *p, reverse=rev))
        h = torch.cat(outs, dim=2)
    return h


def layer_norm(x, weight, bias, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)        # biased variance
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def list_axis_attention(x, in_w, in_b, out_w, out_b, n_head, prob_mask=None):
    """Multi-head self-attention over axis 0 of x (B,S,E): at each position s and head h the
    B lists attend to each other; scores scaled by 1/sqrt(E/n_head).  `prob_mask` (S,head,B,B): train-mode dropout
    of the attention probabilities as data - keep ? 1/(1-p) : 0, multiplied onto softmax(score) like F.dropout does."""
    n_list, n_pos, emb = x.shape
    hd = emb // n_head
    qkv = x @ in_w.t() + in_b                                # (B,S,3E)
    q, k, v = qkv.split(emb, dim=2)
    # -> (S, head, B, hd)
    shape = lambda t: t.reshape(n_list, n_pos, n_head, hd).permute(1, 2, 0, 3)
    q, k, v = shape(q), shape(k), shape(v)
    score = (q @ k.transpose(-1, -2)) / math.sqrt(hd)        # (S,head,B,B)
    prob = torch.softmax(score, dim=-1)
    if prob_mask is not None:
        prob = prob * prob_mask
    ctx = prob @ v                                           # (S,head,B,hd)
    ctx = ctx.permute(2, 0, 1, 3).reshape(n_list, n_pos, emb)
    return ctx @ out_w.t() + out_b


def encoder_layer(x, sd, prefix, n_head, relu_gate=None, masks=None):
    """Post-norm encoder layer from a `TransformerEncoderLayer` state_dict; dropout 0 unless `masks` is given.

    `masks` (tests only): the four train-mode dropouts of nn.TransformerEncoderLayer with their keep-masks as DATA
    (keep ? 1/(1-p) : 0): "attn" (S,head,B,B) on the attention probabilities, "res1" (B,S,E) = dropout1 on the attention
    branch, "ffn" (B,S,FF) on the ReLU output, "res2" (B,S,E) = dropout2 on the FFN branch.  CPU-torch's RNG stream cannot
    be reproduced on the device, so train-mode parity is defined on identical masks.

    `relu_gate(z) -> 0/1 tensor` (tests only) replaces the ReLU's own branch decision `z > 0`: ReLU is discontinuous in
    its derivative, so a unit whose pre-activation lies within rounding distance of zero may legitimately take either
    branch in two implementations; a test that wants to bound everything ELSE passes the other implementation's
    decisions for exactly those units (and counts them)."""
    g = lambda name: sd[prefix + name].to(x.dtype)
    m = masks or {}
    att = list_axis_attention(x, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"),
                              g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), n_head, m.get("attn"))
    if "res1" in m:
        att = att * m["res1"]
    x = layer_norm(x + att, g("norm1.weight"), g("norm1.bias"))
    z = x @ g("linear1.weight").t() + g("linear1.bias")
    hid = torch.relu(z) if relu_gate is None else z * relu_gate(z.detach())
    if "ffn" in m:
        hid = hid * m["ffn"]
    ff = hid @ g("linear2.weight").t() + g("linear2.bias")
    if "res2" in m:
        ff = ff * m["res2"]
    return layer_norm(x + ff, g("norm2.weight"), g("norm2.bias"))


def attncut_forward(x, sd, n_head=4, num_layers=1):
    """AttnCut (models/AttnCut.py:16-20) from its state_dict, any float dtype."""
    h = bilstm(x, sd, "encoding_layer.")
    for i in range(num_layers):
        h = encoder_layer(h, sd, f"attention_layer.layers.{i}.", n_head)
    logit = h @ sd["decison_layer.0.weight"].to(x.dtype).t() + sd["decison_layer.0.bias"].to(x.dtype)
    return torch.softmax(logit, dim=1)


def choopy_forward(x, sd, n_head=8, num_layers=3):
    """Choopy (models/Choopy.py:18-23) from its state_dict."""
    pe = sd["position_encoding"].to(x.dtype)
    h = torch.cat((x, pe.expand(x.shape[0], *pe.shape)), dim=2)
    for i in range(num_layers):
        h = encoder_layer(h, sd, f"attention_layer.layers.{i}.", n_head)
    logit = h @ sd["decison_layer.0.weight"].to(x.dtype).t() + sd["decison_layer.0.bias"].to(x.dtype)
    return torch.softmax(logit, dim=1)
