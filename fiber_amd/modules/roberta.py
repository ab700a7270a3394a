"""RoBERTa text backbone with text->image cross-attention, MI355X-native.

Mirrors the module tree / state-dict keys / call signatures of the reference's coarse_grained/fiber/modules/roberta.py
(RobertaEmbeddings :142, RobertaSelfAttention :218, RobertaSelfOutput :330 -- NO residual / LayerNorm there,
RobertaAttention :344, RobertaIntermediate :394, RobertaOutput :410, RobertaLayer :427, RobertaEncoder :506,
RobertaModel :708).  Arithmetic runs on the HIP kernels through fiber_amd.ops; HF's PreTrainedModel machinery
(`from_pretrained`, pruning, caches, decoder paths) is outside the fused path and not reproduced.
"""
import math
import types

import torch
import torch.nn as nn

from .. import ops

NUM_FUSE_BLOCK = 6      # module globals set by FIBERTransformerSS.__init__ (fiber_module.py:46-47)
DIM_IMG = 1024


def roberta_base_config(**over):
    cfg = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1,
               hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    cfg.update(over)
    return types.SimpleNamespace(**cfg)


class RobertaEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.padding_idx = config.pad_token_id
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size, padding_idx=self.padding_idx)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))

    def forward(self, input_ids=None):
        y = ops.roberta_embed(input_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                              self.token_type_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias,
                              pad=self.padding_idx, eps=self.LayerNorm.eps, p_drop=self.dropout.p, training=self.training)
        return ops.start_stream(y)                          # the text residual stream begins (fp32 payload in fp32 mode)


class RobertaSelfAttention(nn.Module):
    def __init__(self, config, layer_index=None):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        assert self.attention_head_size in (32, 64), "MHA kernel covers head_dim 32/64"
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        if layer_index is None:
            kv = config.hidden_size
        else:
            kv = int(DIM_IMG / 2) if layer_index < 10 else DIM_IMG          # roberta.py:236-241
        self.key = nn.Linear(kv, self.all_head_size)
        self.value = nn.Linear(kv, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, kv_alias_sink=None):
        B, S, _ = hidden_states.shape
        C, scale = self.all_head_size, 1.0 / math.sqrt(self.attention_head_size)
        p = self.dropout.p if self.training else 0.0
        seed = ops.next_seed() if p > 0 else 0
        if encoder_hidden_states is None:
            # q, k, v of one input as ONE GEMM (three nn.Linear modules in the state dict, roberta.py:231-241)
            qkv = ops.linear_packed(hidden_states, [(self.query.weight, self.query.bias), (self.key.weight, self.key.bias),
                                                    (self.value.weight, self.value.bias)])
            o = ops.mha_qkv_packed(qkv.view(B * S, 3 * C), attention_mask, B, self.num_attention_heads, scale, p, seed)
        else:
            # t2i: keys / values from the image tokens as one GEMM; roberta.py:276: the cross-attention mask is None
            q = ops.linear(hidden_states, self.query.weight, self.query.bias).view(B * S, C)
            Lk = encoder_hidden_states.shape[1]
            # (kv_alias_sink: the caller wants an alias of the image tokens for their other consumer, see ops._LinearPacked)
            kv = ops.linear_packed(encoder_hidden_states, [(self.key.weight, self.key.bias), (self.value.weight, self.value.bias)],
                                   fork_sink=kv_alias_sink)
            o = ops.mha_kv_packed(q, kv.view(B * Lk, 2 * C), None, B, self.num_attention_heads, scale, p, seed)
        return o.view(B, S, C)


class RobertaSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)   # in the state dict; applied by RobertaLayer
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, residual=None):
        if residual is not None and not (self.training and self.dropout.p > 0):
            return ops.linear(hidden_states, self.dense.weight, self.dense.bias, residual=residual)
        y = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        if residual is None:
            return ops.dropout(y, self.dropout.p, self.training)
        return ops.stream_add(residual, y, p_a=self.dropout.p, training=self.training)     # dropout + residual: one pass


class RobertaAttention(nn.Module):
    def __init__(self, config, layer_index=None):
        super().__init__()
        self.self = RobertaSelfAttention(config, layer_index=layer_index)
        self.output = RobertaSelfOutput(config)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, residual=None):
        return self.output(self.self(hidden_states, attention_mask, encoder_hidden_states), residual)


class RobertaIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        return ops.linear(hidden_states, self.dense.weight, self.dense.bias, act="gelu")


class RobertaOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor, last_norm=True):
        if self.training and self.dropout.p > 0:
            h = ops.stream_add(input_tensor, ops.linear(hidden_states, self.dense.weight, self.dense.bias), p_a=self.dropout.p)
        else:
            h = ops.linear(hidden_states, self.dense.weight, self.dense.bias, residual=input_tensor)
        return self._norm(h, last_norm)

    def _norm(self, h, last_norm):
        if not last_norm:
            return h
        # post-LN: this output is the next layer's residual -> it keeps an fp32 copy in fp32-stream mode
        return ops.layernorm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, want_f32=ops.residual_fp32())

    def ffn(self, intermediate, input_tensor, last_norm=True):
        """intermediate + output of one layer as ONE autograd node for fc1 -> GELU -> fc2 (ops.mlp): the backward is the fused
        (dY.W2^T) * gelu'(H) GEMM instead of a dX GEMM + a gelu' pass over [B*S, 3072] (roberta.py:398-423; same forward kernels)."""
        wi, bi = intermediate.dense.weight, intermediate.dense.bias
        if self.training and self.dropout.p > 0:
            h = ops.stream_add(input_tensor, ops.mlp(input_tensor, wi, bi, self.dense.weight, self.dense.bias), p_a=self.dropout.p)
        else:
            h = ops.mlp(input_tensor, wi, bi, self.dense.weight, self.dense.bias, residual=input_tensor)
        return self._norm(h, last_norm)


class RobertaLayer(nn.Module):
    def __init__(self, config, layer_index=None):
        super().__init__()
        self.attention = RobertaAttention(config)
        if layer_index >= config.num_hidden_layers - NUM_FUSE_BLOCK:       # roberta.py:435 (12 - NUM_FUSE_BLOCK)
            self.crossattention_t2i = RobertaAttention(config, layer_index=layer_index)
        self.intermediate = RobertaIntermediate(config)
        self.output = RobertaOutput(config)
        self.alpha_t2i = nn.Parameter(torch.Tensor([0]))

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, last_norm=True, image_alias_sink=None):
        ln = self.attention.output.LayerNorm
        if encoder_hidden_states is None:
            a = self.attention(hidden_states, attention_mask, residual=hidden_states)      # dense(attn) + h fused
        else:
            assert hasattr(self, "crossattention_t2i"), "layer built without cross-attention"
            a = self.attention(hidden_states, attention_mask)
            ca = self.crossattention_t2i
            c = ca.self(a, None, encoder_hidden_states, kv_alias_sink=image_alias_sink)
            c = ops.linear(c, ca.output.dense.weight, ca.output.dense.bias)
            # hidden + (a + alpha_t2i * dropout(c)) in one pass (roberta.py:474-485: RobertaSelfOutput's dropout, the gate, the residual)
            a = ops.stream_add(hidden_states, a, b=c, alpha=self.alpha_t2i, p_b=ca.output.dropout.p, training=self.training)
        a = ops.layernorm(a, ln.weight, ln.bias, ln.eps, want_f32=ops.residual_fp32())
        return (self.output.ffn(self.intermediate, a, last_norm=last_norm),)


class RobertaEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([RobertaLayer(config, layer_index=i) for i in range(config.num_hidden_layers)])


class RobertaPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)    # present in checkpoints, unused on the fused path


class RobertaModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = RobertaEmbeddings(config)
        self.encoder = RobertaEncoder(config)
        self.pooler = RobertaPooler(config)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):                                   # HF PreTrainedModel._init_weights, N(0, 0.02)
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(mean=0.0, std=0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.LayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)

    @classmethod
    def from_pretrained(cls, name, **over):
        """Config-only construction (random init): there is no network for the roberta-base checkpoint; real
        weights arrive through FIBERTransformerSS's `load_path` checkpoint, whose keys this tree matches."""
        if name != "roberta-base":
            raise ValueError(f"unknown text backbone {name}")
        return cls(roberta_base_config(**over))

    @staticmethod
    def get_extended_attention_mask(attention_mask, input_shape=None, device=None):
        """transformers==4.6.0 semantics: (1 - mask)[:, None, None, :] * -10000.0 (fp32)."""
        return (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
