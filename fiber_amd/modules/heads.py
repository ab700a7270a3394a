"""Pooler / ITMHead / MLMHead / VQA classifier with the reference's parameter names
(coarse_grained/fiber/modules/heads.py:8-43, fiber_module.py:149-157).

Small caller-side GEMMs: the 768x768 dense layers use the HIP GEMM; the 2-way ITM classifier and the 50265-way
vocabulary decoder are plain library GEMMs ("keep on library GEMM").
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class Pooler(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        first = hidden_states[:, 0].contiguous()
        return torch.tanh(ops.linear(first.to(torch.bfloat16), self.dense.weight, self.dense.bias))


class ITMHead(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, 2)

    def forward(self, x):
        return ops.lib_linear(x.float(), self.fc.weight, self.fc.bias)


class BertPredictionHeadTransform(nn.Module):
    """transformers==4.6.0 BertPredictionHeadTransform: dense -> gelu(erf) -> LayerNorm(eps from config)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, x):
        h = ops.linear(x, self.dense.weight, self.dense.bias, act="gelu")
        return ops.layernorm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)


class MLMHead(nn.Module):
    def __init__(self, config, weight=None):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        if weight is not None:
            self.decoder.weight = weight
        self.last_hidden = None          # transform output of the latest forward (objectives._mlm_ce: fp32 label logits)

    def forward(self, x):
        h = self.transform(x)
        self.last_hidden = h.detach()
        return ops.lib_linear(h, ops.cast_bf16(self.decoder.weight), ops.cast_bf16(self.bias))


class VQAClassifier(nn.Sequential):
    """fiber_module.py:149-157 `vqa_classifier` = Sequential(Linear, LayerNorm, GELU, Linear); the Sequential base keeps the
    checkpoint keys (`vqa_classifier.0.weight`, `.1.weight`, `.3.weight`).  The 1536x1536 dense layer and the LayerNorm run
    on the HIP kernels; the 3129-way answer classifier is a library GEMM like the other caller-side heads."""

    def __init__(self, hidden, n_answers):
        super().__init__(nn.Linear(hidden, hidden), nn.LayerNorm(hidden), nn.GELU(), nn.Linear(hidden, n_answers))

    def forward(self, x):
        fc0, ln, _, fc1 = self
        h = ops.linear(x.to(torch.bfloat16), fc0.weight, fc0.bias)
        h = ops.layernorm(h, ln.weight, ln.bias, ln.eps)
        return ops.lib_linear(F.gelu(h.float()), fc1.weight, fc1.bias)
