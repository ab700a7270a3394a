"""Import shim that executes the REFERENCE's own hot-path files unmodified.  CONTAINER-ONLY.

Used solely by ``oracle/gen_golden.py`` in the build container, where /root/reference is
mounted; nothing here travels to the GPU box and nothing from the reference is copied.
It provides the third-party names the reference imports that are absent/moved in this
image (timm==0.4.12, transformers==4.6.0 -- SURVEY.md section 8c / Appendix C); the few that carry
arithmetic (PatchEmbed, Mlp, DropPath, trunc_normal_, _init_vit_weights) are restated from
the pinned versions' published semantics.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("FIBER_REFERENCE", "/root/reference")
MODS = os.path.join(REF, "coarse_grained", "fiber", "modules")


def available():
    return os.path.isfile(os.path.join(MODS, "swin_transformer.py"))


def _install_stubs():
    import transformers  # noqa: F401  (must be imported before the timm stubs exist)
    import transformers.modeling_utils as mu
    import transformers.file_utils as fu
    import transformers.pytorch_utils as pu

    if "timm" in sys.modules and getattr(sys.modules["timm"], "_fiber_stub", False):
        return

    def mod(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        return m

    timm = mod("timm")
    timm._fiber_stub = True
    data = mod("timm.data")
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    models = mod("timm.models")
    helpers = mod("timm.models.helpers")
    helpers.build_model_with_cfg = lambda *a, **k: None
    helpers.overlay_external_default_cfg = lambda default_cfg, kwargs: None
    layers = mod("timm.models.layers")
    registry = mod("timm.models.registry")
    registry.register_model = lambda f: f
    vt = mod("timm.models.vision_transformer")
    feats = mod("timm.models.features")
    hub = mod("timm.models.hub")
    for n in ("FeatureListNet", "FeatureDictNet", "FeatureHookNet"):
        setattr(feats, n, type(n, (nn.Module,), {}))
    hub.has_hf_hub = lambda *a, **k: False
    for n in ("download_cached_file", "load_state_dict_from_hf", "load_state_dict_from_url"):
        setattr(hub, n, lambda *a, **k: None)
    timm.data, timm.models = data, models
    models.helpers, models.layers, models.registry = helpers, layers, registry
    models.vision_transformer, models.features, models.hub = vt, feats, hub

    # ---- timm 0.4.12 arithmetic, restated ------------------------------------------------
    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
            super().__init__()
            img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
            self.img_size, self.patch_size = img_size, patch_size
            self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
            self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

        def forward(self, x):
            return self.norm(self.proj(x).flatten(2).transpose(1, 2))

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features or in_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    class DropPath(nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if not self.drop_prob or not self.training:
                return x
            keep = 1 - self.drop_prob
            r = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
            return x.div(keep) * r.floor_()

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    def _init_vit_weights(m, n="", head_bias=0.0, jax_impl=False):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.zeros_(m.bias)
            nn.init.ones_(m.weight)

    layers.PatchEmbed, layers.Mlp, layers.DropPath = PatchEmbed, Mlp, DropPath
    layers.to_2tuple, layers.trunc_normal_ = to_2tuple, trunc_normal_
    layers.Conv2dSame, layers.Linear = nn.Conv2d, nn.Linear
    vt.checkpoint_filter_fn = lambda sd, model: sd
    vt._init_vit_weights = _init_vit_weights

    # ---- transformers 4.6.0 names that moved in 5.x -------------------------------------------
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    if not hasattr(mu, "prune_linear_layer"):
        mu.prune_linear_layer = pu.prune_linear_layer
    if not hasattr(mu, "apply_chunking_to_forward"):
        mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    fu.add_code_sample_docstrings = lambda *a, **k: (lambda f: f)
    for n in ("add_start_docstrings", "add_start_docstrings_to_model_forward", "replace_return_docstrings"):
        if not hasattr(fu, n):
            setattr(fu, n, lambda *a, **k: (lambda f: f))


def _load(name, path, package):
    spec = importlib.util.spec_from_file_location(f"{package}.{name}", path, submodule_search_locations=None)
    m = importlib.util.module_from_spec(spec)
    sys.modules[f"{package}.{name}"] = m
    spec.loader.exec_module(m)
    return m


_CACHE = {}


def load_reference():
    """Return (swin_transformer, roberta) reference modules, executed from /root/reference by path."""
    if "mods" in _CACHE:
        return _CACHE["mods"]
    if not available():
        raise RuntimeError(f"reference not mounted at {REF}")
    _install_stubs()
    pkg = "_fiber_reference_modules"
    p = types.ModuleType(pkg)
    p.__path__ = [MODS]
    sys.modules[pkg] = p
    _load("swin_helpers", os.path.join(MODS, "swin_helpers.py"), pkg)
    sw = _load("swin_transformer", os.path.join(MODS, "swin_transformer.py"), pkg)
    rb = _load("roberta", os.path.join(MODS, "roberta.py"), pkg)
    _CACHE["mods"] = (sw, rb)
    return sw, rb


def roberta_config(**over):
    from transformers import RobertaConfig
    kw = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
              intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
              pad_token_id=1, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    kw.update(over)
    return RobertaConfig(**kw)
