"""Load the reference's hot-path Python files on CPU (THIS CONTAINER ONLY).

Test infrastructure, not product code.  Used only by ``tests/golden/gen_golden.py``
to emit golden vectors; nothing here travels to the GPU box in a usable form
(``/root/reference`` does not exist there) and nothing under ``racformer_amd/``
imports it.

The reference's ``models/__init__.py`` pulls mmdet / mmdet3d / flash_attn, which are
not installed.  We therefore register empty namespace packages whose ``__path__``
points into ``/root/reference/models`` and load the ten hot-path files one by one
(recipe: SURVEY.md Appendix C).  The handful of mmcv / mmdet names those files
import are stubbed below from the documented behaviour of mmcv-full 1.6.0 and
mmdet 2.28.2 (``README.md:44-45`` of the reference); each stub sits on a torch
primitive (``nn.MultiheadAttention``, ``F.grid_sample``) so the arithmetic is the
reference's own.  The stubs share no code with ``oracle/`` or ``racformer_amd/``.
"""
import importlib.util
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True  # keep /root/reference pristine

REF_ROOT = os.environ.get("RACFORMER_REFERENCE", "/root/reference")


# ----------------------------------------------------------------------------- stubs
class _BaseModule(nn.Module):
    """mmcv.runner.BaseModule: nn.Module + init_cfg bookkeeping."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg

    def init_weights(self):
        """mmcv 1.6.0 BaseModule.init_weights with ``init_cfg=None`` (the only case on this path): no parameter of the
        module itself is touched; every child that has an ``init_weights`` of its own is asked to run it."""
        assert self.init_cfg is None, "only init_cfg=None is stubbed"
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()
        self._is_init = True


def _identity_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn

    return wrap


def _bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


def _xavier_init(module, gain=1, bias=0, distribution="normal"):
    assert distribution in ("uniform", "normal")
    if hasattr(module, "weight") and module.weight is not None:
        if distribution == "uniform":
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


class _MultiheadAttention(_BaseModule):
    """mmcv.cnn.bricks.transformer.MultiheadAttention (1.6.0) semantics:
    identity + dropout(nn.MultiheadAttention(q, k, v, attn_mask)[0]) with its own
    batch_first transposes.  Third positional arg is attn_drop."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=None, init_cfg=None, batch_first=False, **kwargs):
        super().__init__(init_cfg)
        if "dropout" in kwargs:
            attn_drop = kwargs.pop("dropout")
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_pos=None, attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query = query.transpose(0, 1)
            key = key.transpose(0, 1)
            value = value.transpose(0, 1)
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


class _FFN(_BaseModule):
    """mmcv FFN (1.6.0): [Linear-ReLU-Drop] x (num_fcs-1), Linear, Drop, + identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=None, ffn_drop=0.0, dropout_layer=None, add_identity=True,
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(in_channels, feedforward_channels),
                                        nn.ReLU(inplace=True), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


class _LearnedPositionalEncoding(_BaseModule):
    """mmdet 2.28.2 LearnedPositionalEncoding."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__(init_cfg)
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.num_feats = num_feats
        self.row_num_embed = row_num_embed
        self.col_num_embed = col_num_embed

    def forward(self, mask):
        h, w = mask.shape[-2:]
        x = torch.arange(w, device=mask.device)
        y = torch.arange(h, device=mask.device)
        x_embed = self.col_embed(x)
        y_embed = self.row_embed(y)
        pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1),
                         y_embed.unsqueeze(1).repeat(1, w, 1)), dim=-1)
        pos = pos.permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)
        return pos


def _build_positional_encoding(cfg, default_args=None):
    cfg = dict(cfg)
    kind = cfg.pop("type")
    assert kind == "LearnedPositionalEncoding", kind
    return _LearnedPositionalEncoding(**cfg)


def _msda_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """mmcv.ops.multi_scale_deform_attn.multi_scale_deformable_attn_pytorch (1.6.0)."""
    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, num_heads, num_levels, num_points, _ = sampling_locations.shape
    value_list = value.split([int(H_ * W_) for H_, W_ in value_spatial_shapes], dim=1)
    sampling_grids = 2 * sampling_locations - 1
    sampling_value_list = []
    for level, (H_, W_) in enumerate(value_spatial_shapes):
        H_, W_ = int(H_), int(W_)
        value_l_ = value_list[level].flatten(2).transpose(1, 2).reshape(
            bs * num_heads, embed_dims, H_, W_)
        sampling_grid_l_ = sampling_grids[:, :, :, level].transpose(1, 2).flatten(0, 1)
        sampling_value_l_ = F.grid_sample(value_l_, sampling_grid_l_, mode="bilinear",
                                          padding_mode="zeros", align_corners=False)
        sampling_value_list.append(sampling_value_l_)
    attention_weights = attention_weights.transpose(1, 2).reshape(
        bs * num_heads, 1, num_queries, num_levels * num_points)
    output = (torch.stack(sampling_value_list, dim=-2).flatten(-2) * attention_weights)
    output = output.sum(-1).view(bs, num_heads * embed_dims, num_queries)
    return output.transpose(1, 2).contiguous()


class _Registry:
    """mmcv Registry, plumbing only: remembers the decorated classes by name so that a config's
    ``type`` can be resolved (``build``); no behaviour of its own."""

    def __init__(self):
        self.classes = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.classes[cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.classes[cfg.pop("type")](**cfg)


class _BaseBBoxCoder:
    """mmdet.core.bbox.BaseBBoxCoder (2.28.2): an abstract base without arithmetic."""

    def __init__(self, **kwargs):
        pass


class _LiDARInstance3DBoxes:
    """mmdet3d LiDARInstance3DBoxes as used by get_bboxes (racformer_head.py:503): a holder of the
    [n, box_dim] tensor; none of its geometry helpers are on the path."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        assert tuple(origin) == (0.5, 0.5, 0), "only the default (bottom-centre) origin is stubbed"
        self.tensor = tensor
        self.box_dim = box_dim


_TRANSFORMER, _BBOX_CODERS, _HEADS = _Registry(), _Registry(), _Registry()


class _DETRHead(_BaseModule):
    """mmdet 2.28.2 DETRHead.__init__ reduced to its plumbing: remember the constructor arguments the
    subclass reads, build ``transformer`` from its config through the TRANSFORMER registry, call the
    subclass' ``_init_layers``.  The losses / assigner / sine positional encoding it also builds are
    training-only or unused by RaCFormer_head.forward and are not instantiated."""

    def __init__(self, num_classes, in_channels, num_query=100, num_reg_fcs=2, transformer=None,
                 sync_cls_avg_factor=False, positional_encoding=None, loss_cls=None, loss_bbox=None,
                 loss_iou=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.num_query, self.num_classes, self.in_channels = num_query, num_classes, in_channels
        self.num_reg_fcs, self.train_cfg, self.test_cfg = num_reg_fcs, train_cfg, test_cfg
        self.fp16_enabled = False
        self.transformer = _TRANSFORMER.build(transformer)
        self.embed_dims = self.transformer.embed_dims
        self._init_layers()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    if "mmcv" in sys.modules and getattr(sys.modules["mmcv"], "_rac_stub", False):
        return
    _mod("mmcv", _rac_stub=True)
    _mod("mmcv.runner", BaseModule=_BaseModule, auto_fp16=_identity_decorator,
         force_fp32=_identity_decorator)
    _mod("mmcv.runner.base_module", BaseModule=_BaseModule)
    _mod("mmcv.cnn", bias_init_with_prob=_bias_init_with_prob, xavier_init=_xavier_init)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.transformer", MultiheadAttention=_MultiheadAttention, FFN=_FFN,
         build_positional_encoding=_build_positional_encoding)
    ext_loader = _mod("mmcv.utils.ext_loader", load_ext=lambda *a, **k: types.SimpleNamespace())
    _mod("mmcv.utils", ext_loader=ext_loader)
    _mod("mmcv.ops")
    _mod("mmcv.ops.multi_scale_deform_attn", multi_scale_deformable_attn_pytorch=_msda_pytorch)
    _mod("mmdet")
    _mod("mmdet.models")
    _mod("mmdet.models.utils")
    _mod("mmdet.models.utils.builder", TRANSFORMER=_TRANSFORMER)
    # decode / head end of the path (models/bbox/coders/nms_free_coder.py:3-4, models/racformer_head.py:4-9)
    _mod("mmdet.core", multi_apply=None, reduce_mean=None)       # training-only helpers, never called here
    _mod("mmdet.core.bbox", BaseBBoxCoder=_BaseBBoxCoder)
    _mod("mmdet.core.bbox.builder", BBOX_CODERS=_BBOX_CODERS)
    sys.modules["mmdet.models"].HEADS = _HEADS
    _mod("mmdet.models.dense_heads", DETRHead=_DETRHead)
    _mod("mmdet3d")
    _mod("mmdet3d.core")
    _mod("mmdet3d.core.bbox")
    _mod("mmdet3d.core.bbox.coders", build_bbox_coder=_BBOX_CODERS.build)
    _mod("mmdet3d.core.bbox.structures")
    _mod("mmdet3d.core.bbox.structures.lidar_box3d", LiDARInstance3DBoxes=_LiDARInstance3DBoxes)


_FILES = [
    ("models.utils", "models/utils.py"),
    ("models.bbox.utils", "models/bbox/utils.py"),
    ("models.csrc.wrapper", "models/csrc/wrapper.py"),
    ("models.checkpoint", "models/checkpoint.py"),
    ("models.sparsebev_sampling", "models/sparsebev_sampling.py"),
    ("models.multi_scale_deformable_attn_function", "models/multi_scale_deformable_attn_function.py"),
    ("models.bev_self_attention", "models/bev_self_attention.py"),
    ("models.racformer_transformer", "models/racformer_transformer.py"),
    ("models.bbox.coders.nms_free_coder", "models/bbox/coders/nms_free_coder.py"),
    ("models.racformer_head", "models/racformer_head.py"),
]


def load_reference():
    """Returns a namespace with the loaded reference modules (CPU fallback path)."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}; goldens can only be "
                           "generated in the build container")
    _install_stubs()
    for pkg, sub in (("models", "models"), ("models.bbox", "models/bbox"),
                     ("models.csrc", "models/csrc"), ("models.bbox.coders", "models/bbox/coders")):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF_ROOT, sub)]
            sys.modules[pkg] = m
    out = {}
    import contextlib
    import io
    for name, rel in _FILES:
        if name in sys.modules:
            out[name.split(".")[-1]] = sys.modules[name]
            continue
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        with contextlib.redirect_stdout(io.StringIO()):  # wrapper.py prints an import warning
            spec.loader.exec_module(mod)
        out[name.split(".")[-1]] = mod
    return types.SimpleNamespace(**out)
