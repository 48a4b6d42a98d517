"""Producer-side pyramid layout: the output convolutions of the image neck writing the
decoder's sampling layout.

The reference's image neck is mmdet's ``FPN`` (configs/racformer_r50_nuimg_704x256_f8.py:78-82; mmdet 2.28.2, a third-party
dependency that is not in the reference tree -- the in-tree ``CustomFPN`` has the same structure, models/necks/fpn.py:109-132,
180): per level one ``fpn_convs[i]`` = Conv2d(256, 256, 3, padding=1) with bias, no norm, no activation, applied to the
top-down-merged lateral of that level.  Its outputs ``[B*T*N, 256, H_l, W_l]`` reach the decoder as ``[B, T*N, 256, H_l,
W_l]``, which the decoder first copies into ``[B*T*G, N, H_l, W_l, 64]`` (models/racformer_transformer.py:112-124: 735 MB
read + 735 MB written per sample at f8).  ``FPNOutputWriter`` is that last stage of the neck with the copy folded into the
convolution's epilogue (rac_fpn_conv_fwd: the implicit-GEMM kernel of the temporal-fusion convolution, f16 matrix cores on
hi/lo-split operands, fp32-convolution accuracy); ``RaCFormerTransformerDecoder.pregrouped = True`` then consumes its
output as it stands.  Backbone and the lateral / top-down part of the neck stay out of scope."""
import torch
import torch.nn as nn

from . import _lib
from .fused import ConvImage, pack_conv3x3_weight


class FPNOutputWriter(nn.Module):
    """``fpn_convs`` of the image neck (state_dict keys ``fpn_convs.{i}.conv.{weight,bias}`` as in mmdet's FPN, whose
    fpn_convs are ConvModules)."""

    def __init__(self, num_levels=4, channels=256, num_cams=6, groups=4):
        super().__init__()
        if channels != 256 or groups != 4:
            raise ValueError("FPNOutputWriter: built for 256 output channels in 4 groups of 64 (the decoder's layout)")
        self.num_cams, self.groups = num_cams, groups

        class _ConvModule(nn.Module):
            def __init__(self):
                super().__init__()
                self.conv = nn.Conv2d(channels, channels, 3, padding=1)

        self.fpn_convs = nn.ModuleList([_ConvModule() for _ in range(num_levels)])
        self._packs = {}

    def _pack(self, i):
        w = self.fpn_convs[i].conv.weight
        sig = (w.data_ptr(), w._version, str(w.device))
        hit = self._packs.get(i)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = self._packs[i] = (sig, pack_conv3x3_weight(w))
        return hit[1]

    @torch.no_grad()
    def forward(self, laterals):
        """laterals: per level a float32 device tensor [B*T*N, 256, H_l, W_l] (image index (b*T + t)*N + cam) ->
        per level [B*T*G, N, H_l, W_l, 64], the ``mlvl_feats`` of a ``pregrouped`` decoder."""
        outs = []
        for i, x in enumerate(laterals):
            _lib.require_gpu(x, what="FPNOutputWriter")
            n, c, h, w = x.shape
            if c != 256 or x.dtype != torch.float32 or n % self.num_cams != 0:
                raise RuntimeError("FPNOutputWriter: laterals must be float32 [B*T*N, 256, H, W] with whole (batch, frame) groups")
            ws, alpha = self._pack(i)
            if ws is None:
                raise RuntimeError("FPNOutputWriter: weights cannot be held as f16 pairs (zero / non-finite weights)")
            img = ConvImage(n, h, w, c, x.device).begin([x]).pack(x, 0)
            out = torch.empty(n // self.num_cams * self.groups, self.num_cams, h, w, 64, device=x.device, dtype=torch.float32)
            bias = self.fpn_convs[i].conv.bias
            _lib.check(_lib.lib().rac_fpn_conv_fwd(_lib.ptr(img.xs), _lib.ptr(ws), _lib.ptr(bias) if bias is not None else None,
                                                   _lib.ptr(img.amax), float(alpha), _lib.ptr(out), n, h, w, c, self.num_cams,
                                                   _lib.stream_ptr()), "rac_fpn_conv_fwd")
            outs.append(out)
        return outs
