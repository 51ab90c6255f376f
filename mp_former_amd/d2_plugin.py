"""Registry binding of the two HIP decoders (the plugin boundary B3 / B4 of SURVEY.md 8(b)).

The reference selects its decoders BY NAME from two registries:

* the pixel decoder from detectron2's ``SEM_SEG_HEADS_REGISTRY`` (mask2former/modeling/pixel_decoder/fpn.py:21-34,
  ``build_pixel_decoder``: ``SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME)(cfg, input_shape)``);
* the transformer decoder from ``TRANSFORMER_DECODER_REGISTRY``
  (mask2former/modeling/transformer_decoder/maskformer_transformer_decoder.py:16-28, ``build_transformer_decoder``:
  ``TRANSFORMER_DECODER_REGISTRY.get(cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME)(cfg, in_channels, mask_classification)``).

``register()`` adds ``MSDeformAttnPixelDecoderHIP`` and ``MultiScaleMaskedTransformerDecoderMaskDNHIP`` to them, so that a
stock MP-Former checkout runs the HIP path with two config overrides (run_50ep_no_noise_all_ly.sh:18-21):

    MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME MSDeformAttnPixelDecoderHIP
    MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME MultiScaleMaskedTransformerDecoderMaskDNHIP

Importing this module registers when detectron2 and mask2former are importable and is a no-op otherwise (this image has
neither: ``registered()`` is then False and tests/test_host_logic_cpu.py drives ``register`` with stub registries of
detectron2's ``Registry`` interface).  The construction convention of the registries — ``cls(cfg, ...)`` resolved through
``cls.from_config`` — is detectron2's ``@configurable``; where detectron2 is absent the same call convention is provided by
``_configurable`` below, so the registered classes behave identically in both worlds.
"""
import functools

from . import pixel_decoder as _P
from . import transformer_decoder as _T

PIXEL_DECODER_NAME = "MSDeformAttnPixelDecoderHIP"
TRANSFORMER_DECODER_NAME = "MultiScaleMaskedTransformerDecoderMaskDNHIP"

_registered = False


def _looks_like_cfg(x):
    """detectron2.config.config._called_with_cfg: a CfgNode (or anything with its attribute tree) in first position / as `cfg`."""
    return hasattr(x, "MODEL") and not isinstance(x, (int, float, str, dict, list, tuple))


def _configurable(init):
    """detectron2.config.configurable for ``__init__``: ``cls(cfg, *args)`` -> ``cls(**cls.from_config(cfg, *args))``;
    explicit keyword construction passes through."""
    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        cfg = args[0] if args else kwargs.get("cfg")
        if _looks_like_cfg(cfg):
            init(self, **type(self).from_config(*args, **kwargs))
        else:
            init(self, *args, **kwargs)
    return wrapped


def make_classes(configurable=None):
    """The two registrable subclasses; ``configurable`` = detectron2's decorator when it is there."""
    deco = configurable or _configurable

    class MSDeformAttnPixelDecoderHIP(_P.MSDeformAttnPixelDecoder):
        @deco
        def __init__(self, *a, **k):
            super().__init__(*a, **k)

    class MultiScaleMaskedTransformerDecoderMaskDNHIP(_T.MultiScaleMaskedTransformerDecoderMaskDN):
        @deco
        def __init__(self, *a, **k):
            super().__init__(*a, **k)

    MSDeformAttnPixelDecoderHIP.__name__ = MSDeformAttnPixelDecoderHIP.__qualname__ = PIXEL_DECODER_NAME
    MultiScaleMaskedTransformerDecoderMaskDNHIP.__name__ = MultiScaleMaskedTransformerDecoderMaskDNHIP.__qualname__ = TRANSFORMER_DECODER_NAME
    return MSDeformAttnPixelDecoderHIP, MultiScaleMaskedTransformerDecoderMaskDNHIP


def register(sem_seg_heads_registry=None, transformer_decoder_registry=None, configurable=None):
    """Register both classes.  Without arguments: detectron2's ``SEM_SEG_HEADS_REGISTRY`` and mask2former's
    ``TRANSFORMER_DECODER_REGISTRY`` (returns False, registering nothing, when either package is missing).  With
    arguments: any two objects with detectron2's ``Registry`` interface (``register(obj)`` / ``get(name)``).
    Returns the two classes (truthy) on success."""
    global _registered
    if sem_seg_heads_registry is None or transformer_decoder_registry is None:
        try:
            from detectron2.config import configurable as d2_configurable
            from detectron2.modeling import SEM_SEG_HEADS_REGISTRY
            from mask2former.modeling.transformer_decoder.maskformer_transformer_decoder import TRANSFORMER_DECODER_REGISTRY
        except ImportError:
            return False
        sem_seg_heads_registry = sem_seg_heads_registry or SEM_SEG_HEADS_REGISTRY
        transformer_decoder_registry = transformer_decoder_registry or TRANSFORMER_DECODER_REGISTRY
        configurable = configurable or d2_configurable
        default = True
    else:
        default = False
    pix, dec = make_classes(configurable)
    sem_seg_heads_registry.register(pix)
    transformer_decoder_registry.register(dec)
    if default:
        _registered = True
    return pix, dec


def registered():
    """True when the import of this module found detectron2 + mask2former and registered the classes there."""
    return _registered


register()
