"""Registry boundary of the hot path (SURVEY.md 8(b) "Registry API").

The reference builds every component through mmengine registries
(`segdino3d/builder.py:3-82`): `build_architecture(cfg)` looks `cfg['type']` up in
`ARCHITECTURES` and calls the class with the remaining keys as kwargs.  This module exposes the
same registry names and `build_*` helpers.  When mmengine is importable its `Registry` /
`build_from_cfg` are used, so that the reference's unchanged `train_3d.py` sees genuine mmengine
objects; on boxes without mmengine (this image) a small compatible implementation is used.
"""
from __future__ import annotations

import copy
import inspect

try:  # pragma: no cover - mmengine is absent from the build image
    from mmengine import Registry, build_from_cfg  # type: ignore
    HAVE_MMENGINE = True
except Exception:  # noqa: BLE001
    HAVE_MMENGINE = False

    class Registry:
        """Name -> class table with the mmengine `register_module` decorator protocol."""

        def __init__(self, name: str):
            self.name = name
            self._table = {}

        def __contains__(self, key):
            return key in self._table

        def __len__(self):
            return len(self._table)

        def get(self, key):
            return self._table.get(key)

        @property
        def module_dict(self):
            return self._table

        def _add(self, cls, name=None, force=False):
            names = [name or cls.__name__] if not isinstance(name, (list, tuple)) else list(name)
            for n in names:
                if n in self._table and not force:
                    raise KeyError(f"{n} is already registered in {self.name}")
                self._table[n] = cls

        def register_module(self, name=None, force=False, module=None):
            if module is not None:
                self._add(module, name, force)
                return module

            def deco(cls):
                self._add(cls, name, force)
                return cls

            return deco

        def build(self, cfg, **default_args):
            return build_from_cfg(cfg, self, default_args or None)

        def __repr__(self):
            return f"Registry(name={self.name}, items={sorted(self._table)})"

    def build_from_cfg(cfg, registry, default_args=None):
        """`cfg['type']` (str or class) -> instance, other keys passed as kwargs."""
        if cfg is None:
            return None
        if not hasattr(cfg, "keys") or "type" not in cfg:
            raise KeyError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
        args = {k: cfg[k] for k in cfg.keys()}
        args = copy.copy(args)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        kind = args.pop("type")
        if isinstance(kind, str):
            cls = registry.get(kind)
            if cls is None:
                raise KeyError(f"{kind} is not in the {registry.name} registry")
        elif inspect.isclass(kind) or callable(kind):
            cls = kind
        else:
            raise TypeError(f"type must be a str or class, got {type(kind)}")
        return cls(**args)


BACKBONES = Registry("backbone")
NECKS = Registry("neck")
POS_EMBEDDINGS = Registry("position_embedding")
FUSERS = Registry("fuser")
ENCODERS = Registry("encoder")
DECODERS = Registry("decoder")
ARCHITECTURES = Registry("architecture")
TEXT_ENCODERS = Registry("text_encoder")
HEADS = Registry("head")
PREPARERS = Registry("preparer")
DATASETS = Registry("dataset")
TRANSFORMS = Registry("transform")
LOSSES = Registry("loss")
MATCHERS = Registry("matcher")
EVALUATORS = Registry("evaluator")


def _builder(registry):
    def build(cfg):
        return build_from_cfg(cfg, registry)
    build.__doc__ = f"Build a {registry.name} from its config dict."
    return build


build_backbone = _builder(BACKBONES)
build_neck = _builder(NECKS)
build_position_embedding = _builder(POS_EMBEDDINGS)
build_fuser = _builder(FUSERS)
build_encoder = _builder(ENCODERS)
build_decoder = _builder(DECODERS)
build_architecture = _builder(ARCHITECTURES)
build_text_encoder = _builder(TEXT_ENCODERS)
build_head = _builder(HEADS)
build_preparer = _builder(PREPARERS)
build_dataset = _builder(DATASETS)
build_transform = _builder(TRANSFORMS)
build_loss = _builder(LOSSES)
build_matcher = _builder(MATCHERS)
build_evaluator = _builder(EVALUATORS)

REGISTRY_NAMES = ["ARCHITECTURES", "BACKBONES", "DECODERS", "ENCODERS", "FUSERS", "POS_EMBEDDINGS",
                  "PREPARERS", "NECKS", "TEXT_ENCODERS", "HEADS", "DATASETS", "TRANSFORMS", "LOSSES",
                  "MATCHERS", "EVALUATORS"]
BUILDER_NAMES = ["build_architecture", "build_backbone", "build_decoder", "build_encoder",
                 "build_fuser", "build_position_embedding", "build_preparer", "build_transform",
                 "build_dataset", "build_neck", "build_text_encoder", "build_head", "build_loss",
                 "build_matcher", "build_evaluator"]
__all__ = REGISTRY_NAMES + BUILDER_NAMES + ["Registry", "build_from_cfg", "HAVE_MMENGINE"]
