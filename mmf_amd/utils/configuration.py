"""Minimal attribute-dict configuration, used when OmegaConf (the reference's config library,
mmf/utils/configuration.py) is not installed.  OmegaConf `DictConfig`s are accepted everywhere a
`Config` is: only `.get`, `in`, attribute and item access are used."""
import collections


class Config(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, Config):
        return v
    if isinstance(v, collections.abc.Mapping):
        return Config(v)
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


def to_container(cfg):
    if isinstance(cfg, collections.abc.Mapping):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [to_container(v) for v in cfg]
    return cfg
