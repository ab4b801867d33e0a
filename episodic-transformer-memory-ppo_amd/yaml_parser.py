"""YAML config -> nested dict (same keys as upstream configs; PyYAML instead of ruamel which this image lacks)."""
import re

import yaml


class _Loader(yaml.SafeLoader):
    """SafeLoader with the YAML 1.2 float grammar: PyYAML follows YAML 1.1, where ``3e-4`` / ``1e-5`` (no dot) are strings;
    upstream's ruamel parser (YAML 1.2) reads them as floats, and the schedules of the configs are written that way."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+]?[0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)
                    |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


class YamlParser:
    def __init__(self, path):
        with open(path, "r") as stream:
            docs = [d for d in yaml.load_all(stream, Loader=_Loader) if d is not None]
        self._config = dict(docs[-1]) if docs else {}

    def get_config(self):
        return self._config
